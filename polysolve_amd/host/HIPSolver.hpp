// HIPSolver.hpp -- the C++ adapter a PolySolve build adds to get Solver::create("HIP").
//
//     class HIPSolver : public polysolve::linear::Solver
//
// Header-only, pimpl-free (the pimpl is the C handle), modelled on the way the reference wires its
// in-tree GPU backend (src/polysolve/linear/MASSolver.hpp:36-71, MASSolver.cu:597-650) and with the
// solve semantics of EigenIterative (src/polysolve/linear/EigenSolver.tpp:68-114).  It needs the
// reference's own headers (<polysolve/linear/Solver.hpp>, Eigen, nlohmann::json), so it compiles inside a
// PolySolve tree (integration/apply_hip_hooks.py adds the six hooks) -- and against the small interface
// stand-in under tests/stubs/, which is how this file is compiled and RUN in this repository's tests
// (tests/test_adapter.py).  Everything numerical happens behind include/psolve_hip.h in libpsolve_hip.so.
#pragma once

#if __has_include(<polysolve/linear/Solver.hpp>)

#include <polysolve/linear/Solver.hpp>

#include <psolve_hip.h>

#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace polysolve::linear
{
    class HIPSolver : public Solver
    {
    public:
        /// @param precond  the factory's preconditioner string (Solver.cpp:606-609): "" and
        ///                 "Eigen::DiagonalPreconditioner" -> Jacobi, "Eigen::IdentityPreconditioner" -> none;
        ///                 any other name runs the solver's default like the reference (Solver.cpp:194-198),
        ///                 with a warning, since e.g. Eigen::IncompleteCholesky is honoured there
        explicit HIPSolver(const std::string &precond = "", const std::vector<int> &devices = {0})
        {
            open(devices);
            if (precond == "Eigen::IncompleteCholesky")
                std::fprintf(stderr, "[HIP] note: Eigen::IncompleteCholesky = precond \"ic\": Eigen's factorization in its default "
                                     "(AMD) ordering, both restated from the published algorithms, not validated against Eigen\n");
            else if (!precond.empty() && precond != "Eigen::DiagonalPreconditioner" && precond != "Eigen::IdentityPreconditioner")
                std::fprintf(stderr, "[HIP] warning: preconditioner '%s' is not available in the HIP backend; using Jacobi "
                                     "(params[\"HIP\"][\"precond\"] selects none / jacobi / amg / ic)\n", precond.c_str());
            set("precond", precond == "Eigen::IdentityPreconditioner" ? 0 : (precond == "Eigen::IncompleteCholesky" ? 4 : 1));
        }
        ~HIPSolver() override { psolve_hip_destroy(h_); }
        POLYSOLVE_DELETE_MOVE_COPY(HIPSolver)

        // Solver.hpp:90 -- reads params["HIP"] only (EigenSolver.tpp:68-82, MASSolver.cu:605-614); the keys,
        // types and defaults are the `/HIP` rules of integration/linear-solver-spec.hip.json
        // (The JSON calls below are the ones every nlohmann::json release has -- count(), iterators with key() / value(),
        // get<T>() -- not contains() (3.6), structured bindings over items() (3.x late) or the implicit conversions a build may
        // switch off: tests/test_adapter.py compiles and runs this header against the real library where the image has one.)
        static bool has(const json &o, const char *k) { return o.is_object() && o.count(k) > 0; }
        static bool has(const json &o, const std::string &k) { return o.is_object() && o.count(k) > 0; }

        void set_parameters(const json &params) override
        {
            if (!has(params, name()))
                return;
            const json &p = params[name()];
            if (has(p, "devices") && p["devices"].is_array())
            {
                std::vector<int> ids;
                for (const json &d : p["devices"])
                    ids.push_back(d.get<int>());
                if (!ids.empty() && ids != devices_)
                    open(ids); // new handle on the listed devices; the parameters set so far are replayed
            }
            for (auto it = p.begin(); it != p.end(); ++it)
            {
                const std::string key = it.key();
                const json &value = it.value();
                if (key == "devices" || key == "tolerance" || key == "amgcl_params")
                    continue;
                if (key == "precond" && value.is_string())
                {
                    const std::string s = value.get<std::string>();
                    if (!s.empty()) // empty: keep what the factory's precond string selected
                        set("precond", s == "none" ? 0 : (s == "amg" ? 2 : (s == "schwarz" ? 3 : (s == "ic" ? 4 : 1))));
                }
                else if ((key == "amg" || key == "schwarz" || key == "ic") && value.is_object())
                {
                    for (auto i2 = value.begin(); i2 != value.end(); ++i2)
                    {
                        const json &v2 = i2.value();
                        const std::string k2 = i2.key();
                        // /HIP/amg/{aggregation, coarsening, relax_type} are STRINGS in the spec (their defaults arrive here
                        // through the factory's inject_defaults, Solver.cpp:152-155): names -> psolve_hip_set_param codes
                        if (v2.is_string())
                            set(key + "." + k2, name_code(key, k2, v2.get<std::string>()));
                        else
                            set(key + "." + k2, v2.is_boolean() ? (v2.get<bool>() ? 1.0 : 0.0) : v2.get<double>());
                    }
                }
                else if (value.is_boolean())
                    set(key, value.get<bool>() ? 1.0 : 0.0);
                else
                    set(key, value.get<double>());
            }
            // "tolerance" (the Eigen solvers' key) is an alias that wins over relative_tolerance; negative = not set
            if (has(p, "tolerance") && p["tolerance"].get<double>() >= 0)
                set("tolerance", p["tolerance"].get<double>());
            // "amgcl_params": the reference's params["AMGCL"] block as well (AMGCL.cpp:32-128), for callers who switch
            // "solver" from "AMGCL" to "HIP".  What that block defines -- by the reference's defaults or by the caller --
            // wins over the /HIP keys above (after the factory's inject_defaults those cannot be told from the spec's
            // defaults, Solver.cpp:152)
            if (has(p, "amgcl_params") && p["amgcl_params"].is_boolean() && p["amgcl_params"].get<bool>())
                apply_amgcl_block(params);
        }

        // Solver.hpp:93 -- both key families the reference's callers read
        void get_info(json &params) const override
        {
            psolve_hip_info i;
            check(psolve_hip_get_info(h_, &i));
            params["solver_iter"] = i.solver_iter;       // EigenSolver.tpp:88
            params["solver_error"] = i.solver_error;     // EigenSolver.tpp:89
            params["num_iterations"] = i.num_iterations; // AMGCL.cpp:142
            params["final_res_norm"] = i.final_res_norm; // AMGCL.cpp:143
            static const char *status[] = {"Running", "Reach relative tolerance", "Reach absolute tolerance",
                                           "Reach max iterations", "Non-finite residual"}; // MASSolver.hpp:18-33 (+1)
            params["solver_status"] = status[i.solver_status >= 0 && i.solver_status <= 4 ? i.solver_status : 0];
            params["true_residual"] = i.true_residual;
            params["amg_levels"] = i.amg_levels;
            params["time_factorize"] = i.time_factorize;
            params["time_solve"] = i.time_solve;
        }

        // Solver.hpp:96
        void analyze_pattern(const StiffnessMatrix &A, const int precond_num) override
        {
            const StiffnessMatrix &C = compressed(A);
            check(psolve_hip_analyze_pattern(h_, C.rows(), C.nonZeros(), idx32(C.outerIndexPtr(), C.rows() + 1, outer32_),
                                             idx32(C.innerIndexPtr(), C.nonZeros(), inner32_), precond_num));
        }

        // Solver.hpp:99 -- failures become std::runtime_error, which Newton catches (Newton.cpp:191-202)
        void factorize(const StiffnessMatrix &A) override
        {
            if (A.rows() != A.cols())
                throw std::runtime_error("[HIP] square matrix expected");
            const StiffnessMatrix &C = compressed(A);
            // ColMajor arrays of a symmetric matrix == its CSR arrays (AMGCL.hpp:36-43)
            check(psolve_hip_factorize(h_, C.rows(), C.nonZeros(), idx32(C.outerIndexPtr(), C.rows() + 1, outer32_),
                                       idx32(C.innerIndexPtr(), C.nonZeros(), inner32_), C.valuePtr()));
            n_ = C.rows();
        }

        bool is_dense() const override { return false; } // sparse Newton rejects dense solvers (Newton.cpp:72-73)
        void set_block_size(int block_size) override { set("block_size", block_size); }
        void set_tolerance(const double tol) override { set("tolerance", tol); }

        // Solver.hpp:128 -- x is the initial guess on entry (Solver.hpp:119-127); non-convergence is
        // not an error (the caller inspects get_info / the residual, Newton.cpp:207)
        void solve(const Ref<const VectorXd> b, Ref<VectorXd> x) override
        {
            if (b.size() != x.size() || b.size() != n_)
                throw std::runtime_error("[HIP] Size mismatch. Did you forget to call factorize?"); // MASSolver.cu:380-383
            // Ref<VectorXd> has inner stride 1 by default: data() is contiguous (AMGCL.cpp:203-204 relies on the same)
            check(psolve_hip_solve(h_, b.data(), x.data()));
        }

        std::string name() const override { return "HIP"; }

    private:
        // one device: psolve_hip_create; several: the in-process multi-device handle (one host thread and one
        // RCCL rank per device inside this process; factorize splits the rows, solve scatters / gathers)
        void open(const std::vector<int> &devices)
        {
            if (h_)
                psolve_hip_destroy(h_);
            h_ = nullptr;
            const int rc = devices.size() == 1 ? psolve_hip_create(&h_, devices[0])
                                               : psolve_hip_create_multi(&h_, devices.data(), (int)devices.size());
            if (rc != PSOLVE_HIP_OK)
                throw std::runtime_error(std::string("[HIP] ") + psolve_hip_last_error(nullptr));
            devices_ = devices;
            n_ = -1;
            for (const auto &kv : set_log_)
                check(psolve_hip_set_param(h_, kv.first.c_str(), kv.second));
        }
        void set(const std::string &key, double v)
        {
            check(psolve_hip_set_param(h_, key.c_str(), v));
            set_log_[key] = v;
        }
        void check(int rc) const
        {
            if (rc != PSOLVE_HIP_OK)
                throw std::runtime_error(std::string("[HIP] ") + psolve_hip_last_error(h_));
        }
        // the string-valued /HIP/<block>/<key> parameters (integration/linear-solver-spec.hip.json) and their codes
        // (solver.hpp: AmgParams; the same tables as polysolve_amd/solver.py AMG_NAMES); an unknown name is refused by name
        static double name_code(const std::string &block, const std::string &key, const std::string &got)
        {
            struct Table { const char *block, *key; std::initializer_list<const char *> names; };
            static const Table tables[] = {{"amg", "aggregation", {"amgcl", "parallel", "compact"}},
                                           {"amg", "coarsening", {"smoothed_aggregation", "aggregation"}},
                                           {"amg", "relax_type", {"chebyshev", "damped_jacobi", "spai0", "gauss_seidel", "ilu0"}},
                                           {"amg", "class", {"amg", "relaxation"}}};
            for (const Table &t : tables)
            {
                if (block != t.block || key != t.key)
                    continue;
                int code = 0;
                std::string all;
                for (const char *nm : t.names)
                {
                    if (got == nm)
                        return code;
                    ++code;
                    all += (all.empty() ? "" : " | ") + std::string(nm);
                }
                throw std::runtime_error("[HIP] " + block + "." + key + " = '" + got + "': not one of " + all);
            }
            throw std::runtime_error("[HIP] " + block + "." + key + " = '" + got + "': this parameter takes a number, not a name");
        }
        // params["AMGCL"] = {"precond": {...}, "solver": {...}, "block_size": b} patched over the reference's defaults
        // (AMGCL.cpp:32-65, set_params :67-92) -> the parameters that build the same solver here.  cg + amg with coarsening
        // smoothed_aggregation | aggregation and relaxation chebyshev | damped_jacobi | spai0 | gauss_seidel | ilu0, or the class
        // "relaxation" (that smoother alone), are built; anything else is refused.
        void apply_amgcl_block(const json &params)
        {
            static const json none;
            const json &a = has(params, "AMGCL") ? params["AMGCL"] : none;
            const json &pre = has(a, "precond") ? a["precond"] : none;
            const json &sol = has(a, "solver") ? a["solver"] : none;
            const json &rel = has(pre, "relax") ? pre["relax"] : none;
            const json &coa = has(pre, "coarsening") ? pre["coarsening"] : none;
            const json &agg = has(coa, "aggr") ? coa["aggr"] : none;
            auto num = [](const json &o, const char *k, double dflt) { return has(o, k) ? o[k].get<double>() : dflt; };
            auto flag = [](const json &o, const char *k, bool dflt) {
                return has(o, k) ? (o[k].is_boolean() ? o[k].get<bool>() : o[k].get<double>() != 0.0) : dflt;
            };
            // (round 5) amgcl's runtime wrappers build whatever the free strings name (AMGCL.cpp:67-92,
            // linear-solver-spec.json:393-397, 423-427); this backend builds cg + amg with coarsening smoothed_aggregation |
            // aggregation and relaxation chebyshev | damped_jacobi | spai0 | gauss_seidel | ilu0 (round 6: ordered sweeps),
            // direct_coarse either way, class amg | relaxation
            auto choice = [](const json &o, const char *k, const char *dflt, std::initializer_list<const char *> names) {
                const std::string got = (has(o, k) && o[k].is_string()) ? o[k].get<std::string>() : std::string(dflt);
                int code = 0;
                for (const char *nm : names) {
                    if (got == nm) return code;
                    ++code;
                }
                std::string all;
                for (const char *nm : names) all += (all.empty() ? "" : " | ") + std::string(nm);
                throw std::runtime_error(std::string("[HIP] AMGCL ") + k + " = '" + got + "': the HIP backend builds " + all + " only");
            };
            choice(sol, "type", "cg", {"cg"});
            set("amg.class", choice(pre, "class", "amg", {"amg", "relaxation"}));
            const int coarsening = choice(coa, "type", "smoothed_aggregation", {"smoothed_aggregation", "aggregation"});
            const int relax_type = choice(rel, "type", "chebyshev", {"chebyshev", "damped_jacobi", "spai0", "gauss_seidel", "ilu0"});
            set("amg.coarsening", coarsening);
            set("amg.relax_type", relax_type);
            set("amg.direct_coarse", flag(pre, "direct_coarse", false) ? 1 : 0);
            if (relax_type == 0) set("amg.cheb_scale", flag(rel, "scale", true) ? 1 : 0);
            if (relax_type == 1 && has(rel, "damping")) set("amg.damping", num(rel, "damping", 0.72));
            if (relax_type == 4 && has(rel, "damping")) set("amg.ilu_damping", num(rel, "damping", 1.0));
            if (coarsening == 1 && has(coa, "over_interp")) set("amg.over_interp", num(coa, "over_interp", 1.5));
            set("precond", 2);
            set("tolerance", num(sol, "tol", 1e-10));
            set("max_iter", num(sol, "maxiter", 1000));
            if (has(sol, "abstol"))
                set("absolute_tolerance", num(sol, "abstol", 0.0));
            set("amg.max_levels", num(pre, "max_levels", 6));
            set("amg.ncycle", num(pre, "ncycle", 2));
            // (amgcl parameters the reference's defaults do not spell out: only when the caller's block does)
            if (has(pre, "npre")) set("amg.npre", num(pre, "npre", 1));
            if (has(pre, "npost")) set("amg.npost", num(pre, "npost", 1));
            if (has(pre, "coarse_enough")) set("amg.coarse_enough", num(pre, "coarse_enough", 3000));
            if (relax_type == 0) {
                set("amg.cheb_degree", num(rel, "degree", 16));
                set("amg.cheb_power_iters", num(rel, "power_iters", 100));
                set("amg.cheb_higher", num(rel, "higher", 2));
                set("amg.cheb_lower", num(rel, "lower", 0.008333333333));
            }
            if (coarsening == 0) {
                set("amg.sa_relax", num(coa, "relax", 1));
                set("amg.estimate_spectral_radius", flag(coa, "estimate_spectral_radius", true) ? 1 : 0);
            }
            if (has(coa, "power_iters")) set("amg.sa_power_iters", num(coa, "power_iters", 0));
            set("amg.eps_strong", num(agg, "eps_strong", 0));
            if (has(a, "block_size"))
                set("block_size", a["block_size"].get<double>());
        }
        // The C ABI is int32 per shard (like MAS, BSRMatrix.cu:438-442).  A build with POLYSOLVE_LARGE_INDEX
        // (Types.hpp:11-15: StorageIndex = std::ptrdiff_t) hands over 64-bit indices: narrowed into a scratch copy,
        // refused when a value does not fit.  The default build (int) passes the caller's arrays through.
        template <typename I>
        static const int32_t *idx32(const I *p, long long count, std::vector<int32_t> &buf)
        {
            if constexpr (sizeof(I) == sizeof(int32_t))
                return reinterpret_cast<const int32_t *>(p);
            else
            {
                buf.resize((size_t)(count > 0 ? count : 0));
                for (long long i = 0; i < count; ++i)
                {
                    if (p[i] < 0 || p[i] > (I)2147483647)
                        throw std::runtime_error("[HIP] matrix exceeds int32 indexing (n or nnz >= 2^31): partition it over more GPUs");
                    buf[(size_t)i] = (int32_t)p[i];
                }
                return buf.data();
            }
        }
        // MAS copies + compresses uncompressed input (BSRMatrix.cu:444-452); so do we
        const StiffnessMatrix &compressed(const StiffnessMatrix &A)
        {
            if (A.isCompressed())
                return A;
            tmp_ = A;
            tmp_.makeCompressed();
            return tmp_;
        }

        psolve_hip_t h_ = nullptr;
        std::vector<int> devices_;
        std::map<std::string, double> set_log_;
        long long n_ = -1;
        StiffnessMatrix tmp_;
        std::vector<int32_t> outer32_, inner32_; // POLYSOLVE_LARGE_INDEX only
    };
} // namespace polysolve::linear

#endif // __has_include(<polysolve/linear/Solver.hpp>)
