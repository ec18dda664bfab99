// amg_setup.hpp -- host side of the smoothed-aggregation AMG setup (hierarchy construction).
//
// What it has to reproduce: the hierarchy amgcl::amg builds for the parameters the reference passes in
// /root/reference/src/polysolve/linear/AMGCL.cpp:32-65 (smoothed_aggregation coarsening with
// plain aggregates, eps_strong 0, estimate_spectral_radius true; hierarchy limits max_levels /
// coarse_enough).  This file is the all-host construction (threaded SpGEMM): it serves the block
// coarsening (block_size > 1), "amg.device_setup" = 0 and the GPU-free hierarchy tests.  The scalar
// default path builds the patterns and numbers on the device (amg_symbolic.hip, kernels.hip) and only
// runs the greedy aggregation sweep here -- it is inherently sequential in AMGCL and is kept sequential
// so that the hierarchy is the one the CPU oracle builds.
#pragma once
#include <cstdint>
#include <vector>

namespace psolve {

struct HostCsr {
    int64_t nrows = 0, ncols = 0;
    std::vector<int32_t> ptr, col;
    std::vector<double> val;
    int64_t nnz() const { return ptr.empty() ? 0 : ptr.back(); }
};

struct HostLevel {
    HostCsr A;        // level operator
    HostCsr P, R;     // to / from the next coarser level (empty on the coarsest)
    double omega = 0; // smoothing weight used for P
    int64_t naggregates = 0;
    // symbolic data kept for the numeric-only refresh on the device (same pattern, new values)
    std::vector<int32_t> id;        // aggregate of every row (negative = removed)
    std::vector<int32_t> r_from_p;  // R.val[k] = P.val[r_from_p[k]]
    HostCsr AP;                     // pattern of A * P (values are scratch)
};

struct AmgParams;

// Gershgorin bound on rho(D^-1 A)  (amgcl/backend/builtin.hpp spectral_radius<true>(A, 0))
double gershgorin_scaled(const HostCsr &A);

// plain aggregation (amgcl/coarsening/plain_aggregates.hpp); returns the aggregate count, fills
// id[n] (negative = removed) and strong[nnz]
int64_t plain_aggregates(const HostCsr &A, double eps_strong, std::vector<int32_t> &id, std::vector<char> &strong,
                         int mode = 0); // mode 1: "amg.aggregation" = "parallel" (hashed-priority distance-2 independent set)

// the same sweep on a compacted strength graph (strong off-diagonals + the stored diagonal per row, as
// amg_symbolic.hip builds it on the device); returns the aggregate count, fills id[n].  id_initialised:
// id[] already holds the start state (-1 undefined / -2 removed) computed with the graph.
int64_t aggregate_strength_graph(int64_t n, const int32_t *sptr, const int32_t *scol, std::vector<int32_t> &id,
                                 bool id_initialised = false, int mode = 0);

// P = (I - omega D_f^-1 A_f) P_tent  (amgcl/coarsening/smoothed_aggregation.hpp), sorted columns
HostCsr smoothed_prolongation(const HostCsr &A, const std::vector<char> &strong, const std::vector<int32_t> &id,
                              int64_t nagg, double omega);
HostCsr tentative_prolongation(int64_t n_nodes, const std::vector<int32_t> &id, int64_t nagg, int bs); // "amg.coarsening" = "aggregation"
double over_interp_scale(double over_interp, int bs); // 1 / over_interp as amgcl computes it (a float)
HostCsr transpose(const HostCsr &A, std::vector<int32_t> *entry_map = nullptr);
HostCsr multiply(const HostCsr &A, const HostCsr &B); // threaded Gustavson, sorted columns

// Block value types (polysolve's AMGCL_Block<3>, AMGCL.cpp:243-302): zero-filled b x b block view
struct HostBcsr {
    int64_t nb = 0;
    int b = 1;
    std::vector<int32_t> ptr, col; // block rows / sorted block columns
    std::vector<double> val;       // b*b per block, row-major
};
HostBcsr to_blocks(const HostCsr &A, int b);
void invert_block(int b, const double *X, double *Y); // Gauss-Jordan, partial pivoting, b <= 4

// Builds all levels.  `fine` is consumed (moved into level 0).  prm.block_size > 1 selects the block
// coarsening (aggregation on the block graph, block-diagonal smoothing of P, block Gershgorin).
std::vector<HostLevel> build_hierarchy(HostCsr &&fine, const AmgParams &prm);

} // namespace psolve
