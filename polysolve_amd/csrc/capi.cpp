// capi.cpp -- the extern "C" boundary (include/psolve_hip.h).  Exceptions stop here.
#include <cstring>
#include <mutex>
#include <string>

#include "amg_setup.hpp"
#include "ic.hpp"
#include "host_hash.hpp"
#include "multi.hpp"
#include "solver.hpp"

using psolve::Context;
using psolve::Error;
using psolve::MultiContext;

// One handle = one Context (one device) or one MultiContext (several devices of the node, one process).
struct psolve_hip_ctx {
    std::unique_ptr<Context> single;
    std::unique_ptr<MultiContext> multi;
    std::string &last_error() { return single ? single->last_error : multi->last_error; }
};

static std::string g_create_error;
static std::mutex g_create_mutex;

template <typename F>
static int guarded_handle(psolve_hip_t h, F &&f)
{
    if (!h) return PSOLVE_HIP_EINVAL;
    try {
        f();
        return PSOLVE_HIP_OK;
    } catch (const Error &e) {
        h->last_error() = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        h->last_error() = "host allocation failed";
        return PSOLVE_HIP_EDEVICE;
    } catch (const std::exception &e) {
        h->last_error() = e.what();
        return PSOLVE_HIP_EINVAL;
    }
}

// entry points that exist on one device only (device pointers belong to ONE device)
template <typename F>
static int guarded(psolve_hip_t h, F &&f)
{
    return guarded_handle(h, [&] {
        PS_REQUIRE(h->single != nullptr, PSOLVE_HIP_EINVAL,
                   "this entry point takes a single-device handle (psolve_hip_create); a multi-device handle "
                   "serves the host contract: set_param / analyze_pattern / factorize / solve / get_info");
        f(*h->single);
    });
}

// the host contract: same call on either kind of handle
template <typename FS, typename FM>
static int guarded_any(psolve_hip_t h, FS &&fs, FM &&fm)
{
    return guarded_handle(h, [&] {
        if (h->single) fs(*h->single);
        else fm(*h->multi);
    });
}

// failures outside any handle (create, host-only helpers)
template <typename F>
static int guarded_global(F &&f)
{
    try {
        f();
        return PSOLVE_HIP_OK;
    } catch (const Error &e) {
        std::lock_guard<std::mutex> g(g_create_mutex);
        g_create_error = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        std::lock_guard<std::mutex> g(g_create_mutex);
        g_create_error = "host allocation failed";
        return PSOLVE_HIP_EDEVICE;
    } catch (const std::exception &e) {
        std::lock_guard<std::mutex> g(g_create_mutex);
        g_create_error = e.what();
        return PSOLVE_HIP_EINVAL;
    }
}

extern "C" {

int psolve_hip_abi_version(void) { return PSOLVE_HIP_ABI_VERSION; }

int psolve_hip_device_count(int *count)
{
    if (!count) return PSOLVE_HIP_EINVAL;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        *count = 0;
        return PSOLVE_HIP_EDEVICE;
    }
    *count = c;
    return PSOLVE_HIP_OK;
}

int psolve_hip_default_param(const char *key, double *value)
{
    if (!key || !value) return PSOLVE_HIP_EINVAL;
    return guarded_global([&] {
        const psolve::Params defaults;
        PS_REQUIRE(psolve::param_value(defaults, key, value), PSOLVE_HIP_EINVAL,
                   std::string("unknown parameter '") + key + "'");
    });
}

int psolve_hip_create(psolve_hip_t *out, int device_id)
{
    if (!out) return PSOLVE_HIP_EINVAL;
    *out = nullptr;
    return guarded_global([&] {
        std::unique_ptr<psolve_hip_ctx> h(new psolve_hip_ctx());
        h->single.reset(new Context(device_id));
        *out = h.release();
    });
}

int psolve_hip_create_multi(psolve_hip_t *out, const int *device_ids, int n_devices)
{
    if (!out) return PSOLVE_HIP_EINVAL;
    *out = nullptr;
    return guarded_global([&] {
        PS_REQUIRE(device_ids && n_devices >= 1, PSOLVE_HIP_EINVAL, "create_multi: empty device list");
        std::unique_ptr<psolve_hip_ctx> h(new psolve_hip_ctx());
        if (n_devices == 1) h->single.reset(new Context(device_ids[0]));
        else h->multi.reset(new MultiContext(device_ids, n_devices));
        *out = h.release();
    });
}

void psolve_hip_destroy(psolve_hip_t h) { delete h; }

const char *psolve_hip_last_error(psolve_hip_t h)
{
    if (!h) return g_create_error.c_str();
    return h->last_error().c_str();
}

int psolve_hip_last_spmv_kernel(psolve_hip_t h, char *buf, int buf_len)
{
    if (!buf || buf_len <= 0) return PSOLVE_HIP_EINVAL;
    buf[0] = 0;
    return guarded(h, [&](Context &c) {
        const std::string &k = c.last_spmv_kernel();
        std::snprintf(buf, (size_t)buf_len, "%s", k.c_str());
    });
}

int psolve_hip_last_pcg_kernel(psolve_hip_t h, int which, char *buf, int buf_len)
{
    if (!buf || buf_len <= 0 || which < 0 || which > 2) return PSOLVE_HIP_EINVAL;
    buf[0] = 0;
    return guarded(h, [&](Context &c) {
        const std::string &k = which == 0 ? c.last_spmv_kernel() : c.last_vec_kernel(which - 1);
        std::snprintf(buf, (size_t)buf_len, "%s", k.c_str());
    });
}

int psolve_hip_trim(psolve_hip_t h)
{
    return guarded_any(h, [&](Context &c) { c.use_device(); c.synchronize(); c.meter.trim(); }, [&](MultiContext &m) { m.trim(); });
}

int psolve_hip_set_stream(psolve_hip_t h, void *s)
{
    return guarded(h, [&](Context &c) { c.set_stream(s); });
}

int psolve_hip_synchronize(psolve_hip_t h)
{
    return guarded_any(h, [&](Context &c) { c.synchronize(); }, [&](MultiContext &m) { m.synchronize(); });
}

int psolve_hip_set_param(psolve_hip_t h, const char *key, double value)
{
    return guarded_any(
        h,
        [&](Context &c) {
            PS_REQUIRE(key, PSOLVE_HIP_EINVAL, "null key");
            c.set_param(key, value);
        },
        [&](MultiContext &m) {
            PS_REQUIRE(key, PSOLVE_HIP_EINVAL, "null key");
            m.set_param(key, value);
        });
}

int psolve_hip_get_param(psolve_hip_t h, const char *key, double *value)
{
    return guarded_any(
        h,
        [&](Context &c) {
            PS_REQUIRE(key && value, PSOLVE_HIP_EINVAL, "null key/value");
            *value = std::string(key) == "devices" ? 1.0 : c.get_param(key);
        },
        [&](MultiContext &m) {
            PS_REQUIRE(key && value, PSOLVE_HIP_EINVAL, "null key/value");
            *value = m.get_param(key);
        });
}

int psolve_hip_analyze_pattern(psolve_hip_t h, int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner,
                               int precond_num)
{
    return guarded_any(
        h, [&](Context &c) { c.analyze_pattern(n, nnz, outer, inner, precond_num); },
        [&](MultiContext &m) { m.analyze_pattern(n, nnz, outer, inner, precond_num); });
}

int psolve_hip_factorize(psolve_hip_t h, int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner,
                         const double *values)
{
    return guarded_any(
        h, [&](Context &c) { c.factorize_host(n, nnz, outer, inner, values); },
        [&](MultiContext &m) { m.factorize_host(n, nnz, outer, inner, values); });
}

int psolve_hip_solve(psolve_hip_t h, const double *b, double *x)
{
    return guarded_any(h, [&](Context &c) { c.solve_host(b, x); }, [&](MultiContext &m) { m.solve_host(b, x); });
}

int psolve_hip_get_info(psolve_hip_t h, psolve_hip_info *info)
{
    return guarded_any(
        h,
        [&](Context &c) {
            PS_REQUIRE(info, PSOLVE_HIP_EINVAL, "null info");
            *info = c.info;
        },
        [&](MultiContext &m) {
            PS_REQUIRE(info, PSOLVE_HIP_EINVAL, "null info");
            *info = m.info;
        });
}

int psolve_hip_shard_rows(psolve_hip_t h, int shard, int64_t *row_begin, int64_t *row_end, int *device_id)
{
    return guarded_any(
        h,
        [&](Context &c) {
            PS_REQUIRE(shard == 0, PSOLVE_HIP_EINVAL, "shard_rows: a single-device handle has one shard");
            if (row_begin) *row_begin = 0;
            if (row_end) *row_end = c.A.n;
            if (device_id) *device_id = c.device;
        },
        [&](MultiContext &m) {
            PS_REQUIRE(shard >= 0 && shard < m.world() && (int)m.row_offsets().size() == m.world() + 1,
                       PSOLVE_HIP_EINVAL, "shard_rows: no such shard (or factorize has not run)");
            if (row_begin) *row_begin = m.row_offsets()[(size_t)shard];
            if (row_end) *row_end = m.row_offsets()[(size_t)shard + 1];
            if (device_id) *device_id = m.shard(shard).device;
        });
}

int psolve_hip_factorize_device(psolve_hip_t h, int64_t n_local, int64_t nnz_local, const int32_t *d_rowptr,
                                const int32_t *d_col, const double *d_values)
{
    return guarded(h, [&](Context &c) { c.factorize_device(n_local, nnz_local, d_rowptr, d_col, d_values, false); });
}

int psolve_hip_solve_device(psolve_hip_t h, const double *d_b, double *d_x)
{
    return guarded(h, [&](Context &c) { c.solve_device(d_b, d_x); });
}

int psolve_hip_generate_poisson7(psolve_hip_t h, int nx, int ny, int nz, int z0, int z1)
{
    return guarded(h, [&](Context &c) { c.generate_poisson7(nx, ny, nz, z0, z1); });
}

int psolve_hip_generate_elasticity_q1(psolve_hip_t h, int M, double E, double nu)
{
    return guarded(h, [&](Context &c) { c.generate_elasticity_q1(M, E, nu); });
}

int psolve_hip_generate_elasticity_q1_permuted(psolve_hip_t h, int M, double E, double nu, int mode, int64_t window,
                                               uint64_t seed)
{
    return guarded(h, [&](Context &c) { c.generate_elasticity_q1_permuted(M, E, nu, mode, window, seed); });
}

int psolve_hip_generate_poisson7_permuted(psolve_hip_t h, int nx, int ny, int nz, int mode, int64_t window, uint64_t seed)
{
    return guarded(h, [&](Context &c) { c.generate_poisson7_permuted(nx, ny, nz, mode, window, seed); });
}

int psolve_hip_permutation(int64_t n, int mode, int64_t window, uint64_t seed, int32_t *new_index)
{
    return guarded_global([&] {
        PS_REQUIRE(n > 0 && n < (int64_t)INT32_MAX && new_index && (mode == 1 || (mode == 2 && window >= 2)), PSOLVE_HIP_EINVAL,
                   "permutation: bad arguments");
        psolve::permutation_host(n, mode, window, seed, new_index);
    });
}

int psolve_hip_generate_rhs(psolve_hip_t h, uint64_t seed, double *d_b, double *d_xstar)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(d_b, PSOLVE_HIP_EINVAL, "null d_b");
        c.generate_rhs(seed, d_b, d_xstar);
    });
}

int psolve_hip_spmv_device(psolve_hip_t h, const double *d_x, double *d_y)
{
    return guarded(h, [&](Context &c) { c.spmv(d_x, d_y); });
}

int psolve_hip_spmv_dot_device(psolve_hip_t h, const double *d_x, double *d_y, double *xy)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(xy, PSOLVE_HIP_EINVAL, "null result");
        *xy = c.spmv_dot(d_x, d_y);
    });
}

int psolve_hip_dot_device(psolve_hip_t h, int64_t n, const double *d_a, const double *d_b, double *out)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(out, PSOLVE_HIP_EINVAL, "null result");
        *out = c.dot(n, d_a, d_b);
    });
}

int psolve_hip_axpby_device(psolve_hip_t h, int64_t n, double a, const double *d_x, double b, double *d_y)
{
    return guarded(h, [&](Context &c) { c.axpby(n, a, d_x, b, d_y); });
}

int psolve_hip_precond_apply_device(psolve_hip_t h, const double *d_r, double *d_z)
{
    return guarded(h, [&](Context &c) { c.precond_apply(d_r, d_z); });
}

int psolve_hip_time_spmv(psolve_hip_t h, const double *d_x, double *d_y, int reps, double *ms_avg)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(ms_avg, PSOLVE_HIP_EINVAL, "null result");
        *ms_avg = c.time_spmv(d_x, d_y, reps);
    });
}

int psolve_hip_time_vecops(psolve_hip_t h, int reps, double *ms_update_avg, double *ms_direction_avg)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(ms_update_avg && ms_direction_avg, PSOLVE_HIP_EINVAL, "null result");
        c.time_vecops(reps, ms_update_avg, ms_direction_avg);
    });
}

int psolve_hip_box_probe(psolve_hip_t h, double *out, int n_out)
{
    return guarded(h, [&](Context &c) { c.box_probe(out, n_out); });
}

int psolve_hip_malloc(psolve_hip_t h, void **d_ptr, size_t bytes)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(d_ptr, PSOLVE_HIP_EINVAL, "null out pointer");
        c.use_device();
        PS_HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 1));
    });
}

int psolve_hip_free(psolve_hip_t h, void *d_ptr)
{
    return guarded(h, [&](Context &c) {
        c.use_device();
        PS_HIP_CHECK(hipStreamSynchronize(c.stream));
        if (d_ptr) PS_HIP_CHECK(hipFree(d_ptr));
    });
}

int psolve_hip_memcpy_h2d(psolve_hip_t h, void *d_dst, const void *src, size_t bytes)
{
    return guarded(h, [&](Context &c) {
        c.use_device();
        PS_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c.stream));
        PS_HIP_CHECK(hipStreamSynchronize(c.stream));
    });
}

int psolve_hip_memcpy_d2h(psolve_hip_t h, void *dst, const void *d_src, size_t bytes)
{
    return guarded(h, [&](Context &c) {
        c.use_device();
        PS_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c.stream));
        PS_HIP_CHECK(hipStreamSynchronize(c.stream));
    });
}

int psolve_hip_matrix_shape(psolve_hip_t h, int64_t *n_local, int64_t *nnz_local, int64_t *n_halo)
{
    return guarded(h, [&](Context &c) {
        if (n_local) *n_local = c.A.n;
        if (nnz_local) *nnz_local = c.A.nnz;
        if (n_halo) *n_halo = c.n_halo();
    });
}

int psolve_hip_host_pattern_hash(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int threads,
                                 uint64_t out[2])
{
    if (n < 0 || nnz < 0 || !outer || (!inner && nnz > 0) || !out) return PSOLVE_HIP_EINVAL;
    const psolve::HostPatternHash h = psolve::hash_host_pattern(n, nnz, outer, inner, threads);
    out[0] = h.outer;
    out[1] = h.inner;
    return PSOLVE_HIP_OK;
}

int psolve_hip_amd_order(int64_t n, const int32_t *outer, const int32_t *inner, int32_t *order)
{
    if (n < 0 || !outer || (!inner && n > 0 && outer[n] > 0) || (!order && n > 0)) return PSOLVE_HIP_EINVAL;
    try {
        std::vector<int32_t> o;
        psolve::amd_order(n, outer, inner, o);
        for (int64_t i = 0; i < n; ++i) order[i] = o[(size_t)i];
    } catch (...) {
        return PSOLVE_HIP_ERANGE;
    }
    return PSOLVE_HIP_OK;
}

int psolve_hip_matrix_copy(psolve_hip_t h, int32_t *rowptr, int32_t *col, double *val)
{
    return guarded(h, [&](Context &c) { c.matrix_copy(rowptr, col, val); });
}

int psolve_hip_amg_level_info(psolve_hip_t h, int level, int64_t *rows, int64_t *nnz, double *rho)
{
    return guarded(h, [&](Context &c) { c.amg_level_info(level, rows, nnz, rho); });
}

int psolve_hip_amg_time_level_ops(psolve_hip_t h, int level, int reps, double out_us[5])
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(out_us, PSOLVE_HIP_EINVAL, "null result");
        c.amg_time_level_ops(level, reps, out_us);
    });
}

int psolve_hip_amg_level_matrix_shape(psolve_hip_t h, int level, int what, int64_t out[3])
{
    if (!out) return PSOLVE_HIP_EINVAL;
    return guarded(h, [&](Context &c) { c.amg_level_matrix_shape(level, what, out); });
}

int psolve_hip_amg_level_matrix_copy(psolve_hip_t h, int level, int what, int32_t *rowptr, int32_t *col, double *val)
{
    if (!rowptr || !col || !val) return PSOLVE_HIP_EINVAL;
    return guarded(h, [&](Context &c) { c.amg_level_matrix_copy(level, what, rowptr, col, val); });
}

int psolve_hip_ic_host_factorize(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                                 double initial_shift, int32_t *colptr, int32_t *rowidx, double *vals, double *scale,
                                 double *shift, int *attempts)
{
    return guarded_global([&] {
        PS_REQUIRE(n > 0 && rowptr && col && val && colptr && rowidx && vals && scale, PSOLVE_HIP_EINVAL,
                   "ic_host_factorize: null / empty arguments");
        psolve::IcFactor F;
        psolve::ic_factorize(n, rowptr, col, val, initial_shift, F);
        PS_REQUIRE(F.ok, PSOLVE_HIP_ENUMERIC, "incomplete Cholesky: no positive pivots after 10 shifts (matrix not SPD?)");
        std::memcpy(colptr, F.colptr.data(), F.colptr.size() * sizeof(int32_t));
        std::memcpy(rowidx, F.rowidx.data(), F.rowidx.size() * sizeof(int32_t));
        std::memcpy(vals, F.vals.data(), F.vals.size() * sizeof(double));
        std::memcpy(scale, F.scale.data(), F.scale.size() * sizeof(double));
        if (shift) *shift = F.shift;
        if (attempts) *attempts = F.attempts;
    });
}

int psolve_hip_amg_level_perm(psolve_hip_t h, int level, int32_t *perm, int *renumbered)
{
    return guarded(h, [&](Context &c) {
        const bool r = c.amg_level_perm(level, perm);
        if (renumbered) *renumbered = r ? 1 : 0;
    });
}

int psolve_hip_reorder_perm(psolve_hip_t h, int32_t *new_of_old, int *reordered)
{
    return guarded_any(
        h,
        [&](Context &c) {
            PS_REQUIRE(new_of_old != nullptr, PSOLVE_HIP_EINVAL, "reorder_perm: null output");
            const bool r = c.reorder_perm(new_of_old);
            if (reordered) *reordered = r ? 1 : 0;
        },
        [&](psolve::MultiContext &m) {
            PS_REQUIRE(new_of_old != nullptr, PSOLVE_HIP_EINVAL, "reorder_perm: null output");
            if (m.reordered()) std::memcpy(new_of_old, m.new_of_old().data(), m.new_of_old().size() * sizeof(int32_t));
            if (reordered) *reordered = m.reordered() ? 1 : 0;
        });
}

// ---- host-only view of the AMG setup (no GPU needed; what the CPU tests compare with the oracle) ----
struct psolve_hip_amg_host {
    std::vector<psolve::HostLevel> levels;
};

static const psolve::HostCsr *pick(const psolve_hip_amg_host *H, int level, int what)
{
    if (!H || level < 0 || level >= (int)H->levels.size()) return nullptr;
    const psolve::HostLevel &L = H->levels[(size_t)level];
    const psolve::HostCsr *M = what == 0 ? &L.A : what == 1 ? &L.P : what == 2 ? &L.R : nullptr;
    if (!M || (what != 0 && M->nrows == 0)) return nullptr;
    return M;
}

int psolve_hip_amg_host_build(psolve_hip_amg_host_t *out, int64_t n, int64_t nnz, const int32_t *rowptr,
                              const int32_t *col, const double *val, int max_levels, int coarse_enough,
                              double eps_strong, double sa_relax, int estimate_spectral_radius, int block_size,
                              int *n_levels)
{
    if (!out || !rowptr || !col || !val || n <= 0 || nnz < 0 || !n_levels) return PSOLVE_HIP_EINVAL;
    *out = nullptr;
    return guarded_global([&] {
        PS_REQUIRE(rowptr[0] == 0 && rowptr[n] == nnz, PSOLVE_HIP_EINVAL,
                   "amg_host_build: rowptr[0] != 0 or rowptr[n] != nnz");
        psolve::HostCsr A;
        A.nrows = A.ncols = n;
        A.ptr.assign(rowptr, rowptr + n + 1);
        A.col.assign(col, col + nnz);
        A.val.assign(val, val + nnz);
        psolve::AmgParams prm;
        prm.max_levels = max_levels;
        prm.coarse_enough = coarse_enough;
        prm.eps_strong = eps_strong;
        prm.sa_relax = sa_relax;
        prm.estimate_spectral_radius = estimate_spectral_radius;
        prm.block_size = block_size > 1 ? block_size : 1;
        auto *H = new psolve_hip_amg_host();
        H->levels = psolve::build_hierarchy(std::move(A), prm);
        *n_levels = (int)H->levels.size();
        *out = H;
    });
}

int psolve_hip_amg_host_build2(psolve_hip_amg_host_t *out, int64_t n, int64_t nnz, const int32_t *rowptr,
                               const int32_t *col, const double *val, int max_levels, int coarse_enough,
                               double eps_strong, double sa_relax, int estimate_spectral_radius, int block_size,
                               int aggregation, int coarsening, double over_interp, int *n_levels)
{
    if (!out || !rowptr || !col || !val || n <= 0 || nnz < 0 || !n_levels) return PSOLVE_HIP_EINVAL;
    *out = nullptr;
    return guarded_global([&] {
        PS_REQUIRE(rowptr[0] == 0 && rowptr[n] == nnz, PSOLVE_HIP_EINVAL,
                   "amg_host_build: rowptr[0] != 0 or rowptr[n] != nnz");
        PS_REQUIRE(aggregation >= 0 && aggregation <= 2 && coarsening >= 0 && coarsening <= 1, PSOLVE_HIP_EINVAL,
                   "amg_host_build: aggregation / coarsening out of range");
        psolve::HostCsr A;
        A.nrows = A.ncols = n;
        A.ptr.assign(rowptr, rowptr + n + 1);
        A.col.assign(col, col + nnz);
        A.val.assign(val, val + nnz);
        psolve::AmgParams prm;
        prm.max_levels = max_levels;
        prm.coarse_enough = coarse_enough;
        prm.eps_strong = eps_strong;
        prm.sa_relax = sa_relax;
        prm.estimate_spectral_radius = estimate_spectral_radius;
        prm.block_size = block_size > 1 ? block_size : 1;
        prm.aggregation = aggregation;
        prm.coarsening = coarsening;
        prm.over_interp = over_interp;
        auto *H = new psolve_hip_amg_host();
        H->levels = psolve::build_hierarchy(std::move(A), prm);
        *n_levels = (int)H->levels.size();
        *out = H;
    });
}

int psolve_hip_amg_host_level_shape(psolve_hip_amg_host_t H, int level, int what, int64_t out[3], double *omega)
{
    const psolve::HostCsr *M = pick(H, level, what);
    if (!M || !out) return PSOLVE_HIP_EINVAL;
    out[0] = M->nrows;
    out[1] = M->ncols;
    out[2] = M->nnz();
    if (omega) *omega = H->levels[(size_t)level].omega;
    return PSOLVE_HIP_OK;
}

int psolve_hip_amg_host_level_copy(psolve_hip_amg_host_t H, int level, int what, int32_t *rowptr, int32_t *col,
                                   double *val)
{
    const psolve::HostCsr *M = pick(H, level, what);
    if (!M || !rowptr || !col || !val) return PSOLVE_HIP_EINVAL;
    std::memcpy(rowptr, M->ptr.data(), M->ptr.size() * sizeof(int32_t));
    std::memcpy(col, M->col.data(), (size_t)M->nnz() * sizeof(int32_t));
    std::memcpy(val, M->val.data(), (size_t)M->nnz() * sizeof(double));
    return PSOLVE_HIP_OK;
}

void psolve_hip_amg_host_free(psolve_hip_amg_host_t H) { delete H; }

int psolve_hip_comm_unique_id(char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path)
{
    if (!id) return PSOLVE_HIP_EINVAL;
    return guarded_global([&] { psolve::Comm::unique_id(id, rccl_path); });
}

int psolve_hip_comm_init(psolve_hip_t h, int rank, int world, const char id[PSOLVE_HIP_UNIQUE_ID_BYTES],
                         const char *rccl_path)
{
    return guarded(h, [&](Context &c) {
        PS_REQUIRE(id, PSOLVE_HIP_EINVAL, "null id");
        c.comm_init(rank, world, id, rccl_path);
    });
}

int psolve_hip_local_group_create(psolve_hip_local_group_t *out, int world)
{
    if (!out) return PSOLVE_HIP_EINVAL;
    return guarded_global([&] { *out = (psolve_hip_local_group_t)psolve::local_group_create(world); });
}

void psolve_hip_local_group_destroy(psolve_hip_local_group_t g) { psolve::local_group_destroy((psolve::LocalGroup *)g); }

int psolve_hip_comm_init_local(psolve_hip_t h, psolve_hip_local_group_t g, int rank)
{
    return guarded(h, [&](Context &c) { c.comm_init_local((psolve::LocalGroup *)g, rank); });
}

int psolve_hip_set_partition(psolve_hip_t h, int64_t n_global, int64_t row_begin, int64_t row_end)
{
    return guarded(h, [&](Context &c) { c.set_partition(n_global, row_begin, row_end); });
}

int psolve_hip_partition_rows(int64_t n, const int32_t *outer, int world, int block_size, int64_t *row_offsets)
{
    if (!outer || !row_offsets || n <= 0) return PSOLVE_HIP_EINVAL;
    return guarded_global([&] {
        std::vector<int64_t> off;
        psolve::partition_rows_by_nnz(n, outer, world, block_size, off);
        std::memcpy(row_offsets, off.data(), off.size() * sizeof(int64_t));
    });
}

int psolve_hip_plan_halo(int rank, int world, const int64_t *row_offsets, int64_t n_cols, const int32_t *cols,
                         int32_t *halo_out, int64_t *n_halo, int64_t *recv_counts)
{
    if (!row_offsets || (!cols && n_cols > 0) || !halo_out || !n_halo || !recv_counts) return PSOLVE_HIP_EINVAL;
    return guarded_global([&] {
        std::vector<int32_t> halo;
        std::vector<int64_t> rc;
        psolve::plan_halo(rank, world, row_offsets, n_cols, cols, halo, rc);
        std::memcpy(halo_out, halo.data(), halo.size() * sizeof(int32_t));
        *n_halo = (int64_t)halo.size();
        std::memcpy(recv_counts, rc.data(), rc.size() * sizeof(int64_t));
    });
}

} // extern "C"
