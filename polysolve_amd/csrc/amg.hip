// amg.hip -- device side of the Chebyshev-smoothed aggregation AMG preconditioner.
//
// Hierarchy: scalar systems are coarsened on the device (device_full_setup below: patterns by
// amg_symbolic.hip, numbers by the kernels.hip setup kernels, only the sequential aggregation sweep on
// the host); block_size > 1 and "amg.device_setup" = 0 take the all-host construction of amg_setup.cpp
// and upload it.  Here: level storage in HBM, the Chebyshev spectral radius estimate (power iterations
// as fused SpMV launches) and the cycle.  The cycle restates
// amgcl::amg::cycle / apply and amgcl::relaxation::chebyshev::solve (restated for the CPU in
// oracle/amg_oracle.c) with every vector operation fused into an SpMV epilogue:
//     Chebyshev step   : spmv_csr_pipe<SPMV_CHEB>      res = D^-1 (f - A x); p = a res + b p; x' = x + p
//     residual         : spmv_csr_pipe<SPMV_RESIDUAL>  t = f - A x
//     restriction      : spmv_csr_pipe<SPMV_PLAIN> on R
//     prolongation     : spmv_csr_pipe<SPMV_ADD>   on P   x += P u
// so one pre- or post-smoothing of degree k is exactly k launches, and a level visit moves
// (2k + 1) x (12 nnz + ~44 n) bytes plus the two transfer operators.
#include "amg.hpp"
#include <functional>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <future>

#include "amg_setup.hpp"
#include "amg_symbolic.hpp"
#include "sell.hpp"
#include "solver.hpp"

namespace psolve {

namespace {

struct DevCsr {
    DeviceBuffer<int> ptr, col;
    DeviceBuffer<double> val;
    DeviceBuffer<float> val32; // optional single-precision copy of val ("amg.matrix_fp32")
    CsrDev view;
    void set_fp32(const Launch &L, bool on)
    {
        view.val32 = nullptr;
        if (!on || view.nnz == 0) return;
        val32.ensure((size_t)view.nnz + 4);
        launch_to_f32(L, view.nnz, val.ptr, val32.ptr);
        view.val32 = val32.ptr;
    }
    // 16-bit columns (CsrDev::col16) for the products of the cycle: built once per pattern (the values do not enter)
    Col16 c16;
    void set_col16(const Launch &L, bool on)
    {
        view.col16 = nullptr;
        view.rb_base = nullptr;
        view.col16_R = 0;
        if (!on || view.val32 || view.sell || view.n < 4096 || view.nnz <= 0) return;
        if (!c16.valid && !c16_tried) {
            c16_tried = true;
            c16.build(L, view);
        }
        if (c16.valid) c16.attach(view);
    }
    bool c16_tried = false;
    // wide-row operators (several threads per row on spmv_csr_dma): row-blocks of variable height, packed on the host from
    // the row pointers once per pattern -- every block fits the LDS tile in one pass (CsrDev::rb_start)
    DeviceBuffer<int> rbs;
    bool rbs_valid = false;
    int rbs_count = 0, rbs_R = 0, rbs_tile = 0;
    void set_row_blocks(const Launch &L)
    {
        view.rb_start = nullptr;
        view.rb_count = 0;
        const int R = view.rows_per_block;
        if (!L.lab.var_row_blocks || R >= 256 || view.n < 4096 || view.nnz <= 0 || view.val32 || view.col16 || view.sell || view.bsr3) return;
        const int tile = spmv_dma_tile(L.lab, R, (double)view.nnz / (double)view.n);
        if (!rbs_valid || rbs_R != R || rbs_tile != tile) {
            const double t0 = wall_seconds();
            std::vector<int> hp((size_t)view.n + 1), starts;
            PS_HIP_CHECK(hipMemcpyAsync(hp.data(), view.rowptr, hp.size() * sizeof(int), hipMemcpyDeviceToHost, L.stream));
            PS_HIP_CHECK(hipStreamSynchronize(L.stream));
            pack_row_blocks(view.n, hp.data(), R, tile, starts);
            rbs.ensure(starts.size() + 4);
            PS_HIP_CHECK(hipMemcpyAsync(rbs.ptr, starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice, L.stream));
            PS_HIP_CHECK(hipStreamSynchronize(L.stream));
            rbs_count = (int)starts.size() - 1;
            rbs_R = R;
            rbs_tile = tile;
            rbs_valid = true;
            if (std::getenv("PSOLVE_TIMING"))
                std::fprintf(stderr, "[psolve timing] amg row-blocks packed to the tile  rows=%d blocks=%d %.4f s\n", view.n, rbs_count, wall_seconds() - t0);
        }
        view.rb_start = rbs.ptr;
        view.rb_count = rbs_count;
        view.rb_R = rbs_R;
        view.rb_tile = rbs_tile;
    }
    // "amg.sell": wide-row operators multiply through a SELL-64-sigma copy (built once per pattern, refilled
    // with the numbers of a refresh); narrow ones (7-point level 0, the prolongations) keep the row-block kernels
    SellMatrix sell;
    void set_sell(const Launch &L, int mode, SymbolicScratch &S)
    {
        view.sell = nullptr;
        if (!mode || view.val32 || view.n <= 0 || view.nnz <= 0) return;
        if (mode == 1 && (view.n < 4096 || view.nnz < 12ll * view.n)) return;
        if (sell.valid) sell.refill(L, view);
        else sell.build(L, view, S, mode == 2 ? 64.0 : 1.25);
        if (sell.valid) view.sell = &sell.view;
    }
    void set_view(int nrows, int ncols, int64_t nnz)
    {
        view.n = nrows;
        view.n_ext = ncols;
        view.nnz = nnz;
        view.rowptr = ptr.ptr;
        view.col = col.ptr;
        view.val = val.ptr;
        view.rows_per_block = spmv_rows_per_block(nrows ? (double)nnz / (double)nrows : 1.0);
        view.col16 = nullptr; // (a new pattern: the 16-bit columns are encoded again when the cycle's copies are made)
        view.rb_base = nullptr;
        view.col16_R = 0;
        c16.valid = false;
        c16_tried = false;
        rbs_valid = false; // (a new pattern)
        view.rb_start = nullptr;
        view.rb_count = 0;
    }
    void upload(const HostCsr &H, hipStream_t s)
    {
        const size_t n = (size_t)H.nrows, nnz = (size_t)H.nnz();
        ptr.ensure(n + 1);
        col.ensure(nnz + 4);
        val.ensure(nnz + 4);
        PS_HIP_CHECK(hipMemcpyAsync(ptr.ptr, H.ptr.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        if (nnz) {
            PS_HIP_CHECK(hipMemcpyAsync(col.ptr, H.col.data(), nnz * sizeof(int), hipMemcpyHostToDevice, s));
            PS_HIP_CHECK(hipMemcpyAsync(val.ptr, H.val.data(), nnz * sizeof(double), hipMemcpyHostToDevice, s));
        }
        PS_HIP_CHECK(hipStreamSynchronize(s)); // H may be a temporary
        view.n = (int)H.nrows;
        view.n_ext = (int)H.ncols;
        view.nnz = (int64_t)nnz;
        view.rowptr = ptr.ptr;
        view.col = col.ptr;
        view.val = val.ptr;
        view.rows_per_block = spmv_rows_per_block(n ? (double)nnz / (double)n : 1.0);
    }
};

// std::mt19937 + libstdc++'s uniform_real_distribution<double>(-1, 1): the start vector of
// amgcl::backend::spectral_radius (seed = thread id 0).  Block form: one state refill (624 words) yields
// 312 doubles; the loops carry no modulo and vectorise.
struct Mt19937 {
    uint32_t mt[624], out[624];
    explicit Mt19937(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    static inline uint32_t twist(uint32_t a, uint32_t b, uint32_t m)
    {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return m ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    void refill()
    {
        for (int i = 0; i < 227; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i + 397]);
        for (int i = 227; i < 623; ++i) mt[i] = twist(mt[i], mt[i + 1], mt[i - 227]);
        mt[623] = twist(mt[623], mt[0], mt[396]);
        for (int i = 0; i < 624; ++i) {
            uint32_t y = mt[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[i] = y;
        }
    }
    // v[0..n) = the first n draws of uniform_real_distribution<double>(-1, 1)
    void fill(double *v, size_t n)
    {
        size_t k = 0;
        while (k < n) {
            refill();
            const size_t m = std::min<size_t>(n - k, 312);
            for (size_t j = 0; j < m; ++j) {
                const double lo = (double)out[2 * j], hi = (double)out[2 * j + 1];
                double c = (lo + hi * 4294967296.0) * 5.421010862427522170037264004349708557128906250e-20; // / 2^64
                if (c >= 1.0) c = std::nextafter(1.0, 0.0);
                v[k + j] = 2.0 * c - 1.0;
            }
            k += m;
        }
    }
};

} // namespace

struct Level {
    int sweep = 0; // direction of the next product on A_l inside a cycle (toggled per launch, reset per application)
    DevCsr A_own, P, R;
    CsrDev A;                      // level operator (level 0 aliases the solver's matrix)
    DeviceBuffer<double> dinv;     // chebyshev "M" = inverted diagonal (scale = true)
    DeviceBuffer<double> dinv_blk; // block_size > 1: inverted diagonal blocks instead
    DeviceBuffer<double> f, u;     // rhs / solution of this level (levels > 0)
    DeviceBuffer<double> t, p, xb; // residual, chebyshev direction, ping-pong iterate
    // symbolic data for the numeric refresh (scalar path): aggregate map, R<-P entry map, pattern of A P
    DeviceBuffer<int> id, r_from_p;
    DevCsr AP;
    // locality renumbering (amg_renumber.hip): perm[i] = row of this level's operator that row i of the setup's
    // (the oracle's, AMGCL's) numbering became; empty: the level kept its numbering
    DeviceBuffer<int> perm;
    bool renumbered = false;
    // block value types: block graph of A (pattern, values, strength flags) and the block pattern of P
    BlockGraph blk_own;
    BlockGraph *blk = &blk_own; // level 0 shares the solver's BSR copy when there is one
    bool blk_shared = false;
    // block_size 3, operator stored with full node blocks: the block patterns of A P, R = P^T (+ where block p of R sits in
    // P's block values) and R (A P), kept for the numeric products on blocks (amg_bspgemm.hip)
    DeviceBuffer<int> apb_ptr, apb_col, rb_ptr, rb_col, rb_map, acb_ptr, acb_col;
    bool bspgemm = false;
    // "amg.direct_coarse": dense inverse of the coarsest operator (amg_relax.hip) -- the level is solved, not relaxed
    DeviceBuffer<double> cinv, cinv_work;
    bool direct = false;
    bool blk_current = false;   // *blk holds the values of THIS setup / refresh (set where they are filled, cleared when a new one starts)
    DeviceBuffer<float> A0_val32; // level 0 under "amg.matrix_fp32": single-precision copy of the solver's values
    DeviceBuffer<float> bsr_val32; // ... and of the 3x3-block copy; the cycle then multiplies through bsr3_cycle
    DeviceBuffer<float> A_blk32, P_blk32, R_blk32; // round 6: ... and of the block copies of A_l (l >= 1), P_l, R_l (40 B per block)
    Bsr3Dev bsr3_cycle;
    // block_size 3, "amg.block_levels": the operators of the cycle as 3x3 blocks (76 B and 3 gathers per block instead of
    // 108 B and 9): A_l of levels >= 1 (blk_own), P_l, R_l -- patterns once per hierarchy, values at every setup / refresh
    BlockGraph P_blk, R_blk;
    Bsr3Dev A_bsr, P_bsr, R_bsr;
    bool blk_own_built = false, P_blk_built = false, R_blk_built = false;
    bool aggregated_on_device = false;
    bool smoother_enqueued = false; // first setup only: already queued under a host sweep
    // "amg.refresh_power_iters" >= 0: the unit-norm iterate the last power iteration ended with, kept for the next refresh
    DeviceBuffer<double> pw;
    bool pw_valid = false;
    bool jacobi_like = false; // the level's smoother is one diagonally scaled residual step (amg.relax_type damped_jacobi / spai0)
    // round 6, amg.relax_type gauss_seidel (3) / ilu0 (4): ordered sweeps (amg_sweep.hip) over the level's (block) rows
    int sweeps = 0;
    DeviceBuffer<double> sw_lu, sw_work, sw_dinv; // ilu0: the factors on the level's pattern, the inverted pivots
    DeviceBuffer<int> sw_ctrl;
    bool smoother_is_coarsest = false; // set by whoever knows that this level is the hierarchy's last (direct_coarse skips its smoother)
    DeviceBuffer<int> pbptr, pbcol;
    DeviceBuffer<double> pbval;
    int64_t pbnnz = 0;
    double rho = 0, d = 0, c = 0;
    int n = 0;
    // scalar path, eps_strong = 0: which stored entries of this level's operator were nonzero when the
    // patterns of P / A P / R A P were derived from it (the strength graph is "stored value != 0")
    unsigned long long nz_hash = 0;
    Launch L; // grids fitted to this level's size
    // 1 / ||first n / bs draws of the random stream|| (power-iteration start vector), cached for the refresh
    double b0_scale = 0;
    int b0_n = -1, b0_bs = 0;
};

struct AmgHierarchy::Impl {
    std::vector<std::unique_ptr<Level>> lv;
    AmgParams prm;
    DeviceBuffer<double> partials; // 2 x kMaxPartials
    PinnedBuffer<double> host2;
    DeviceBuffer<unsigned long long> hash_dev;
    DeviceBuffer<unsigned long long> nz_hash_dev; // one slot per level
    PinnedBuffer<unsigned long long> nz_hash_host;
    // start vector of the power iterations: the std::mt19937(0) stream of amgcl::backend::spectral_radius,
    // drawn once by a side thread (it overlaps the first strength graph) and kept on the device; a level of
    // n rows uses the first n / bs draws, scaled to unit norm
    std::unique_ptr<double[]> rng_host; // plain array: no zero fill of up to a gigabyte
    size_t rng_host_count = 0, rng_host_cap = 0; // valid draws / capacity of rng_host
    DeviceBuffer<double> rng_dev;
    size_t rng_dev_count = 0;
    double rng_norm0 = 0;
    size_t rng_norm0_count = 0;
    int rng_norm0_bs = 0;
    std::future<void> rng_job;
    std::future<void> rng_free_job; // (the previous host copy going back to the allocator: nobody waits for it but the next one)
    PinnedBuffer<double> rho_host; // spectral radius of every level, written by async copies
    DeviceBuffer<int> bad_flags;
    // round 5, "amg.overlap_smoothers": the smoothers' power iterations (20 bandwidth-bound products per level) run on a
    // second stream beside the latency-bound work of the main one -- the aggregation sweep of a first setup, the Galerkin
    // chain of a refresh.  They read the level's operator (final by then) and write only the level's own smoother state and
    // work vectors, with their own partial-sum scratch; fork = an event on the main stream the side stream waits for, join =
    // the reverse, before anything reads the radii or touches the level vectors again.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DeviceBuffer<double> partials_side;
    PinnedBuffer<int> bad_host; // singular-diagonal-block flags of the block smoothers, read after the join
    int forks = 0;              // side-stream enqueues since the last join
    bool in_refresh = false;    // smoother_enqueue is called from refresh_numeric ("amg.refresh_power_iters")
    ~Impl()
    {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
    // device-side setup: scratch of the symbolic kernels, strength graph, diagonal
    SymbolicScratch sym;
    AggregateScratch agg;
    DeviceBuffer<int> sptr, scol;
    DeviceBuffer<int> bp_ap, bc_ap, bp_r, bc_r, bmap_r, bp_c, bc_c; // block patterns of A P, R, R (A P) (block value types)
    DeviceBuffer<double> dia;
    bool symbolic_valid = false, reused = false;
    unsigned long long pattern_hash = 0;
    // shards, "amg.dist_global": the hierarchy is the GLOBAL one (built by every rank from the gathered matrix, so
    // it is the single-device hierarchy, aggregates and all); level 0 is applied on the shard -- local rows of A
    // (halo exchange before every product), local rows of P_0, their transpose as the local part of R_0 followed by
    // one all-reduce of the level-1 right-hand side -- and levels >= 1 run replicated on every rank
    struct DistTop {
        bool on = false;
        int row0 = 0, n_loc = 0, n1 = 0;
        DevCsr P_loc, R_loc;                      // n_loc x n1 and its transpose n1 x n_loc
        DeviceBuffer<int> r_from_p;
        DeviceBuffer<double> x_ext, xb_ext, t, p; // level-0 work vectors (the iterates carry the halo tail)
    } top;
    int pattern_n = 0;
    int64_t pattern_nnz = 0;
};

AmgHierarchy::AmgHierarchy() : impl(new Impl()) {}
AmgHierarchy::~AmgHierarchy() = default;
int AmgHierarchy::levels() const { return (int)impl->lv.size(); }
bool AmgHierarchy::last_setup_reused() const { return impl->reused; }

constexpr int kMaxLevelSlots = 64;

// draws `count` values of the stream on a side thread and ships them to the device.  The stream is a fixed sequence
// (mt19937, seed 0): a host copy kept from an earlier setup (drop_rng_host) already holds its first rng_host_count values
static void start_rng(AmgHierarchy::Impl &I, size_t count, int bs, int device, hipStream_t main_stream)
{
    if (I.rng_job.valid()) I.rng_job.get();
    const bool upload = I.rng_dev_count < count;
    if (upload) {
        I.rng_dev.ensure(count);
        I.rng_dev_count = 0;
        // (round-4 advice) the block may come out of the handle's cache, where blocks wait whose last readers are still
        // queued on the handle's stream; the upload below runs on a stream of its own: let the handle's stream drain first
        PS_HIP_CHECK(hipStreamSynchronize(main_stream));
    }
    I.rng_job = std::async(std::launch::async, [&I, count, bs, device, upload] {
        if (I.rng_host_count < count) {
            if (I.rng_host_cap < count) {
                I.rng_host.reset(new double[count]);
                I.rng_host_cap = count;
            }
            Mt19937 rng(0);
            rng.fill(I.rng_host.get(), count);
            I.rng_host_count = count;
        }
        double norm = 0.0;
        for (size_t k = 0; k < count; ++k) norm += bs * I.rng_host[k] * I.rng_host[k];
        I.rng_norm0 = norm;
        I.rng_norm0_count = count;
        I.rng_norm0_bs = bs;
        if (upload) {
            PS_HIP_CHECK(hipSetDevice(device));
            hipStream_t st;
            PS_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            PS_HIP_CHECK(hipMemcpyAsync(I.rng_dev.ptr, I.rng_host.get(), count * sizeof(double), hipMemcpyHostToDevice, st));
            PS_HIP_CHECK(hipStreamSynchronize(st));
            PS_HIP_CHECK(hipStreamDestroy(st));
            I.rng_dev_count = count;
        }
    });
}

static void finish_rng(AmgHierarchy::Impl &I)
{
    if (I.rng_job.valid()) I.rng_job.get();
}

// What becomes of the host copy of the draws after a setup.  Handing 80 MB of touched pages (216^3; 134 MB at 256^3) back to
// the allocator costs 10 ms of munmap -- on the caller's thread a tenth of the whole setup, and on a side thread no less for a
// setup that follows at once (the process's mapping lock: its own allocations and hipFree calls wait).  Up to kRngHostKeep
// bytes the copy therefore stays with the handle: the sequence is fixed, so the next full setup finds its draws there and
// only sums their squares.  Larger copies go back on a side job (joined by the next drop or the future's destructor).
constexpr size_t kRngHostKeep = (size_t)256 << 20;

static void drop_rng_host(AmgHierarchy::Impl &I)
{
    finish_rng(I);
    if (!I.rng_host || I.rng_host_cap * sizeof(double) <= kRngHostKeep) return;
    I.rng_host_count = 0;
    I.rng_host_cap = 0;
    double *p = I.rng_host.release();
    if (I.rng_free_job.valid()) I.rng_free_job.get();
    I.rng_free_job = std::async(std::launch::async, [p] { delete[] p; });
}

// unit-norm scale of the first n / bs draws (sequential sum, as the oracle's)
static void level_b0_scale(AmgHierarchy::Impl &I, Level &lv, int bs)
{
    if (lv.b0_n == lv.n && lv.b0_bs == bs) return;
    finish_rng(I);
    const size_t draws = (size_t)(lv.n / bs);
    PS_REQUIRE(draws <= I.rng_host_count && draws <= I.rng_dev_count, PSOLVE_HIP_EINVAL, "AMG: random stream too short");
    double norm = 0.0;
    if (draws == I.rng_norm0_count && bs == I.rng_norm0_bs) {
        norm = I.rng_norm0;
    } else {
        for (size_t k = 0; k < draws; ++k) norm += bs * I.rng_host[k] * I.rng_host[k];
    }
    lv.b0_scale = 1.0 / std::sqrt(norm);
    lv.b0_n = lv.n;
    lv.b0_bs = bs;
}

// rho(D^-1 A) by `iters` power iterations (amgcl/backend/builtin.hpp spectral_radius<true>); with
// bs > 1 D is block diagonal, the start vector is constant per block and |<s_i, b_i>| is summed per block.
// Enqueues only: the radius lands in *rho_slot (pinned) when the stream gets there.
static void power_iteration_enqueue(const Launch &L, AmgHierarchy::Impl &I, Level &lv, int iters, int bs,
                                    double *rho_slot, double *partials, bool warm, bool keep)
{
    const int n = lv.n;
    // b0 lives in xb, b1 in t
    if (warm) { // "amg.refresh_power_iters": continue from where the previous factorize's iteration ended
        PS_HIP_CHECK(hipMemcpyAsync(lv.xb.ptr, lv.pw.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
    } else if (lv.renumbered) { // the same start vector on the same nodes: entry i of the stream belongs to the setup's row i
        launch_scale_expand(L, n, bs, lv.b0_scale, I.rng_dev.ptr, lv.t.ptr);
        launch_permute_f64(L, n, lv.t.ptr, lv.perm.ptr, lv.xb.ptr);
    } else {
        launch_scale_expand(L, n, bs, lv.b0_scale, I.rng_dev.ptr, lv.xb.ptr);
    }
    SpmvExtra ex;
    ex.dinv = lv.dinv.ptr;
    ex.partials2 = partials + kMaxPartials;
    const int np = bs > 1 ? L.grid : L.spmv_grid;
    for (int it = 0; it < iters; ++it) {
        if (bs > 1) {
            launch_spmv(L, lv.A, SPMV_PLAIN, lv.xb.ptr, nullptr, lv.t.ptr, nullptr, nullptr);
            launch_block_power(L, n, bs, lv.dinv_blk.ptr, lv.t.ptr, lv.xb.ptr, partials, partials + kMaxPartials);
        } else {
            launch_spmv(L, lv.A, SPMV_POWER, lv.xb.ptr, nullptr, lv.t.ptr, partials, nullptr, &ex);
        }
        if (it + 1 < iters) launch_scale_by_norm(L, n, partials, np, lv.t.ptr, lv.xb.ptr);
    }
    if (keep) { // the last iterate, normalised, for the next refresh
        lv.pw.ensure((size_t)n + 2);
        launch_scale_by_norm(L, n, partials, np, lv.t.ptr, lv.pw.ptr);
        lv.pw_valid = true;
    }
    launch_sum_partials(L, partials + kMaxPartials, np, kMaxPartials, partials, 1);
    PS_HIP_CHECK(hipMemcpyAsync(rho_slot, partials, sizeof(double), hipMemcpyDeviceToHost, L.stream));
}

static double device_gershgorin(const Launch &L, AmgHierarchy::Impl &I, const CsrDev &A)
{
    launch_gershgorin(L, A, I.partials.ptr);
    std::vector<double> h((size_t)L.grid);
    PS_HIP_CHECK(hipMemcpyAsync(h.data(), I.partials.ptr, h.size() * sizeof(double), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    double m = 0.0;
    for (double v : h) m = std::max(m, v);
    return m;
}

static void setup_smoother(Context &ctx, const Launch &L, AmgHierarchy::Impl &I, Level &lv, int slot);
static void smoother_enqueue(Context &ctx, const Launch &L, AmgHierarchy::Impl &I, Level &lv, int slot, bool on_side = false);
static bool smoother_fork(Context &ctx, const Launch &Lmain, AmgHierarchy::Impl &I, Level &lv, int slot);
static void coarse_solver_setup(Context &ctx, const Launch &L, AmgHierarchy::Impl &I);
static void smoothers_join(Context &ctx, const Launch &Lmain, AmgHierarchy::Impl &I);
static void smoother_finish(AmgHierarchy::Impl &I, Level &lv, int slot);
static SweepView sweep_view(const Level &lv, int bs);
static void level_workspace(Level &lv, bool coarse)
{
    const size_t n = (size_t)lv.n;
    lv.t.ensure(n + 2);
    lv.p.ensure(n + 2);
    lv.xb.ensure(n + 2);
    if (coarse) {
        lv.f.ensure(n + 2);
        lv.u.ensure(n + 2);
    }
}

// first factorize (or a new pattern): hierarchy on the host (amg_setup.cpp), uploaded level by level
static void full_setup(Context &ctx, const Launch &L, AmgHierarchy::Impl &I, const CsrDev &A)
{
    const AmgParams &prm = I.prm;
    hipStream_t s = L.stream;
    I.lv.clear();
    HostCsr H;
    H.nrows = H.ncols = A.n;
    H.ptr.resize((size_t)A.n + 1);
    H.col.resize((size_t)A.nnz);
    H.val.resize((size_t)A.nnz);
    PS_HIP_CHECK(hipMemcpyAsync(H.ptr.data(), A.rowptr, ((size_t)A.n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipMemcpyAsync(H.col.data(), A.col, (size_t)A.nnz * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipMemcpyAsync(H.val.data(), A.val, (size_t)A.nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const bool timing = std::getenv("PSOLVE_TIMING") != nullptr;
    double tt = wall_seconds();
    if (timing) std::fprintf(stderr, "[psolve timing] amg D2H of the fine matrix\n");
    std::vector<HostLevel> hl = build_hierarchy(std::move(H), prm);
    if (timing) {
        std::fprintf(stderr, "[psolve timing] amg host hierarchy total %.3f s\n", wall_seconds() - tt);
        tt = wall_seconds();
    }

    for (size_t l = 0; l < hl.size(); ++l) {
        std::unique_ptr<Level> lv(new Level());
        HostLevel &h = hl[l];
        lv->n = (int)h.A.nrows;
        if (l == 0) {
            lv->A = A; // no second copy of the fine matrix
        } else {
            lv->A_own.upload(h.A, s);
            lv->A = lv->A_own.view;
        }
        if (h.P.nrows > 0) {
            lv->P.upload(h.P, s);
            lv->R.upload(h.R, s);
            if (!h.id.empty()) { // symbolic data for refresh_numeric
                lv->id.ensure(h.id.size());
                lv->r_from_p.ensure(h.r_from_p.size() + 1);
                PS_HIP_CHECK(hipMemcpyAsync(lv->id.ptr, h.id.data(), h.id.size() * sizeof(int), hipMemcpyHostToDevice, s));
                PS_HIP_CHECK(hipMemcpyAsync(lv->r_from_p.ptr, h.r_from_p.data(), h.r_from_p.size() * sizeof(int),
                                            hipMemcpyHostToDevice, s));
                lv->AP.upload(h.AP, s);
            }
        }
        h = HostLevel(); // free host memory as we go
        level_workspace(*lv, l > 0);
        lv->smoother_is_coarsest = l + 1 == hl.size();
        setup_smoother(ctx, L, I, *lv, (int)l);
        I.lv.push_back(std::move(lv));
    }
    coarse_solver_setup(ctx, L, I);
    PS_HIP_CHECK(hipStreamSynchronize(s));
    if (timing) std::fprintf(stderr, "[psolve timing] amg uploads + smoother setup %.3f s\n", wall_seconds() - tt);
}

// "amg.matrix_fp32": the operators of the cycle (A_l, P_l, R_l) stream single-precision VALUES (8 B instead
// of 12 B per nonzero); vectors, products and sums stay double, and PCG's own product uses the original
// matrix -- the preconditioner is a slightly perturbed, still fixed and symmetric, linear operator.
// The smoothers' spectral radii were computed with the double-precision operators before this.
static void apply_matrix_precision(const Launch &L, AmgHierarchy::Impl &I)
{
    const bool on = I.prm.matrix_fp32 != 0;
    for (size_t l = 0; l < I.lv.size(); ++l) {
        Level &lv = *I.lv[l];
        if (l == 0) {
            lv.A.val32 = nullptr;
            if (on && lv.A.nnz > 0) {
                lv.A0_val32.ensure((size_t)lv.A.nnz + 4);
                launch_to_f32(L, lv.A.nnz, lv.A.val, lv.A0_val32.ptr);
                lv.A.val32 = lv.A0_val32.ptr;
                if (lv.A.bsr3) { // the cycle's own view of the block copy; PCG keeps multiplying by the double one
                    lv.bsr3_cycle = *lv.A.bsr3;
                    lv.bsr_val32.ensure((size_t)lv.bsr3_cycle.nnzb * 9 + 4);
                    launch_to_f32(L, lv.bsr3_cycle.nnzb * 9, lv.bsr3_cycle.val, lv.bsr_val32.ptr);
                    lv.bsr3_cycle.val32 = lv.bsr_val32.ptr;
                    lv.A.bsr3 = &lv.bsr3_cycle;
                }
            }
        } else {
            lv.A_own.set_fp32(L, on);
            lv.A_own.set_sell(L, I.prm.sell, I.sym);
            lv.A_own.set_col16(L, I.prm.col16 != 0);
            lv.A_own.set_row_blocks(L);
            lv.A = lv.A_own.view;
        }
        if (lv.P.view.n > 0) {
            lv.P.set_fp32(L, on);
            lv.R.set_fp32(L, on);
            lv.P.set_sell(L, I.prm.sell, I.sym);
            lv.R.set_sell(L, I.prm.sell, I.sym);
            lv.P.set_col16(L, I.prm.col16 != 0);
            lv.R.set_col16(L, I.prm.col16 != 0);
            lv.R.set_row_blocks(L);
        }
    }
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
}

// block_size 3 ("amg.block_levels", AMGCL_Block<3>'s value type end to end, AMGCL.cpp:243-302): every operator of the
// cycle multiplies through a 3x3-block copy -- A_l of the levels >= 1, the prolongations and the restrictions, which
// consist of full blocks by construction.  The scalar CSR arrays stay what the setup / refresh kernels compute on; the
// block copies are (re)filled from them here, after the smoothers (whose power iterations ran on the CSR arrays).
static Bsr3Dev bsr3_view(const BlockGraph &G)
{
    Bsr3Dev B;
    B.nb = G.nb;
    B.nnzb = G.nnzb;
    B.rowptr = G.ptr.ptr;
    B.col = G.col.ptr;
    B.val = G.val.ptr;
    B.brows_per_group = bsr3_brows_per_group((double)G.nnzb / (double)std::max(1, G.nb));
    return B;
}

static void attach_block_copies(const Launch &Lbase, AmgHierarchy::Impl &I)
{
    // (round 6: with "amg.matrix_fp32" the block copies stay and get single-precision values -- the LDS-DMA block kernel
    // streams either; until round 5 the option fell back to the scalar CSR float kernels for every operator but A_0)
    const bool on = I.prm.block_size == 3 && I.prm.block_levels != 0;
    const bool f32 = I.prm.matrix_fp32 != 0;
    auto with_f32 = [&](const Launch &L, Bsr3Dev &B, DeviceBuffer<float> &buf) {
        if (!f32 || B.nnzb <= 0) return;
        buf.ensure((size_t)B.nnzb * 9 + 4);
        launch_to_f32(L, B.nnzb * 9, B.val, buf.ptr);
        B.val32 = buf.ptr;
    };
    for (size_t l = 0; l < I.lv.size(); ++l) {
        Level &lv = *I.lv[l];
        if (l > 0) {
            lv.A_own.view.bsr3 = nullptr;
            lv.A.bsr3 = nullptr;
        }
        lv.P.view.bsr3 = nullptr;
        lv.R.view.bsr3 = nullptr;
        if (!on) continue;
        Launch L = fit_setup_launch(Lbase, lv.n, lv.A.nnz, lv.A.rows_per_block);
        L.stream = Lbase.stream;
        if (l > 0 && lv.n % 3 == 0 && lv.A_own.view.nnz > 0) {
            const CsrDev A = lv.A_own.view;
            if (!lv.blk_own_built) {
                device_block_graph(L, A, 3, lv.blk_own, I.sym);
                lv.blk_own_built = true;
            }
            if (!(lv.blk == &lv.blk_own && lv.blk_current)) device_block_values(L, A, lv.blk_own); // (the coarsest level: nobody filled it yet)
            lv.A_bsr = bsr3_view(lv.blk_own);
            with_f32(L, lv.A_bsr, lv.A_blk32);
            lv.A_own.view.bsr3 = &lv.A_bsr;
            lv.A = lv.A_own.view;
        }
        if (lv.P.view.n > 0 && lv.P.view.n % 3 == 0 && lv.P.view.n_ext % 3 == 0 && lv.P.view.nnz > 0) {
            if (!lv.P_blk_built) {
                device_block_graph(L, lv.P.view, 3, lv.P_blk, I.sym);
                device_block_graph(L, lv.R.view, 3, lv.R_blk, I.sym);
                lv.P_blk_built = lv.R_blk_built = true;
            }
            device_block_values(L, lv.P.view, lv.P_blk);
            device_block_values(L, lv.R.view, lv.R_blk);
            lv.P_bsr = bsr3_view(lv.P_blk);
            lv.R_bsr = bsr3_view(lv.R_blk);
            with_f32(L, lv.P_bsr, lv.P_blk32);
            with_f32(L, lv.R_bsr, lv.R_blk32);
            lv.P.view.bsr3 = &lv.P_bsr;
            lv.R.view.bsr3 = &lv.R_bsr;
        }
    }
    PS_HIP_CHECK(hipStreamSynchronize(Lbase.stream));
}

// Locality renumbering of the levels >= 1 (amg_renumber.hip), after the hierarchy has been built in the setup's own
// numbering (the aggregates therefore ARE the sequential sweep's): level l is ordered by (new id of the node's
// aggregate on level l + 1, old id), from the coarsest level down; then A_l, P_l, the pattern of A_l P_l and the
// aggregate maps are rewritten in the new numbering with sorted columns, R_l is transposed again.
static void renumber_levels(Context &ctx, const Launch &Lmax, AmgHierarchy::Impl &I)
{
    const AmgParams &prm = I.prm;
    const int nl = (int)I.lv.size();
    if (!prm.renumber || prm.block_size > 1 || nl < 3) return;
    hipStream_t s = Lmax.stream;
    DeviceBuffer<int> w_key, w_iota, w_ptr, w_order, w_map, t_ptr, t_col, t_id;
    DeviceBuffer<double> t_val;
    // 1. the orders, coarsest level first (the coarsest level itself has no aggregates: it keeps its numbering)
    for (int l = nl - 2; l >= 1; --l) {
        Level &lv = *I.lv[l];
        Level &up = *I.lv[l + 1];
        lv.renumbered = false;
        if (lv.n < prm.renumber_min_rows) continue;
        Launch L = fit_launch(ctx.launch_max(), lv.n, lv.A.rows_per_block);
        L.stream = s;
        device_order_by_parent(L, lv.n, lv.id.ptr, up.renumbered ? up.perm.ptr : nullptr, up.n, lv.perm, I.sym, w_key,
                               w_iota, w_ptr, w_order, w_map);
        lv.renumbered = true;
        lv.smoother_enqueued = false; // a smoother queued under a host sweep used the setup's numbering: redo it
    }
    // 2. the operators
    for (int l = 0; l + 1 < nl; ++l) {
        Level &lv = *I.lv[l];
        Level &up = *I.lv[l + 1];
        const int *rp = lv.renumbered ? lv.perm.ptr : nullptr, *cp = up.renumbered ? up.perm.ptr : nullptr;
        if (!rp && !cp) continue;
        Launch L = fit_launch(ctx.launch_max(), lv.n, lv.A.rows_per_block);
        L.stream = s;
        if (rp) { // A_l (l >= 1: level 0 is the caller's matrix and never renumbered)
            const CsrDev A = lv.A_own.view;
            device_permute_csr(L, A.n, A.nnz, A.rowptr, A.col, A.val, rp, rp, t_ptr, t_col, &t_val, I.sym);
            lv.A_own.ptr.swap(t_ptr);
            lv.A_own.col.swap(t_col);
            lv.A_own.val.swap(t_val);
            lv.A_own.set_view(A.n, A.n_ext, A.nnz);
            lv.A = lv.A_own.view;
            PS_HIP_CHECK(hipMemsetAsync(I.nz_hash_dev.ptr + l, 0, sizeof(unsigned long long), s));
            launch_hash_nonzero(L, lv.A.nnz, lv.A.val, I.nz_hash_dev.ptr + l); // the hash is over entry positions
        }
        { // P_l, then R_l = P_l^T
            const CsrDev P = lv.P.view;
            device_permute_csr(L, P.n, P.nnz, P.rowptr, P.col, P.val, rp, cp, t_ptr, t_col, &t_val, I.sym);
            lv.P.ptr.swap(t_ptr);
            lv.P.col.swap(t_col);
            lv.P.val.swap(t_val);
            lv.P.set_view(P.n, P.n_ext, P.nnz);
            device_transpose_pattern(L, P.n, P.n_ext, lv.P.ptr.ptr, lv.P.col.ptr, P.nnz, lv.R.ptr, lv.R.col, lv.r_from_p, I.sym);
            lv.R.set_view(P.n_ext, P.n, P.nnz);
            launch_gather(L, (int)P.nnz, lv.r_from_p.ptr, lv.P.val.ptr, lv.R.val.ptr);
        }
        { // pattern of A_l P_l (its values are scratch of the numeric refresh)
            const CsrDev AP = lv.AP.view;
            device_permute_csr(L, AP.n, AP.nnz, AP.rowptr, AP.col, nullptr, rp, cp, t_ptr, t_col, nullptr, I.sym);
            lv.AP.ptr.swap(t_ptr);
            lv.AP.col.swap(t_col);
            lv.AP.set_view(AP.n, AP.n_ext, AP.nnz);
        }
        // aggregate map: row i -> perm_l, value -> perm_{l+1}
        t_id.ensure((size_t)lv.n + 1);
        launch_relabel_ids(L, lv.n, lv.id.ptr, rp, cp, t_id.ptr);
        lv.id.swap(t_id);
    }
    PS_HIP_CHECK(hipStreamSynchronize(s));
}


struct SideJoinGuard { // every way out of a setup / refresh -- a refused refresh, an exception -- lets the two streams meet
    Context &ctx;
    const Launch &L;
    AmgHierarchy::Impl &I;
    ~SideJoinGuard()
    {
        try {
            smoothers_join(ctx, L, I);
        } catch (...) {
        }
    }
};

// first factorize (or a new pattern), scalar systems: the hierarchy is built where the matrix lives.
// Per level: strength graph (kernel) -> D2H of that graph only -> greedy aggregation sweep (host,
// sequential by definition) -> H2D of the aggregate map -> patterns of P, R = P^T, A P, R (A P) by the
// row-set kernels of amg_symbolic.hip -> numbers by the same kernels the numeric refresh uses.  The
// result equals build_hierarchy()'s bit for bit (tests/test_gpu_amg.py).
static void device_full_setup(Context &ctx, const Launch &Lmax, AmgHierarchy::Impl &I, const CsrDev &A0)
{
    const AmgParams &prm = I.prm;
    hipStream_t s = Lmax.stream;
    const bool timing = std::getenv("PSOLVE_TIMING") != nullptr;
    double t0 = wall_seconds();
    auto lap = [&](const char *what, int64_t rows) {
        if (!timing) return;
        PS_HIP_CHECK(hipStreamSynchronize(s));
        const double t1 = wall_seconds();
        std::fprintf(stderr, "[psolve timing] amg device %-22s rows=%lld %.4f s\n", what, (long long)rows, t1 - t0);
        t0 = t1;
    };
    PS_REQUIRE(prm.sa_power_iters == 0, PSOLVE_HIP_EINVAL,
               "amg.sa_power_iters > 0 is not supported (AMGCL's default 0 = Gershgorin is)");
    SideJoinGuard join_guard{ctx, Lmax, I};
    I.lv.clear();
    I.nz_hash_dev.ensure(2 * kMaxLevelSlots);
    I.nz_hash_host.ensure(2 * kMaxLevelSlots);
    PS_HIP_CHECK(hipMemsetAsync(I.nz_hash_dev.ptr, 0, 2 * kMaxLevelSlots * sizeof(unsigned long long), s));
    CsrDev A = A0;
    double eps = prm.eps_strong;
    // host copies of the strength graph: plain arrays (no zero fill), sized by the finest level, reused
    std::unique_ptr<int32_t[]> h_sptr, h_scol;
    size_t cap_sptr = 0, cap_scol = 0;
    std::vector<int32_t> h_id;
    std::unique_ptr<Level> pending; // level whose operator is A
    pending.reset(new Level());
    pending->A = A;
    pending->n = A.n;
    DeviceBuffer<int> id0;
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    while (A.n > prm.coarse_enough) {
        Level &lv = *pending;
        if ((int)I.lv.size() + 1 >= prm.max_levels) break;
        const int slot = (int)I.lv.size();
        Launch L = fit_setup_launch(ctx.launch_max(), A.n, A.nnz, A.rows_per_block);
        L.stream = s;
        const int ng = A.n / bs; // nodes of the strength graph (block rows when bs > 1)
        // strength graph + start state of the sweep
        id0.ensure((size_t)ng);
        int64_t snnz;
        if (bs > 1) {
            BlockGraph *shared = slot == 0 ? ctx.shared_block_graph(bs) : nullptr;
            if (shared) { // built (pattern + values) by the solver's factorize for its BSR products
                lv.blk = shared;
                lv.blk_shared = true;
            } else {
                device_block_graph(L, A, bs, *lv.blk, I.sym);
                device_block_values(L, A, *lv.blk);
                lv.blk_own_built = true;
            }
            lv.blk_current = true;
            snnz = device_block_strength_graph(L, *lv.blk, eps, I.sptr, I.scol, id0.ptr, I.sym);
        } else {
            I.dia.ensure((size_t)A.n);
            launch_extract_diagonal(L, A, I.dia.ptr);
            snnz = device_strength_graph(L, A, eps, I.dia.ptr, I.sptr, I.scol, id0.ptr, I.sym);
            launch_hash_nonzero(L, A.nnz, A.val, I.nz_hash_dev.ptr + slot); // read back with the smoothers' radii
        }
        lap("strength graph", A.n);
        // this level's smoother beside its aggregation sweep (round 5): its operator is final
        if (!lv.smoother_enqueued && prm.overlap_smoothers) {
            level_workspace(lv, slot > 0);
            smoother_fork(ctx, Lmax, I, lv, slot);
        }
        // aggregation: on the device when the graph qualifies (symmetric, sorted, moderate dependency depth)
        lv.id.ensure((size_t)ng);
        int64_t nagg = -1;
        lv.aggregated_on_device = false;
        // ("parallel": a dozen synchronous rounds at any size -- small levels too, so that every level is the same algorithm)
        if (prm.device_aggregation && (ng >= prm.aggregation_min_rows || prm.aggregation >= 1)) { // (a small level is swept faster by the host)
            int rounds = 0;
            nagg = device_aggregate(L, ng, I.sptr.ptr, I.scol.ptr, id0.ptr, lv.id.ptr, prm.aggregation_max_rounds, I.agg,
                                    I.sym, &rounds, prm.aggregation == 2 ? 4 : (prm.aggregation == 1 ? 3 : (prm.aggregation_rounds ? 1 : 2)));
            if (timing)
                std::fprintf(stderr, "[psolve timing] amg device aggregation: %s after %d rounds\n",
                             nagg >= 0 ? "done" : "fell back to the host sweep", rounds);
            lap("aggregation (device)", A.n);
            lv.aggregated_on_device = nagg >= 0;
        }
        if (nagg < 0) {
            // host sweep ahead: this level's smoother (diagonal, power iterations) runs on the device meanwhile.
            // Otherwise the smoothers are enqueued at the end: the random start vector (drawn by a side thread)
            // then has the whole setup to arrive.
            if (!lv.smoother_enqueued) {
                level_workspace(lv, slot > 0);
                smoother_enqueue(ctx, Lmax, I, lv, slot);
                lv.smoother_enqueued = true;
            }
            if (cap_sptr < (size_t)ng + 1) {
                cap_sptr = (size_t)ng + 1;
                h_sptr.reset(new int32_t[cap_sptr]);
            }
            if (cap_scol < (size_t)snnz + 1) {
                cap_scol = (size_t)snnz + 1;
                h_scol.reset(new int32_t[cap_scol]);
            }
            h_id.resize((size_t)ng);
            PS_HIP_CHECK(hipMemcpyAsync(h_sptr.get(), I.sptr.ptr, ((size_t)ng + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
            if (snnz)
                PS_HIP_CHECK(hipMemcpyAsync(h_scol.get(), I.scol.ptr, (size_t)snnz * sizeof(int), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipMemcpyAsync(h_id.data(), id0.ptr, (size_t)ng * sizeof(int), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
            lap("graph D2H", A.n);
            nagg = aggregate_strength_graph(ng, h_sptr.get(), h_scol.get(), h_id, true, prm.aggregation);
            lap("aggregation sweep (host)", A.n);
            if (nagg > 0)
                PS_HIP_CHECK(hipMemcpyAsync(lv.id.ptr, h_id.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice, s));
        }
        const double eps_level = eps;
        eps *= 0.5;
        if (nagg == 0) break; // amgcl error::empty_level: the level is (block-)diagonal, it becomes the coarsest
        double omega = prm.sa_relax;
        const int nc = (int)nagg * bs; // coarse scalar size
        int64_t pnnz;
        if (bs > 1) {
            if (prm.coarsening == 1) {
                // amgcl::coarsening::aggregation: P = the tentative prolongation (one identity block per kept node)
                lv.pbnnz = device_tentative_prolongation(L, ng, lv.id.ptr, bs, lv.pbptr, lv.pbcol, &lv.pbval, I.sym);
            } else {
            // (level 0 on the solver's own block graph whose block rows come in kinds: the bound of one row per kind is the
            // bound of all rows)
            const Bsr3KindDev *bk = (lv.blk_shared && lv.blk == ctx.shared_block_graph(bs)) ? ctx.shared_block_kinds() : nullptr;
            omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / device_block_gershgorin(L, *lv.blk, I.partials.ptr, bk ? bk->krep : nullptr, bk ? bk->nk : 0)
                                                  : 2.0 / 3.0;
            // block pattern of P = strength graph x aggregate map, block values, then scalar CSR with full blocks
            lv.pbnnz = device_spgemm_symbolic(L, ng, I.sptr.ptr, I.scol.ptr, nullptr, lv.id.ptr, (int)nagg, lv.pbptr,
                                              lv.pbcol, I.sym);
            lv.pbval.ensure((size_t)lv.pbnnz * bs * bs + 4);
            launch_block_prolongation_values(L, *lv.blk, lv.id.ptr, omega, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr);
            }
            pnnz = lv.pbnnz * bs * bs;
            PS_REQUIRE(pnnz < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
            lv.P.ptr.ensure((size_t)A.n + 1);
            lv.P.col.ensure((size_t)pnnz + 4);
            lv.P.val.ensure((size_t)pnnz + 4);
            launch_expand_block_csr(L, ng, bs, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr, lv.P.ptr.ptr, lv.P.col.ptr,
                                    lv.P.val.ptr);
        } else if (prm.coarsening == 1) {
            pnnz = device_tentative_prolongation(L, A.n, lv.id.ptr, 1, lv.P.ptr, lv.P.col, &lv.P.val, I.sym);
        } else {
            omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / device_gershgorin(L, I, A) : 2.0 / 3.0;
            // P: pattern = strength graph x aggregate map, then the numbers
            pnnz = device_spgemm_symbolic(L, A.n, I.sptr.ptr, I.scol.ptr, nullptr, lv.id.ptr, (int)nagg, lv.P.ptr,
                                          lv.P.col, I.sym);
            lv.P.val.ensure((size_t)pnnz + 4);
            CsrMut P{A.n, lv.P.ptr.ptr, lv.P.col.ptr, lv.P.val.ptr};
            launch_prolongation_values(L, A, lv.id.ptr, omega, eps_level != 0.0 ? I.dia.ptr : nullptr, eps_level, P);
        }
        lv.P.set_view(A.n, nc, pnnz);
        lap("P", A.n);
        // R = P^T
        device_transpose_pattern(L, A.n, nc, lv.P.ptr.ptr, lv.P.col.ptr, pnnz, lv.R.ptr, lv.R.col, lv.r_from_p, I.sym);
        lv.R.val.ensure((size_t)pnnz + 4);
        lv.R.set_view(nc, A.n, pnnz);
        launch_gather(L, (int)pnnz, lv.r_from_p.ptr, lv.P.val.ptr, lv.R.val.ptr);
        lap("R = P^T", A.n);
        // A P.  Block value types: P (and so R) consists of full blocks, hence the pattern of row i of A P is the
        // union of the BLOCK rows of P over the nodes row i touches, expanded, and that of R (A P) the product of
        // the block patterns, expanded -- one ninth of the symbolic work for 3 x 3 blocks (the Q1 elasticity setup
        // spent 0.16 of its 0.37 s in the scalar row sets).  When the blocks of A are stored in full the rows of a
        // node agree and the block product does it all; otherwise (a caller's matrix with dropped zeros) the rows
        // of A are folded onto node columns first and keep their own patterns.
        const bool block_patterns = bs > 1;
        int64_t apnnz;
        if (block_patterns) {
            const int64_t apb = device_spgemm_symbolic(L, ng, lv.blk->ptr.ptr, lv.blk->col.ptr, lv.pbptr.ptr,
                                                       lv.pbcol.ptr, (int)nagg, I.bp_ap, I.bc_ap, I.sym);
            lv.AP.ptr.ensure((size_t)A.n + 1);
            // (round 4: the expanded block pattern also where the caller's matrix dropped zeros inside its node blocks --
            // Dirichlet rows after elimination, FEMSolver.cpp:136-161 --: the rows of a node then carry a few stored zeros
            // in A P, R (A P) is the product of the block patterns either way, and the numeric products can run on blocks)
            apnnz = apb * bs * bs;
            PS_REQUIRE(apnnz < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
            lv.AP.col.ensure((size_t)apnnz + 4);
            launch_expand_block_csr(L, ng, bs, I.bp_ap.ptr, I.bc_ap.ptr, nullptr, lv.AP.ptr.ptr, lv.AP.col.ptr, nullptr);
        } else {
            apnnz = device_spgemm_symbolic(L, A.n, A.rowptr, A.col, lv.P.ptr.ptr, lv.P.col.ptr, nc, lv.AP.ptr, lv.AP.col,
                                           I.sym);
        }
        lv.AP.val.ensure((size_t)apnnz + 4);
        lv.AP.set_view(A.n, nc, apnnz);
        CsrMut AP{A.n, lv.AP.ptr.ptr, lv.AP.col.ptr, lv.AP.val.ptr};
        // 3 x 3 blocks stored in full: the products run on the block patterns (amg_bspgemm.hip), same numbers
        lv.bspgemm = block_patterns && bs == 3 && lv.blk_current;
        if (lv.bspgemm)
            launch_bspgemm3_numeric(L, ng, I.bp_ap.ptr, I.bc_ap.ptr, lv.AP.val.ptr, lv.blk->ptr.ptr, lv.blk->col.ptr,
                                    lv.blk->val.ptr, nullptr, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr, false,
                                    (double)apnnz / 9.0 / std::max(1, ng));
        else
            launch_spgemm_numeric(L, AP, A, lv.P.view, (double)apnnz / std::max(1, A.n));
        lap("A P", A.n);
        // A_c = R (A P)
        std::unique_ptr<Level> nx(new Level());
        int64_t acnnz;
        if (block_patterns) {
            device_transpose_pattern(L, ng, (int)nagg, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbnnz, I.bp_r, I.bc_r, I.bmap_r, I.sym);
            const int64_t acb = device_spgemm_symbolic(L, (int)nagg, I.bp_r.ptr, I.bc_r.ptr, I.bp_ap.ptr, I.bc_ap.ptr,
                                                       (int)nagg, I.bp_c, I.bc_c, I.sym);
            acnnz = acb * bs * bs;
            PS_REQUIRE(acnnz < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
            nx->A_own.ptr.ensure((size_t)nc + 1);
            nx->A_own.col.ensure((size_t)acnnz + 4);
            launch_expand_block_csr(L, (int)nagg, bs, I.bp_c.ptr, I.bc_c.ptr, nullptr, nx->A_own.ptr.ptr,
                                    nx->A_own.col.ptr, nullptr);
        } else {
            acnnz = device_spgemm_symbolic(L, nc, lv.R.ptr.ptr, lv.R.col.ptr, lv.AP.ptr.ptr, lv.AP.col.ptr, nc,
                                           nx->A_own.ptr, nx->A_own.col, I.sym);
        }
        nx->A_own.val.ensure((size_t)acnnz + 4);
        nx->A_own.set_view(nc, nc, acnnz);
        CsrMut Ac{nc, nx->A_own.ptr.ptr, nx->A_own.col.ptr, nx->A_own.val.ptr};
        if (lv.bspgemm) {
            launch_bspgemm3_numeric(L, (int)nagg, I.bp_c.ptr, I.bc_c.ptr, nx->A_own.val.ptr, I.bp_r.ptr, I.bc_r.ptr,
                                    lv.pbval.ptr, I.bmap_r.ptr, I.bp_ap.ptr, I.bc_ap.ptr, lv.AP.val.ptr, true,
                                    (double)acnnz / 9.0 / std::max<double>(1, (double)nagg));
            // the block patterns stay with the level for the numeric refresh (the scratch arrays are made again below)
            lv.apb_ptr.swap(I.bp_ap);
            lv.apb_col.swap(I.bc_ap);
            lv.rb_ptr.swap(I.bp_r);
            lv.rb_col.swap(I.bc_r);
            lv.rb_map.swap(I.bmap_r);
            lv.acb_ptr.swap(I.bp_c);
            lv.acb_col.swap(I.bc_c);
        } else {
            launch_spgemm_numeric(L, Ac, lv.R.view, lv.AP.view, (double)acnnz / std::max(1, nc));
        }
        // amgcl/coarsening/aggregation.hpp: the Galerkin operator of the unsmoothed prolongation is scaled by 1 / over_interp
        if (prm.coarsening == 1) launch_scale_values(L, acnnz, over_interp_scale(prm.over_interp, bs), nx->A_own.val.ptr);
        lap("R (A P)", A.n);
        nx->A = nx->A_own.view;
        nx->n = nc;
        I.lv.push_back(std::move(pending));
        pending = std::move(nx);
        A = pending->A;
    }
    I.lv.push_back(std::move(pending));
    I.lv.back()->smoother_is_coarsest = true;
    if (!I.lv.back()->smoother_enqueued && prm.overlap_smoothers && I.lv.size() > 1) { // the coarsest level's, beside the others'
        level_workspace(*I.lv.back(), true);
        smoother_fork(ctx, Lmax, I, *I.lv.back(), (int)I.lv.size() - 1);
    }
    smoothers_join(ctx, Lmax, I);
    renumber_levels(ctx, Lmax, I);
    lap("renumbering", A0.n);
    for (size_t l = 0; l < I.lv.size(); ++l) {
        Level &lv = *I.lv[l];
        if (lv.smoother_enqueued) continue;
        level_workspace(lv, l > 0);
        smoother_enqueue(ctx, Lmax, I, lv, (int)l);
        lv.smoother_enqueued = true;
    }
    PS_HIP_CHECK(hipMemcpyAsync(I.nz_hash_host.ptr, I.nz_hash_dev.ptr, kMaxLevelSlots * sizeof(unsigned long long),
                                hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t l = 0; l < I.lv.size(); ++l) {
        smoother_finish(I, *I.lv[l], (int)l);
        I.lv[l]->nz_hash = I.nz_hash_host.ptr[l];
    }
    lap("smoothers", A0.n);
    coarse_solver_setup(ctx, Lmax, I);
    if (prm.direct_coarse) lap("coarsest level inverted", A0.n);
    // transient buffers go back to the allocator
    I.sptr.release();
    I.scol.release();
    I.dia.release();
    I.bp_ap.release();
    I.bc_ap.release();
    I.bp_r.release();
    I.bc_r.release();
    I.bmap_r.release();
    I.bp_c.release();
    I.bc_c.release();
    I.sym.tmp.release();
    I.sym.table.release();
    I.sym.cand.release();
    I.sym.tier.release();
    I.sym.cursor.release();
    I.agg.ints.release();
    I.agg.tptr.release();
    I.agg.tcol.release();
    I.agg.tmap.release();
    lap("transients released", A0.n);
}

// same pattern, new values: omega, P, R, A P and R A P of every level recomputed by kernels.  Returns
// false when the strength flags of a block level changed with the new values (the aggregates would differ):
// the caller then rebuilds the hierarchy.
static bool refresh_numeric(Context &ctx, const Launch &Lmain, AmgHierarchy::Impl &I, const CsrDev &A)
{
    const AmgParams &prm = I.prm;
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    I.lv[0]->A = A;
    for (auto &lvp : I.lv) lvp->blk_current = false;
    unsigned long long *nzh = I.nz_hash_dev.ptr + kMaxLevelSlots; // this refresh's flags, level by level
    if (bs == 1) PS_HIP_CHECK(hipMemsetAsync(nzh, 0, kMaxLevelSlots * sizeof(unsigned long long), Lmain.stream));
    for (auto &lvp : I.lv) {
        lvp->smoother_enqueued = false;
        lvp->smoother_is_coarsest = lvp.get() == I.lv.back().get();
    }
    SideJoinGuard join_guard{ctx, Lmain, I};
    struct RefreshFlag {
        bool &f;
        explicit RefreshFlag(bool &x) : f(x) { f = true; }
        ~RefreshFlag() { f = false; }
    } refresh_flag{I.in_refresh};
    for (size_t l = 0; l + 1 < I.lv.size(); ++l) {
        Level &lv = *I.lv[l];
        Level &nx = *I.lv[l + 1];
        // the grids of a first setup (rows x lanes), not the vector kernels' grid of the caller: with PCG's own vector kernels on
        // two workgroups per CU (round 6) the level-0 prolongation values of configs[2] ran on 512 workgroups, 3.6 ms against 2.0
        Launch L = fit_setup_launch(ctx.launch_max(), lv.n, lv.A.nnz, lv.A.rows_per_block);
        L.stream = Lmain.stream;
        double omega = prm.sa_relax;
        if (bs > 1) {
            BlockGraph &G = *lv.blk;
            if (lv.blk_shared) {
                if (ctx.shared_block_graph(bs) != lv.blk) return false; // use_bsr3 switched off meanwhile: rebuild
            } else {
                device_block_values(L, lv.A, G); // (the shared copy got its values in the solver's factorize)
            }
            lv.blk_current = true;
            smoother_fork(ctx, L, I, lv, (int)l); // this level's smoother beside its Galerkin chain (round 5)
            // strength on the new values must select the same blocks (eps = 0: tr(A_ij A_ij) > 0)
            I.sym.tier.ensure((size_t)G.nnzb + 4);
            I.sym.cand.ensure((size_t)G.nb + 1);
            device_block_strong_flags(L, G, 0.0, I.sym.tier.ptr, I.sym.cand.ptr);
            {
                const long long changed = device_block_flag_changes(L, G.nnzb, I.sym.tier.ptr, G.strong.ptr, I.sym);
                if (L.lab.verbose && changed) fprintf(stderr, "[psolve lab] refresh: level %zu: %lld of %lld block strength flags changed\n", l, changed, (long long)G.nnzb);
                if (changed != 0) return false;
            }
            if (prm.coarsening == 0) { // (the tentative prolongation of the aggregation coarsening holds no numbers of A)
            // (level 0 on the solver's own block graph whose block rows come in kinds: the bound of one row per kind is the
            // bound of all rows)
            const Bsr3KindDev *bk = (lv.blk_shared && lv.blk == ctx.shared_block_graph(bs)) ? ctx.shared_block_kinds() : nullptr;
            omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / device_block_gershgorin(L, *lv.blk, I.partials.ptr, bk ? bk->krep : nullptr, bk ? bk->nk : 0)
                                                  : 2.0 / 3.0;
            launch_block_prolongation_values(L, *lv.blk, lv.id.ptr, omega, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr);
            launch_expand_block_csr(L, lv.blk->nb, bs, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr, nullptr, nullptr,
                                    lv.P.val.ptr);
            }
        } else {
            smoother_fork(ctx, L, I, lv, (int)l); // this level's smoother beside its Galerkin chain (round 5)
            // eps_strong = 0: the strength graph is "stored value != 0".  An entry that flipped between zero and
            // nonzero changes the graph, hence the aggregates and P's pattern: checked below, after the queue
            launch_hash_nonzero(L, lv.A.nnz, lv.A.val, nzh + l);
            if (prm.coarsening == 0) {
                omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / device_gershgorin(L, I, lv.A) : 2.0 / 3.0;
                CsrMut P{lv.P.view.n, lv.P.ptr.ptr, lv.P.col.ptr, lv.P.val.ptr};
                launch_prolongation_values(L, lv.A, lv.id.ptr, omega, nullptr, 0.0, P);
            }
        }
        if (prm.coarsening == 0) launch_gather(L, (int)lv.R.view.nnz, lv.r_from_p.ptr, lv.P.val.ptr, lv.R.val.ptr);
        CsrMut AP{lv.AP.view.n, lv.AP.ptr.ptr, lv.AP.col.ptr, lv.AP.val.ptr};
        CsrMut Ac{nx.A_own.view.n, nx.A_own.ptr.ptr, nx.A_own.col.ptr, nx.A_own.val.ptr};
        const bool block_products = bs == 3 && lv.bspgemm && lv.blk_current && lv.apb_ptr.ptr && lv.acb_ptr.ptr;
        // (round 5's kept product plans -- every entry of A P and R (A P) as the sum of its terms, amg_plan.hip -- measured slower
        // than these row-wise kernels, 28.0 -> 35.5 ms per refresh at 216^3, profiles/r05_refresh.md; removed in round 6)
        if (block_products) {
            launch_bspgemm3_numeric(L, lv.blk->nb, lv.apb_ptr.ptr, lv.apb_col.ptr, lv.AP.val.ptr, lv.blk->ptr.ptr,
                                    lv.blk->col.ptr, lv.blk->val.ptr, nullptr, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr, false,
                                    (double)lv.AP.view.nnz / 9.0 / std::max(1, lv.blk->nb));
            launch_bspgemm3_numeric(L, nx.A_own.view.n / 3, lv.acb_ptr.ptr, lv.acb_col.ptr, nx.A_own.val.ptr, lv.rb_ptr.ptr,
                                    lv.rb_col.ptr, lv.pbval.ptr, lv.rb_map.ptr, lv.apb_ptr.ptr, lv.apb_col.ptr,
                                    lv.AP.val.ptr, true, (double)nx.A_own.view.nnz / 9.0 / std::max(1, nx.A_own.view.n / 3));
        } else {
            launch_spgemm_numeric(L, AP, lv.A, lv.P.view, (double)lv.AP.view.nnz / std::max(1, lv.AP.view.n));
            launch_spgemm_numeric(L, Ac, lv.R.view, lv.AP.view, (double)nx.A_own.view.nnz / std::max(1, nx.A_own.view.n));
        }
        if (prm.coarsening == 1) launch_scale_values(L, nx.A_own.view.nnz, over_interp_scale(prm.over_interp, bs), nx.A_own.val.ptr);
    }
    if (I.lv.size() > 1 && !I.lv.back()->smoother_enqueued) smoother_fork(ctx, Lmain, I, *I.lv.back(), (int)I.lv.size() - 1);
    smoothers_join(ctx, Lmain, I);
    for (size_t l = 0; l < I.lv.size(); ++l)
        if (!I.lv[l]->smoother_enqueued) smoother_enqueue(ctx, Lmain, I, *I.lv[l], (int)l);
    if (bs == 1)
        PS_HIP_CHECK(hipMemcpyAsync(I.nz_hash_host.ptr + kMaxLevelSlots, nzh, kMaxLevelSlots * sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, Lmain.stream));
    PS_HIP_CHECK(hipStreamSynchronize(Lmain.stream));
    if (bs == 1)
        for (size_t l = 0; l + 1 < I.lv.size(); ++l)
            if (I.nz_hash_host.ptr[kMaxLevelSlots + l] != I.lv[l]->nz_hash) { // graph changed: rebuild
                if (Lmain.lab.verbose || std::getenv("PSOLVE_TIMING"))
                    std::fprintf(stderr, "[psolve] amg refresh: the nonzero pattern of level %zu's values changed (%llx -> %llx): full setup\n", l,
                                 I.lv[l]->nz_hash, I.nz_hash_host.ptr[kMaxLevelSlots + l]);
                return false;
            }
    for (size_t l = 0; l < I.lv.size(); ++l) smoother_finish(I, *I.lv[l], (int)l);
    coarse_solver_setup(ctx, Lmain, I);
    return true;
}

// chebyshev smoother of one level: M = D^-1 (or inverted diagonal blocks), rho by power iteration or
// Gershgorin, interval [lower, higher] * rho.  enqueue: kernels + an async copy of rho; finish: after the
// stream has been synchronised.  Split so that a level's power iterations run while the host sweeps its
// aggregates.
// the smoother of one level on the side stream, behind everything the main stream has queued so far (its operator is
// final there).  False: overlap is off / the start vector has not arrived yet -- the caller enqueues it on the main stream
// later, as before.
static bool smoother_fork(Context &ctx, const Launch &Lmain, AmgHierarchy::Impl &I, Level &lv, int slot)
{
    const AmgParams &prm = I.prm;
    if (!prm.overlap_smoothers || prm.cheb_power_iters <= 0 || prm.relax_type >= 3) return false;
    if (I.rng_job.valid() && I.rng_job.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return false;
    {
        // A block relaxation (damped_jacobi / spai0 / unscaled chebyshev on block values) whose block copy of the level is not
        // current rebuilds it in smoother_enqueue -- device_block_graph on I.sym, the symbolic scratch the MAIN stream is using
        // for strength, aggregation and the symbolic products at the same time (round-5 advice): such a level is enqueued on
        // the main stream, by the caller, as before the overlap existed.
        const int bs = prm.block_size > 1 ? prm.block_size : 1;
        const bool scaled_by_diagonal = prm.relax_type == 0 && prm.cheb_scale;
        const bool have_blk = lv.blk_current && lv.blk && lv.blk->b == bs && lv.blk->nb == lv.n / bs && lv.blk->didx.ptr && lv.blk->val.ptr;
        if (bs > 1 && !scaled_by_diagonal && !have_blk) return false;
    }
    if (!I.side) {
        PS_HIP_CHECK(hipStreamCreateWithFlags(&I.side, hipStreamNonBlocking));
        PS_HIP_CHECK(hipEventCreateWithFlags(&I.ev_fork, hipEventDisableTiming));
        PS_HIP_CHECK(hipEventCreateWithFlags(&I.ev_join, hipEventDisableTiming));
    }
    I.partials_side.ensure(2 * (size_t)kMaxPartials);
    if (I.forks == 0) ctx.meter.fork(); // blocks released from here on stay out of the cache until the streams have met again
    ++I.forks;
    PS_HIP_CHECK(hipEventRecord(I.ev_fork, Lmain.stream));
    PS_HIP_CHECK(hipStreamWaitEvent(I.side, I.ev_fork, 0));
    Launch Ls = Lmain;
    Ls.stream = I.side;
    smoother_enqueue(ctx, Ls, I, lv, slot, true);
    lv.L.stream = Lmain.stream; // (the cycle launches on the caller's stream)
    lv.smoother_enqueued = true;
    return true;
}

// the main stream waits for the side stream's work; afterwards (and after a synchronisation of the main stream) the radii and
// the flags of the forked smoothers are on the host
static void smoothers_join(Context &ctx, const Launch &Lmain, AmgHierarchy::Impl &I)
{
    if (I.forks == 0) return;
    PS_HIP_CHECK(hipEventRecord(I.ev_join, I.side));
    PS_HIP_CHECK(hipStreamWaitEvent(Lmain.stream, I.ev_join, 0));
    I.forks = 0;
    ctx.meter.join();
}

static void smoother_enqueue(Context &ctx, const Launch &Lbase, AmgHierarchy::Impl &I, Level &lv, int slot, bool on_side)
{
    const AmgParams &prm = I.prm;
    PS_REQUIRE(slot >= 0 && slot < kMaxLevelSlots, PSOLVE_HIP_EINVAL, "AMG: too many levels");
    lv.L = fit_launch(ctx.launch_max(), lv.n, lv.A.rows_per_block, lv.n > 0 ? (double)lv.A.nnz / lv.n : 0.0);
    lv.L.stream = Lbase.stream;
    if (prm.stream_nt == 0) { // the cycle re-reads what it has just written: keep it in the caches
        lv.L.spmv_nt = 0;
        if (lv.L.spmv_kernel < 0) lv.L.spmv_kernel = 0;
    }
    const Launch &L = lv.L;
    hipStream_t s = L.stream;
    const size_t n = (size_t)lv.n;
    lv.dinv.ensure(n);
    int *bad = I.bad_flags.ptr + slot;
    PS_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), s));
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    lv.direct = false;
    lv.jacobi_like = prm.relax_type != 0;
    const bool sweeps = prm.relax_type >= 3;
    lv.sweeps = 0;
    if (prm.direct_coarse && lv.smoother_is_coarsest) {
        // the coarsest level is solved directly (coarse_solver_setup): no smoother
        I.rho_host.ptr[slot] = 1.0;
        return;
    }
    // what scales the residual of a smoothing step: the inverted diagonal (chebyshev, scale = true), the identity
    // (scale = false), or the whole relaxation M of damped_jacobi / spai0 (amg_relax.hip)
    const int scaling = sweeps ? 0 : prm.relax_type == 1 ? 1 : prm.relax_type == 2 ? 2 : (prm.cheb_scale ? 0 : 3);
    if (bs == 1) {
        if (scaling == 0) launch_diag_inverse(L, lv.A, lv.dinv.ptr, bad); // (block value types scale by the inverted diagonal BLOCKS)
        else launch_relax_scaling(L, lv.A, scaling, prm.damping, lv.dinv.ptr, bad);
    }
    if (bs > 1) {
        PS_REQUIRE(lv.n % bs == 0, PSOLVE_HIP_EINVAL, "AMG: level size is not a multiple of block_size");
        lv.dinv_blk.ensure((size_t)(lv.n / bs) * bs * bs);
        const bool have_blk = lv.blk_current && lv.blk && lv.blk->b == bs && lv.blk->nb == lv.n / bs && lv.blk->didx.ptr && lv.blk->val.ptr;
        if ((scaling != 0 || sweeps) && !have_blk) {
            // the block relaxations read the level's blocks: make the block copy of this level current
            if (!(lv.blk == &lv.blk_own && lv.blk_own_built && lv.blk_own.nb == lv.n / bs)) {
                device_block_graph(L, lv.A, bs, lv.blk_own, I.sym);
                lv.blk_own_built = true;
                lv.blk = &lv.blk_own;
                lv.blk_shared = false;
            }
            device_block_values(L, lv.A, lv.blk_own);
            lv.blk_current = true;
        }
        if (scaling == 2 || scaling == 3) {
            launch_block_relax_scaling(L, *lv.blk, scaling, prm.damping, lv.dinv_blk.ptr, bad);
        } else {
            if (have_blk || scaling != 0 || sweeps)
                launch_block_diag_inverse_bsr(L, lv.blk->nb, bs, lv.blk->didx.ptr, lv.blk->val.ptr, lv.dinv_blk.ptr, bad);
            else
                launch_block_diag_inverse(L, lv.A, bs, lv.dinv_blk.ptr, bad);
            if (scaling == 1) launch_block_relax_scaling(L, *lv.blk, 1, prm.damping, lv.dinv_blk.ptr, bad);
        }
        I.bad_host.ensure(kMaxLevelSlots);
        I.bad_host.ptr[slot] = 0;
        PS_HIP_CHECK(hipMemcpyAsync(I.bad_host.ptr + slot, bad, sizeof(int), hipMemcpyDeviceToHost, s));
        if (!on_side) { // (a forked smoother is checked in smoother_finish, after the join)
            PS_HIP_CHECK(hipStreamSynchronize(s));
            PS_REQUIRE(I.bad_host.ptr[slot] == 0, PSOLVE_HIP_ENUMERIC, "AMG: singular diagonal block");
        }
    }
    if (sweeps) {
        lv.sweeps = prm.relax_type;
        lv.sw_ctrl.ensure(8);
        if (prm.relax_type == 4) { // ilu0: the factorization (synchronises)
            const SweepView V = sweep_view(lv, bs);
            lv.sw_dinv.ensure((size_t)V.nb * bs * bs + 2);
            device_ilu0_factor(L, V, lv.sw_work, lv.sw_lu, lv.sw_dinv.ptr, lv.sw_ctrl.ptr);
        }
    }
    if (prm.relax_type != 0) {
        I.rho_host.ptr[slot] = 1.0; // (damped_jacobi / spai0 / the sweeps need no spectral radius)
    } else if (prm.cheb_power_iters > 0) {
        // "amg.refresh_power_iters" >= 0 (opt-in, NOT amgcl's estimate): a refresh continues the power iteration from the
        // vector the previous factorize ended with -- Newton's next Hessian is close to the last one -- for that many steps
        // (0: keeps the previous radius) instead of starting cheb_power_iters steps from the random vector again
        const bool keep = prm.refresh_power_iters >= 0;
        const bool warm = keep && I.in_refresh && lv.pw_valid && lv.rho > 0;
        if (warm && prm.refresh_power_iters == 0) {
            I.rho_host.ptr[slot] = lv.rho;
        } else {
            if (!warm) level_b0_scale(I, lv, bs);
            power_iteration_enqueue(L, I, lv, warm ? prm.refresh_power_iters : prm.cheb_power_iters, bs, I.rho_host.ptr + slot,
                                    on_side ? I.partials_side.ptr : I.partials.ptr, warm, keep);
        }
    } else {
        PS_REQUIRE(prm.cheb_scale != 0, PSOLVE_HIP_EINVAL, "amg.cheb_power_iters = 0 (Gershgorin) needs amg.cheb_scale = 1 in this build");
        if (bs > 1) {
            // block value types: max_i (sum_j ||A_ij||_F) ||D_i^-1||_F on the level's blocks (amgcl::backend::spectral_radius
            // with power_iters = 0; the oracle's block_gershgorin)
            const bool have_blk = lv.blk_current && lv.blk && lv.blk->b == bs && lv.blk->nb == lv.n / bs && lv.blk->didx.ptr && lv.blk->val.ptr;
            if (!have_blk) {
                if (!(lv.blk == &lv.blk_own && lv.blk_own_built && lv.blk_own.nb == lv.n / bs)) {
                    device_block_graph(L, lv.A, bs, lv.blk_own, I.sym);
                    lv.blk_own_built = true;
                    lv.blk = &lv.blk_own;
                    lv.blk_shared = false;
                }
                device_block_values(L, lv.A, lv.blk_own);
                lv.blk_current = true;
            }
            I.rho_host.ptr[slot] = device_block_gershgorin(L, *lv.blk, I.partials.ptr);
        } else {
            I.rho_host.ptr[slot] = device_gershgorin(L, I, lv.A);
        }
    }
}

// "amg.direct_coarse" (/AMGCL/precond/direct_coarse, AMGCL.cpp:46 sets it false; amgcl/amg.hpp: the coarsest level then gets
// a direct solver): the coarsest operator is inverted densely on the device, the cycle multiplies by the inverse
static void coarse_solver_setup(Context &ctx, const Launch &L, AmgHierarchy::Impl &I)
{
    if (I.lv.empty()) return;
    Level &lv = *I.lv.back();
    lv.direct = false;
    const AmgParams &prm = I.prm;
    CsrDev A = lv.A;
    A.bsr3 = nullptr;
    if (prm.direct_coarse) {
        device_dense_inverse(L, A, lv.cinv, lv.cinv_work);
        lv.direct = true;
        return;
    }
    // "amg.coarse_dense" (round 5): a coarsest level that is RELAXED (the reference's configuration) is a fixed linear map
    // rhs -> x (x = 0 on entry, npre + npost smoother applications); for a level of at most coarse_dense rows that map is
    // built once per factorize as a dense matrix -- the smoother's own recurrence run on the identity, one launch per step --
    // and a visit is one dense product instead of (npre + npost) x degree launches of a few microseconds each.  Same
    // operator up to rounding (the cycle's action against the oracle's stays within the parity tolerance).
    const int steps = prm.npre + prm.npost;
    if (prm.coarse_dense <= 0 || lv.n > prm.coarse_dense || I.lv.size() < 2 || steps <= 0 || I.top.on || prm.relax_type >= 3) {
        lv.cinv.release();
        lv.cinv_work.release();
        return;
    }
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    const size_t nn = (size_t)lv.n * lv.n;
    DeviceBuffer<double> pm;
    lv.cinv.ensure(nn + 2);
    lv.cinv_work.ensure(nn + 2);
    pm.ensure(nn + 2);
    const int degree = lv.jacobi_like ? 1 : prm.cheb_degree;
    // the last step must land in cinv: count the steps, start in the buffer that makes it so
    const int total = steps * degree;
    double *cur = (total & 1) ? lv.cinv_work.ptr : lv.cinv.ptr, *other = (total & 1) ? lv.cinv.ptr : lv.cinv_work.ptr;
    const double d = lv.d, c = lv.c;
    bool first = true;
    for (int a = 0; a < steps; ++a) {
        double alpha = 0.0, beta = 0.0;
        for (int k = 0; k < degree; ++k) {
            if (k == 0) {
                alpha = 1.0 / d;
                beta = 0.0;
            } else if (k == 1) {
                alpha = 2 * d * (1.0 / (2 * d * d - c * c));
                beta = alpha * d - 1.0;
            } else {
                alpha = 1.0 / (d - 0.25 * alpha * c * c);
                beta = alpha * d - 1.0;
            }
            if (lv.jacobi_like) {
                alpha = 1.0;
                beta = 0.0;
            }
            launch_dense_smoother_step(L, A, bs, lv.dinv.ptr, lv.dinv_blk.ptr, cur, pm.ptr, other, alpha, beta, first);
            std::swap(cur, other);
            first = false;
        }
    }
    PS_HIP_CHECK(hipStreamSynchronize(L.stream)); // (pm dies with this frame)
    lv.direct = true;
}

static void smoother_finish(AmgHierarchy::Impl &I, Level &lv, int slot)
{
    const AmgParams &prm = I.prm;
    if (prm.block_size > 1 && I.bad_host.ptr)
        PS_REQUIRE(I.bad_host.ptr[slot] == 0, PSOLVE_HIP_ENUMERIC, "AMG: singular diagonal block");
    double hi = I.rho_host.ptr[slot];
    if (prm.cheb_power_iters > 0 && hi < 0) hi = 2.0;
    PS_REQUIRE(std::isfinite(hi) && hi > 0, PSOLVE_HIP_ENUMERIC, "AMG: spectral radius estimate is not positive/finite");
    lv.rho = hi;
    const double lo = hi * prm.cheb_lower;
    hi *= prm.cheb_higher;
    lv.d = 0.5 * (hi + lo);
    lv.c = 0.5 * (hi - lo);
}

static void setup_smoother(Context &ctx, const Launch &Lbase, AmgHierarchy::Impl &I, Level &lv, int slot)
{
    smoother_enqueue(ctx, Lbase, I, lv, slot);
    PS_HIP_CHECK(hipStreamSynchronize(Lbase.stream));
    smoother_finish(I, lv, slot);
}

static unsigned long long pattern_hash(const Launch &L, AmgHierarchy::Impl &I, const CsrDev &A)
{
    I.hash_dev.ensure(2);
    PS_HIP_CHECK(hipMemsetAsync(I.hash_dev.ptr, 0, 2 * sizeof(unsigned long long), L.stream));
    launch_hash_i32(L, (int64_t)A.n + 1, A.rowptr, I.hash_dev.ptr);
    launch_hash_i32(L, A.nnz, A.col, I.hash_dev.ptr + 1);
    unsigned long long h[2];
    PS_HIP_CHECK(hipMemcpyAsync(h, I.hash_dev.ptr, sizeof(h), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    return h[0] * 0x9E3779B97F4A7C15ull + h[1];
}

void AmgHierarchy::setup(Context &ctx, const CsrDev &A, const AmgParams &prm_in)
{
    Impl &I = *impl;
    AmgParams prm = prm_in;
    if (prm.precond_class == 1) { // "amg.class" relaxation: the system matrix's smoother alone -- a hierarchy of one level
        prm.max_levels = 1;
        prm.direct_coarse = 0;
        prm.coarse_dense = 0;
    }
    const double t_entry = wall_seconds();
    I.top.on = false;
    const Launch L = ctx.launch_config();
    I.partials.ensure(2 * (size_t)kMaxPartials);
    I.rho_host.ensure(kMaxLevelSlots);
    I.bad_flags.ensure(kMaxLevelSlots);
    I.reused = false;
    const bool device_path = prm.device_setup != 0;
    const bool reusable_cfg = prm.reuse && device_path && prm.eps_strong == 0.0;
    unsigned long long h = 0;
    if (reusable_cfg) h = (ctx.pattern_id_of_A() != 0 && A.rowptr == ctx.A.rowptr && A.col == ctx.A.col) ? ctx.pattern_id_of_A() : pattern_hash(L, I, A);
    // same sparsity pattern as the hierarchy we hold, same coarsening parameters: keep the aggregates
    // and every pattern, redo the numbers on the device (what Newton needs: it refactorizes a matrix
    // of constant pattern every iteration, Newton.cpp:189-193; cf. MAS's lazy_partitioning)
    if (reusable_cfg && I.symbolic_valid && h == I.pattern_hash && A.n == I.pattern_n && A.nnz == I.pattern_nnz &&
        prm.max_levels == I.prm.max_levels && prm.coarse_enough == I.prm.coarse_enough &&
        prm.sa_relax == I.prm.sa_relax && prm.estimate_spectral_radius == I.prm.estimate_spectral_radius &&
        prm.block_size == I.prm.block_size && prm.aggregation == I.prm.aggregation && prm.coarsening == I.prm.coarsening &&
        prm.over_interp == I.prm.over_interp && prm.direct_coarse == I.prm.direct_coarse) {
        I.prm = prm;
        if (prm.cheb_power_iters > 0 && !I.lv.empty() && I.lv[0]->b0_n != I.lv[0]->n) { // power iterations were off so far
            const int bs = prm.block_size > 1 ? prm.block_size : 1;
            start_rng(I, (size_t)std::max(1, A.n / bs), bs, ctx.device, ctx.stream);
        }
        for (auto &lv : I.lv) { // the refresh kernels work on (and the power iterations use) the double-precision operators
            lv->A.val32 = nullptr;
            lv->A_own.view.val32 = nullptr;
            lv->P.view.val32 = nullptr;
            lv->R.view.val32 = nullptr;
            lv->A.sell = nullptr; // (stale numbers until apply_matrix_precision refills the copies)
            if (lv.get() != I.lv[0].get()) lv->A.bsr3 = nullptr; // (level 0 multiplies through the solver's own, current, block copy)
            lv->A_own.view.bsr3 = nullptr;
            lv->P.view.bsr3 = nullptr;
            lv->R.view.bsr3 = nullptr;
            lv->A_own.view.sell = nullptr;
            lv->P.view.sell = nullptr;
            lv->R.view.sell = nullptr;
        }
        const bool ok = refresh_numeric(ctx, L, I, A);
        if (ok) apply_matrix_precision(L, I);
        if (ok) {
            Launch Lm = ctx.launch_max();
            Lm.stream = L.stream;
            attach_block_copies(Lm, I);
        }
        finish_rng(I);
        drop_rng_host(I);
        if (ok) {
            I.reused = true;
            return;
        }
    }
    I.prm = prm;
    I.symbolic_valid = false; // a failed setup must not be "refreshed" later
    if (prm.cheb_power_iters > 0) {
        const int bs = prm.block_size > 1 ? prm.block_size : 1;
        start_rng(I, (size_t)std::max(1, A.n / bs), bs, ctx.device, ctx.stream);
    }
    const bool timing_s = std::getenv("PSOLVE_TIMING") != nullptr;
    if (timing_s) std::fprintf(stderr, "[psolve timing] amg setup entry -> hierarchy start      %.4f s\n", wall_seconds() - t_entry);
    double ts0 = wall_seconds();
    auto slap = [&](const char *what) {
        if (!timing_s) return;
        (void)hipStreamSynchronize(L.stream);
        const double t1 = wall_seconds();
        std::fprintf(stderr, "[psolve timing] amg setup %-26s %.4f s\n", what, t1 - ts0);
        ts0 = t1;
    };
    if (device_path) device_full_setup(ctx, L, I, A);
    else full_setup(ctx, L, I, A);
    slap("hierarchy (laps above)");
    apply_matrix_precision(L, I);
    slap("cycle copies (col16, row-blocks)");
    {
        Launch Lm = ctx.launch_max();
        Lm.stream = L.stream;
        attach_block_copies(Lm, I);
    }
    slap("block copies");
    finish_rng(I);
    slap("rng thread joined");
    drop_rng_host(I); // the levels keep their scales; the device keeps the stream
    slap("host draws freed");
    I.symbolic_valid = reusable_cfg;
    I.pattern_hash = h;
    I.pattern_n = A.n;
    I.pattern_nnz = A.nnz;
}

// Consecutive products on one operator inside a cycle sweep it from alternating ends: what a product leaves in the Infinity
// Cache is the tail of its stream, which the next one then starts with.  None of these launches reduces, and a row's sum
// does not depend on when its row-block runs: the cycle's action is bit for bit the same.
static inline int next_sweep(Level &lv) { return (lv.L.lab.alternate & 8) ? 0 : (lv.sweep ^= 1); }

// the operator of the ordered relaxations: the level's CSR arrays, or its block copy (block value types)
static SweepView sweep_view(const Level &lv, int bs)
{
    SweepView V;
    V.b = bs;
    if (bs > 1) {
        V.nb = lv.blk->nb;
        V.nnzb = lv.blk->nnzb;
        V.ptr = lv.blk->ptr.ptr;
        V.col = lv.blk->col.ptr;
        V.val = lv.blk->val.ptr;
    } else {
        V.nb = lv.A.n;
        V.nnzb = lv.A.nnz;
        V.ptr = lv.A.rowptr;
        V.col = lv.A.col;
        V.val = lv.A.val;
    }
    return V;
}

// gauss_seidel::apply_pre / apply_post (a forward / a backward sweep), ilu0::apply_pre = apply_post (x += damping (LU)^-1 (rhs - A x))
static void sweep_apply(const Launch &L, Level &lv, const AmgParams &prm, const double *rhs, double *x, bool x_is_zero, int bs,
                        const int *done, bool post)
{
    const size_t bytes = (size_t)lv.n * sizeof(double);
    SweepView V = sweep_view(lv, bs);
    if (x_is_zero) PS_HIP_CHECK(hipMemsetAsync(x, 0, bytes, L.stream));
    if (lv.sweeps == 3) {
        const double *dinv = bs > 1 ? lv.dinv_blk.ptr : lv.dinv.ptr;
        launch_sweep(L, V, post ? 1 : 0, dinv, rhs, x, lv.xb.ptr, lv.sw_ctrl.ptr, done);
        PS_HIP_CHECK(hipMemcpyAsync(x, lv.xb.ptr, bytes, hipMemcpyDeviceToDevice, L.stream));
        return;
    }
    const double *t = rhs;
    if (!x_is_zero) {
        launch_spmv(L, lv.A, SPMV_RESIDUAL, x, rhs, lv.t.ptr, nullptr, done);
        t = lv.t.ptr;
    }
    V.val = lv.sw_lu.ptr;
    launch_sweep(L, V, 2, nullptr, t, nullptr, lv.p.ptr, lv.sw_ctrl.ptr, done);
    launch_sweep(L, V, 3, lv.sw_dinv.ptr, lv.p.ptr, nullptr, lv.xb.ptr, lv.sw_ctrl.ptr, done);
    launch_axpby(L, lv.n, prm.ilu_damping, lv.xb.ptr, 1.0, x);
}

// chebyshev::solve: `degree` steps on (A, rhs) starting from x (x_is_zero: x == 0, first residual = rhs)
static void cheb_solve(const Launch &L, Level &lv, int degree, const double *rhs, double *x, bool x_is_zero, int bs,
                       const int *done, bool fuse_block = true, bool post = false, const AmgParams *prm = nullptr)
{
    if (lv.sweeps) { // gauss_seidel / ilu0: apply_pre / apply_post are sweeps, not polynomial steps
        sweep_apply(L, lv, *prm, rhs, x, x_is_zero, bs, done, post);
        return;
    }
    const double d = lv.d, c = lv.c;
    const bool jacobi_like = lv.jacobi_like;
    if (jacobi_like) degree = 1; // (one application of amgcl's apply_pre / apply_post)
    double alpha = 0.0, beta = 0.0;
    double *cur = x, *other = lv.xb.ptr;
    SpmvExtra exb;
    exb.dinv_blk = lv.dinv_blk.ptr;
    exb.p = lv.p.ptr;
    // Round 6: beyond the Infinity Cache the step runs SPLIT -- the residual product (one operand load and one store per row
    // behind the row sums), then the node-local update as a second, purely streaming launch (120 MB at configs[2]'s size) --
    // which the review of round 5 asked to measure: the fused epilogue's six operand loads and two stores per row sit on the
    // single-buffered workgroup's chain, and once every byte is an HBM byte that costs more than the 7 % of traffic it saves.
    // Level-0 step under a random node numbering, fused | split (profiles/r06_cheb_split.jsonl): M = 100 (2.0 GB of blocks)
    // 445 | 407 us, solve 102.4 -> 98.3 ms; M = 80 (1.0 GB) 221 | 206 us; M = 64 (0.5 GB) 100 | 101 us; M = 40 23 | 27 us.
    // Same operations in the same order either way (bit-equal iterates).  "lab.cheb_split": -1 by size (768 MiB of blocks),
    // 0 never, 1 always; an operator with block-row kinds streams no matrix and stays fused.
    const bool big = lv.A.bsr3 && 76ll * lv.A.bsr3->nnzb >= (768ll << 20);
    const bool split = (L.lab.cheb_split > 0 || (L.lab.cheb_split < 0 && big)) && !(lv.A.bsr3 && lv.A.bsr3->kinds);
    if (bs == 3 && fuse_block && !split && lv.A.bsr3 && bsr3_serves(*lv.A.bsr3, SPMV_CHEB, L, exb)) {
        // 3x3-block copy: the block-scaled step is an epilogue of the block product (the three residuals of a node meet in
        // LDS) -- one launch per step like the scalar path, the iterate ping-pongs between x and xb
        if (x_is_zero && ((degree - 1) & 1)) std::swap(cur, other);
        for (int k = 0; k < degree; ++k) {
            if (k == 0) {
                alpha = 1.0 / d;
                beta = 0.0;
            } else if (k == 1) {
                alpha = 2 * d * (1.0 / (2 * d * d - c * c));
                beta = alpha * d - 1.0;
            } else {
                alpha = 1.0 / (d - 0.25 * alpha * c * c);
                beta = alpha * d - 1.0;
            }
            if (jacobi_like) { // damped_jacobi / spai0: x <- x + M (rhs - A x), M in the place of the inverted diagonal
                alpha = 1.0;
                beta = 0.0;
            }
            if (k == 0 && x_is_zero) {
                launch_block_cheb_update(L, lv.n, bs, lv.dinv_blk.ptr, rhs, lv.p.ptr, cur, alpha, beta, true);
                continue;
            }
            exb.alpha = alpha;
            exb.beta = beta;
            exb.reverse = next_sweep(lv);
            launch_spmv(L, lv.A, SPMV_CHEB, cur, rhs, other, nullptr, done, &exb);
            std::swap(cur, other);
        }
        if (cur != x)
            PS_HIP_CHECK(hipMemcpyAsync(x, cur, (size_t)lv.n * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
        return;
    }
    if (bs > 1) {
        // block scaling needs all residuals of a node: residual SpMV, then a node-local update in place
        for (int k = 0; k < degree; ++k) {
            if (k == 0) {
                alpha = 1.0 / d;
                beta = 0.0;
            } else if (k == 1) {
                alpha = 2 * d * (1.0 / (2 * d * d - c * c));
                beta = alpha * d - 1.0;
            } else {
                alpha = 1.0 / (d - 0.25 * alpha * c * c);
                beta = alpha * d - 1.0;
            }
            if (jacobi_like) { // damped_jacobi / spai0: x <- x + M (rhs - A x), M in the place of the inverted diagonal
                alpha = 1.0;
                beta = 0.0;
            }
            const bool zero = (k == 0 && x_is_zero);
            const double *t = rhs;
            if (!zero) {
                SpmvExtra exr;
                exr.reverse = next_sweep(lv);
                launch_spmv(L, lv.A, SPMV_RESIDUAL, x, rhs, lv.t.ptr, nullptr, done, &exr);
                t = lv.t.ptr;
            }
            launch_block_cheb_update(L, lv.n, bs, lv.dinv_blk.ptr, t, lv.p.ptr, x, alpha, beta, zero);
        }
        return;
    }
    // every product step ping-pongs between x and xb: from x = 0 the first iterate may start in either
    // buffer, so start where the last step lands in x (no copy at the end)
    if (x_is_zero && ((degree - 1) & 1)) std::swap(cur, other);
    for (int k = 0; k < degree; ++k) {
        if (k == 0) {
            alpha = 1.0 / d;
            beta = 0.0;
        } else if (k == 1) {
            alpha = 2 * d * (1.0 / (2 * d * d - c * c));
            beta = alpha * d - 1.0;
        } else {
            alpha = 1.0 / (d - 0.25 * alpha * c * c);
            beta = alpha * d - 1.0;
        }
        if (jacobi_like) { // damped_jacobi / spai0: x <- x + M (rhs - A x), M in the place of the inverted diagonal
            alpha = 1.0;
            beta = 0.0;
        }
        if (k == 0 && x_is_zero) {
            launch_cheb_first(L, lv.n, alpha, lv.dinv.ptr, rhs, lv.p.ptr, cur); // iterate = p
            continue;
        }
        SpmvExtra ex;
        ex.dinv = lv.dinv.ptr;
        ex.p = lv.p.ptr;
        ex.alpha = alpha;
        ex.beta = beta;
        ex.reverse = next_sweep(lv);
        launch_spmv(L, lv.A, SPMV_CHEB, cur, rhs, other, nullptr, done, &ex);
        std::swap(cur, other);
    }
    if (cur != x)
        PS_HIP_CHECK(hipMemcpyAsync(x, cur, (size_t)lv.n * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
}

static void cycle(AmgHierarchy::Impl &I, const Launch &Lbase, size_t l, const double *rhs, double *x, bool x_is_zero,
                  const int *done)
{
    Level &lv = *I.lv[l];
    const AmgParams &prm = I.prm;
    Launch L = lv.L; // grids fitted to this level
    L.stream = Lbase.stream;
    if (l + 1 == I.lv.size()) {
        if (lv.direct) { // "amg.direct_coarse": x = A_c^-1 rhs
            launch_dense_matvec(L, lv.n, lv.cinv.ptr, rhs, x, done);
            return;
        }
        // coarsest level: relaxed, not factorised (direct_coarse = false, AMGCL.cpp:46)
        bool zero = x_is_zero;
        for (int i = 0; i < prm.npre + prm.npost; ++i) { // (amgcl: npre x apply_pre, then npost x apply_post)
            cheb_solve(L, lv, prm.cheb_degree, rhs, x, zero, prm.block_size, done, prm.block_levels != 0, i >= prm.npre, &prm);
            zero = false;
        }
        if (zero) PS_HIP_CHECK(hipMemsetAsync(x, 0, (size_t)lv.n * sizeof(double), L.stream));
        return;
    }
    Level &nx = *I.lv[l + 1];
    Launch Ln = nx.L;
    Ln.stream = Lbase.stream;
    bool zero = x_is_zero;
    for (int j = 0; j < prm.ncycle; ++j) {
        for (int i = 0; i < prm.npre; ++i) {
            cheb_solve(L, lv, prm.cheb_degree, rhs, x, zero, prm.block_size, done, prm.block_levels != 0, false, &prm);
            zero = false;
        }
        if (zero) { // npre == 0: x = 0, residual = rhs
            PS_HIP_CHECK(hipMemsetAsync(x, 0, (size_t)lv.n * sizeof(double), L.stream));
            zero = false;
        }
        SpmvExtra exr;
        exr.reverse = next_sweep(lv);
        launch_spmv(L, lv.A, SPMV_RESIDUAL, x, rhs, lv.t.ptr, nullptr, done, &exr);
        launch_spmv(Ln, lv.R.view, SPMV_PLAIN, lv.t.ptr, nullptr, nx.f.ptr, nullptr, done);
        cycle(I, Lbase, l + 1, nx.f.ptr, nx.u.ptr, true, done);
        launch_spmv(L, lv.P.view, SPMV_ADD, nx.u.ptr, nullptr, x, nullptr, done);
        for (int i = 0; i < prm.npost; ++i)
            cheb_solve(L, lv, prm.cheb_degree, rhs, x, false, prm.block_size, done, prm.block_levels != 0, true, &prm);
    }
}

void AmgHierarchy::time_level_ops(Context &ctx, int l, int reps, double out_us[5])
{
    Impl &I = *impl;
    PS_REQUIRE(l >= 0 && l < (int)I.lv.size() && reps >= 1, PSOLVE_HIP_EINVAL, "amg time_level_ops: no such level");
    PS_REQUIRE(!I.top.on, PSOLVE_HIP_EINVAL, "amg time_level_ops: single-device hierarchies only");
    Level &lv = *I.lv[(size_t)l];
    const AmgParams &prm = I.prm;
    Launch L = lv.L;
    L.stream = ctx.stream;
    DeviceBuffer<double> rhs, x;
    rhs.ensure((size_t)lv.n + 2);
    x.ensure((size_t)lv.n + 2);
    PS_HIP_CHECK(hipMemsetAsync(rhs.ptr, 0, (size_t)lv.n * sizeof(double), L.stream));
    PS_HIP_CHECK(hipMemsetAsync(x.ptr, 0, (size_t)lv.n * sizeof(double), L.stream));
    hipEvent_t e0, e1;
    PS_HIP_CHECK(hipEventCreate(&e0));
    PS_HIP_CHECK(hipEventCreate(&e1));
    auto timed = [&](const std::function<void()> &op, int launches_per_call) {
        op(); // warm
        PS_HIP_CHECK(hipEventRecord(e0, L.stream));
        for (int r = 0; r < reps; ++r) op();
        PS_HIP_CHECK(hipEventRecord(e1, L.stream));
        PS_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        PS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3 * (double)ms / ((double)reps * launches_per_call);
    };
    for (int k = 0; k < 5; ++k) out_us[k] = 0.0;
    const bool fused = prm.block_levels != 0;
    if (lv.direct) { // "amg.direct_coarse": the level has no smoother -- what a visit costs is the dense product
        out_us[0] = out_us[4] = timed([&] { launch_dense_matvec(L, lv.n, lv.cinv.ptr, rhs.ptr, x.ptr, nullptr); }, 1);
        out_us[1] = timed([&] { launch_spmv(L, lv.A, SPMV_RESIDUAL, x.ptr, rhs.ptr, lv.t.ptr, nullptr, nullptr); }, 1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return;
    }
    // two steps from a non-zero iterate: two product launches (block value types without the fused epilogue: + two updates)
    out_us[0] = timed([&] { cheb_solve(L, lv, 2, rhs.ptr, x.ptr, false, prm.block_size, nullptr, fused, false, &prm); }, 2);
    out_us[1] = timed([&] { launch_spmv(L, lv.A, SPMV_RESIDUAL, x.ptr, rhs.ptr, lv.t.ptr, nullptr, nullptr); }, 1);
    if (l + 1 < (int)I.lv.size()) {
        Level &nx = *I.lv[(size_t)l + 1];
        Launch Ln = nx.L;
        Ln.stream = L.stream;
        out_us[2] = timed([&] { launch_spmv(Ln, lv.R.view, SPMV_PLAIN, lv.t.ptr, nullptr, nx.f.ptr, nullptr, nullptr); }, 1);
        PS_HIP_CHECK(hipMemsetAsync(nx.u.ptr, 0, (size_t)nx.n * sizeof(double), L.stream));
        out_us[3] = timed([&] { launch_spmv(L, lv.P.view, SPMV_ADD, nx.u.ptr, nullptr, x.ptr, nullptr, nullptr); }, 1);
    }
    out_us[4] = timed([&] { cheb_solve(L, lv, 1, rhs.ptr, x.ptr, true, prm.block_size, nullptr, fused, false, &prm); }, 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

// ---- level 0 on a shard of the global hierarchy (Impl::DistTop) -------------------------------------------
// chebyshev::solve on the shard's rows: same coefficients as the global level-0 smoother (rho is the global one),
// the iterate's halo exchanged before every product
static void cheb_solve_top(Context &ctx, AmgHierarchy::Impl &I, const Launch &L, int degree, const double *rhs, double *x_ext,
                           bool x_is_zero, const int *done)
{
    Level &lv0 = *I.lv[0];
    AmgHierarchy::Impl::DistTop &T = I.top;
    const double d = lv0.d, c = lv0.c;
    const bool jacobi_like = lv0.jacobi_like;
    if (jacobi_like) degree = 1;
    const double *dinv = lv0.dinv.ptr + T.row0;
    double alpha = 0.0, beta = 0.0;
    double *cur = x_ext, *other = T.xb_ext.ptr;
    if (x_is_zero && ((degree - 1) & 1)) std::swap(cur, other);
    for (int k = 0; k < degree; ++k) {
        if (k == 0) {
            alpha = 1.0 / d;
            beta = 0.0;
        } else if (k == 1) {
            alpha = 2 * d * (1.0 / (2 * d * d - c * c));
            beta = alpha * d - 1.0;
        } else {
            alpha = 1.0 / (d - 0.25 * alpha * c * c);
            beta = alpha * d - 1.0;
        }
        if (jacobi_like) { // damped_jacobi / spai0: x <- x + M (rhs - A x), M in the place of the inverted diagonal
            alpha = 1.0;
            beta = 0.0;
        }
        if (k == 0 && x_is_zero) {
            launch_cheb_first(L, T.n_loc, alpha, dinv, rhs, T.p.ptr, cur);
            continue;
        }
        ctx.halo_exchange(cur);
        SpmvExtra ex;
        ex.dinv = dinv;
        ex.p = T.p.ptr;
        ex.alpha = alpha;
        ex.beta = beta;
        launch_spmv(L, ctx.A, SPMV_CHEB, cur, rhs, other, nullptr, done, &ex);
        std::swap(cur, other);
    }
    if (cur != x_ext)
        PS_HIP_CHECK(hipMemcpyAsync(x_ext, cur, (size_t)T.n_loc * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
}

static void cycle_top(Context &ctx, AmgHierarchy::Impl &I, const Launch &Lbase, const double *rhs, double *z, const int *done)
{
    const AmgParams &prm = I.prm;
    AmgHierarchy::Impl::DistTop &T = I.top;
    Launch L = ctx.launch_config(); // grids fitted to the shard
    L.stream = Lbase.stream;
    Level &nx = *I.lv[1];
    Launch Ln = nx.L;
    Ln.stream = Lbase.stream;
    double *x = T.x_ext.ptr;
    bool zero = true;
    for (int j = 0; j < prm.ncycle; ++j) {
        for (int i = 0; i < prm.npre; ++i) {
            cheb_solve_top(ctx, I, L, prm.cheb_degree, rhs, x, zero, done);
            zero = false;
        }
        if (zero) {
            PS_HIP_CHECK(hipMemsetAsync(x, 0, (size_t)T.n_loc * sizeof(double), L.stream));
            zero = false;
        }
        ctx.halo_exchange(x);
        launch_spmv(L, ctx.A, SPMV_RESIDUAL, x, rhs, T.t.ptr, nullptr, done);
        // f_1 = R_0 t: every rank contributes the columns it owns, one all-reduce makes the sum
        launch_spmv(Ln, T.R_loc.view, SPMV_PLAIN, T.t.ptr, nullptr, nx.f.ptr, nullptr, done);
        ctx.allreduce(nx.f.ptr, T.n1);
        cycle(I, Lbase, 1, nx.f.ptr, nx.u.ptr, true, done); // replicated
        launch_spmv(L, T.P_loc.view, SPMV_ADD, nx.u.ptr, nullptr, x, nullptr, done);
        for (int i = 0; i < prm.npost; ++i) cheb_solve_top(ctx, I, L, prm.cheb_degree, rhs, x, false, done);
    }
    PS_HIP_CHECK(hipMemcpyAsync(z, x, (size_t)T.n_loc * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
}

// the shard's view of level 0 of a hierarchy that was built on the gathered global matrix
void AmgHierarchy::setup_global(Context &ctx, const CsrDev &Aglobal, int row0, int n_loc, const AmgParams &prm)
{
    Impl &I = *impl;
    I.top.on = false;
    PS_REQUIRE(prm.block_size <= 1, PSOLVE_HIP_EINVAL, "amg.dist_global serves scalar systems (block_size 1)");
    setup(ctx, Aglobal, prm);
    if (I.lv.size() < 2) return; // a single level: nothing to split; the caller keeps the per-shard hierarchy
    Impl::DistTop &T = I.top;
    Level &lv0 = *I.lv[0];
    hipStream_t s = ctx.stream;
    const Launch L = ctx.launch_config();
    T.row0 = row0;
    T.n_loc = n_loc;
    T.n1 = I.lv[1]->n;
    // local rows of P_0 (copied: the product kernels want 16-byte aligned column / value arrays)
    int ends[2] = {0, 0};
    PS_HIP_CHECK(hipMemcpyAsync(&ends[0], lv0.P.ptr.ptr + row0, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipMemcpyAsync(&ends[1], lv0.P.ptr.ptr + row0 + n_loc, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t pnnz = (int64_t)ends[1] - ends[0];
    T.P_loc.ptr.ensure((size_t)n_loc + 1);
    T.P_loc.col.ensure((size_t)pnnz + 4);
    T.P_loc.val.ensure((size_t)pnnz + 4);
    PS_HIP_CHECK(hipMemcpyAsync(T.P_loc.ptr.ptr, lv0.P.ptr.ptr + row0, ((size_t)n_loc + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
    launch_add_offset_i32(L, (int64_t)n_loc + 1, T.P_loc.ptr.ptr, -ends[0]);
    PS_HIP_CHECK(hipMemcpyAsync(T.P_loc.col.ptr, lv0.P.col.ptr + ends[0], (size_t)pnnz * sizeof(int), hipMemcpyDeviceToDevice, s));
    PS_HIP_CHECK(hipMemcpyAsync(T.P_loc.val.ptr, lv0.P.val.ptr + ends[0], (size_t)pnnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    T.P_loc.set_view(n_loc, T.n1, pnnz);
    // its transpose = the columns of R_0 this rank owns
    device_transpose_pattern(L, n_loc, T.n1, T.P_loc.ptr.ptr, T.P_loc.col.ptr, pnnz, T.R_loc.ptr, T.R_loc.col, T.r_from_p, I.sym);
    T.R_loc.val.ensure((size_t)pnnz + 4);
    T.R_loc.set_view(T.n1, n_loc, pnnz);
    launch_gather(L, (int)pnnz, T.r_from_p.ptr, T.P_loc.val.ptr, T.R_loc.val.ptr);
    const size_t ne = (size_t)ctx.A.n_ext + 2;
    T.x_ext.ensure(ne);
    T.xb_ext.ensure(ne);
    T.t.ensure((size_t)n_loc + 2);
    T.p.ensure((size_t)n_loc + 2);
    PS_HIP_CHECK(hipStreamSynchronize(s));
    T.on = true;
}

bool AmgHierarchy::global_on_shards() const { return impl->top.on; }

// amg::apply(rhs, x): x = 0, one cycle
void AmgHierarchy::apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag)
{
    PS_REQUIRE(!impl->lv.empty(), PSOLVE_HIP_EINVAL, "AMG hierarchy is empty");
    const Launch L = ctx.launch_config();
    if (impl->prm.precond_class == 1 && !impl->top.on) {
        // amgcl::relaxation::as_preconditioner::apply = relax.apply(A, rhs, x) on the system matrix: chebyshev clears x and runs
        // solve; damped_jacobi / spai0 multiply by their scaling; gauss_seidel clears x, sweeps forward, then backward; ilu0
        // solves with its factors (no damping) -- x = 0 and one apply_pre give the first three and ilu0 with damping 1
        Level &lv = *impl->lv[0];
        Launch Ll = lv.L;
        Ll.stream = L.stream;
        const AmgParams &prm = impl->prm;
        lv.sweep = 0;
        if (lv.sweeps == 4) {
            AmgParams one = prm;
            one.ilu_damping = 1.0;
            cheb_solve(Ll, lv, prm.cheb_degree, d_r, d_z, true, prm.block_size, done_flag, prm.block_levels != 0, false, &one);
        } else {
            cheb_solve(Ll, lv, prm.cheb_degree, d_r, d_z, true, prm.block_size, done_flag, prm.block_levels != 0, false, &prm);
            if (lv.sweeps == 3)
                cheb_solve(Ll, lv, prm.cheb_degree, d_r, d_z, false, prm.block_size, done_flag, prm.block_levels != 0, true, &prm);
        }
    } else if (impl->top.on) cycle_top(ctx, *impl, L, d_r, d_z, done_flag);
    else {
        for (auto &lv : impl->lv) lv->sweep = 0; // (PCG's own product sweeps forward: the cycle's first one starts at the far end)
        cycle(*impl, L, 0, d_r, d_z, true, done_flag);
    }
}

// introspection for the parity tests: shape of level l
int AmgHierarchy::operators_with_packed_row_blocks() const
{
    int c = 0;
    for (auto &lv : impl->lv) c += (lv->A.rb_start != nullptr) + (lv->R.view.rb_start != nullptr);
    return c;
}

int AmgHierarchy::levels_aggregated_on_device() const
{
    int k = 0;
    for (const auto &lv : impl->lv) k += lv->aggregated_on_device ? 1 : 0;
    return k;
}

void AmgHierarchy::level_shape(int l, int64_t *rows, int64_t *nnz, double *rho) const
{
    const Level &lv = *impl->lv.at((size_t)l);
    if (rows) *rows = lv.n;
    if (nnz) *nnz = lv.A.nnz;
    if (rho) *rho = lv.rho;
}

// introspection for the parity tests: matrices of the device-resident hierarchy (0 = A, 1 = P, 2 = R)
static const CsrDev *pick_matrix(const AmgHierarchy::Impl &I, int l, int what)
{
    if (l < 0 || l >= (int)I.lv.size()) return nullptr;
    const Level &lv = *I.lv[(size_t)l];
    const CsrDev *M = what == 0 ? &lv.A : what == 1 ? &lv.P.view : what == 2 ? &lv.R.view : what == 3 ? &lv.AP.view : nullptr;
    if (!M || (what != 0 && M->n == 0)) return nullptr;
    return M;
}

void AmgHierarchy::level_matrix_shape(int l, int what, int64_t out[3]) const
{
    const CsrDev *M = pick_matrix(*impl, l, what);
    PS_REQUIRE(M != nullptr, PSOLVE_HIP_EINVAL, "amg_level_matrix: no such level / matrix");
    out[0] = M->n;
    out[1] = M->n_ext;
    out[2] = M->nnz;
}

bool AmgHierarchy::level_perm_copy(hipStream_t s, int l, int *perm) const
{
    PS_REQUIRE(l >= 0 && l < (int)impl->lv.size() && perm, PSOLVE_HIP_EINVAL, "amg_level_perm: no such level");
    const Level &lv = *impl->lv[(size_t)l];
    if (!lv.renumbered) {
        for (int i = 0; i < lv.n; ++i) perm[i] = i;
        return false;
    }
    PS_HIP_CHECK(hipMemcpyAsync(perm, lv.perm.ptr, (size_t)lv.n * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    return true;
}

void AmgHierarchy::level_matrix_copy(hipStream_t s, int l, int what, int *rowptr, int *col, double *val) const
{
    const CsrDev *M = pick_matrix(*impl, l, what);
    PS_REQUIRE(M != nullptr, PSOLVE_HIP_EINVAL, "amg_level_matrix: no such level / matrix");
    PS_HIP_CHECK(hipMemcpyAsync(rowptr, M->rowptr, ((size_t)M->n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (M->nnz) {
        PS_HIP_CHECK(hipMemcpyAsync(col, M->col, (size_t)M->nnz * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipMemcpyAsync(val, M->val, (size_t)M->nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    PS_HIP_CHECK(hipStreamSynchronize(s));
}

} // namespace psolve
