// solver.hip -- Context: matrix upload, Jacobi/AMG setup, the device-resident PCG driver.
//
// PCG driver = Eigen::internal::conjugate_gradient's recurrence (what the reference reaches through
// EigenSolver.tpp:109-114), restated in oracle/psolve_oracle.c:orc_cg_eigen, executed as three
// kernels per iteration with every scalar on the device:
//     K1  q = A p, partial p.q                               (spmv_csr_pipe<R, SPMV_DOT>)
//     K2  alpha = rz / p.q ; r -= alpha q ; partial r.r, r.z  (pcg_update_r_kernel)
//     K3  x += alpha p ; convergence latch ; beta ; p = z + beta p   (pcg_update_xp_kernel)
// The host enqueues `check_period` iterations at a time and polls an async copy of the state one
// chunk behind, so the GPU never waits for the host; once the latch is set the remaining enqueued
// kernels return immediately and x is exactly the iterate at which the recurrence residual first
// dropped below the threshold (same iteration count as the oracle, no overshoot).
// Structural model for "device scalars + sparse polling": MASSolver.cu:469-595.
#include "solver.hpp"

#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "amg.hpp"
#include "host_hash.hpp"
#include "amg_dist.hpp"
#include "amg_setup.hpp"
#include "ic.hpp"
#include "schwarz.hpp"

namespace psolve {

thread_local std::weak_ptr<AllocMeter> tl_alloc_meter;

namespace {
std::mutex g_meters_mu;
std::vector<std::weak_ptr<AllocMeter>> g_meters;
} // namespace

void register_alloc_meter(const std::shared_ptr<AllocMeter> &m)
{
    std::lock_guard<std::mutex> lk(g_meters_mu);
    g_meters.erase(std::remove_if(g_meters.begin(), g_meters.end(), [](const std::weak_ptr<AllocMeter> &w) { return w.expired(); }),
                   g_meters.end());
    g_meters.push_back(m);
}

void trim_all_alloc_meters()
{
    std::vector<std::shared_ptr<AllocMeter>> live;
    {
        std::lock_guard<std::mutex> lk(g_meters_mu);
        for (auto &w : g_meters)
            if (auto m = w.lock()) live.push_back(std::move(m));
    }
    // (hipFree synchronises the device: whoever still reads a cached block of another handle has finished by then)
    for (auto &m : live) m->trim();
}

double wall_seconds()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// partial-sum arrays inside partials_ (stride kMaxPartials)
enum { P_PQ = 0, P_RR = 1, P_RZ = 2, P_BB = 3, P_TMP = 4, P_COUNT = 6 };
// staging slots inside scal_
enum { S_INIT = 0, S_PQ = 4, S_RR = 5, S_RZ = 6, S_TMP = 8, S_COUNT = 16 };

Context::Context(int device_id) : device(device_id)
{
    tl_alloc_meter = meter_;
    register_alloc_meter(meter_);
    int count = 0;
    PS_HIP_CHECK(hipGetDeviceCount(&count));
    PS_REQUIRE(count > 0, PSOLVE_HIP_EDEVICE, "no HIP device visible (the HIP backend has no CPU fallback)");
    PS_REQUIRE(device_id >= 0 && device_id < count, PSOLVE_HIP_EINVAL, "device id out of range");
    PS_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    PS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the cap of the handle's cache of released blocks: 16 GiB, at most a sixteenth of the device (18 GiB of an MI355X's 288).
    // Round 6: with 4 GiB the transients of a first factorize of configs[2] under a random numbering (5.7 GB) went back to
    // the driver, every hipFree a device synchronisation: 65.3 -> 57.7 ms with the larger cap (scripts/r6/cache_cap.py);
    // psolve_hip_trim hands the cache back on request, "lab.alloc_cache_mb" sets the cap
    meter_->cache_mb = (int)std::min<size_t>(16384, (size_t)prop.totalGlobalMem >> 24);
    PS_HIP_CHECK(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
    stream = own_stream_;
    L_.stream = stream;
    Lmax_.stream = stream;
    L_.num_cus = Lmax_.num_cus = num_cus_;
    set_param("blocks_per_cu", prm.blocks_per_cu);
    set_param("spmv_blocks_per_cu", prm.spmv_blocks_per_cu);
    set_param("spmv_xcd_map", prm.spmv_xcd_map);
    spmv_grid_user_set_ = false; // the defaults above are not a caller's choice
    PS_HIP_CHECK(hipEventCreateWithFlags(&poll_ev_[0], hipEventDisableTiming));
    PS_HIP_CHECK(hipEventCreateWithFlags(&poll_ev_[1], hipEventDisableTiming));
    std::memset(&info, 0, sizeof(info));
    info.true_residual = -1.0;
}

Context::~Context()
{
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    meter_->trim(true); // (nothing is kept from here on: the buffers below go straight back to the device)
    amg_.reset();
    damg_.reset();
    schwarz_.reset();
    ic_.reset();
    if (loop_graph_) (void)hipGraphExecDestroy(loop_graph_);
    for (hipEvent_t e : prof_ev_) (void)hipEventDestroy(e);
    for (hipEvent_t e : prof_ev2_) (void)hipEventDestroy(e);
    for (hipEvent_t e : comm_ev_) (void)hipEventDestroy(e);
    if (poll_ev_[0]) (void)hipEventDestroy(poll_ev_[0]);
    if (poll_ev_[1]) (void)hipEventDestroy(poll_ev_[1]);
    if (ev_p_ready_) (void)hipEventDestroy(ev_p_ready_);
    if (ev_halo_done_) (void)hipEventDestroy(ev_halo_done_);
    if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

void Context::use_device() const
{
    PS_HIP_CHECK(hipSetDevice(device));
    tl_alloc_meter = meter_;
}

void Context::set_stream(void *s)
{
    if (loop_graph_) {
        (void)hipGraphExecDestroy(loop_graph_);
        loop_graph_ = nullptr;
    }
    const hipStream_t ns = s ? (hipStream_t)s : own_stream_;
    // the block cache of the handle (common.hpp: AllocMeter) relies on everything being ordered on ONE stream: what was
    // released on the old stream goes back to the driver (hipFree synchronises) before work is queued on another
    if (ns != stream && meter_) meter_->trim();
    stream = ns;
    L_.stream = stream;
    Lmax_.stream = stream;
}

void Context::synchronize()
{
    use_device();
    PS_HIP_CHECK(hipStreamSynchronize(stream));
}

void Context::set_param(const std::string &k, double v)
{
    if (loop_graph_) { // the captured loop bakes in launch geometry and scalar arguments (max_iter, ...)
        (void)hipGraphExecDestroy(loop_graph_);
        loop_graph_ = nullptr;
    }
    auto as_int = [&](int lo, int hi) {
        PS_REQUIRE(std::isfinite(v) && v >= lo && v <= hi, PSOLVE_HIP_EINVAL, "parameter '" + k + "' out of range");
        return (int)v;
    };
    if (k == "max_iter") prm.max_iter = as_int(0, 1 << 30);
    else if (k == "tolerance" || k == "relative_tolerance") {
        PS_REQUIRE(v >= 0, PSOLVE_HIP_EINVAL, "negative tolerance");
        prm.rel_tol = v;
    } else if (k == "absolute_tolerance") {
        PS_REQUIRE(v >= 0, PSOLVE_HIP_EINVAL, "negative tolerance");
        prm.abs_tol = v;
    } else if (k == "precond") prm.precond = as_int(0, 4);
    else if (k == "ic.initial_shift") {
        PS_REQUIRE(v >= 0, PSOLVE_HIP_EINVAL, "negative shift");
        prm.ic_initial_shift = v;
    }
    else if (k == "ic.ordering") prm.ic_ordering = as_int(0, 1);
    else if (k == "schwarz.levels") prm.schwarz_levels = as_int(1, 4);
    else if (k == "block_size") {
        // 2 and 3: the instantiations of AMGCL_Block the reference builds; anything else runs the scalar
        // solver there (AMGCL.cpp:111-128 falls through to the scalar AMGCL), so it does here
        PS_REQUIRE(std::isfinite(v), PSOLVE_HIP_EINVAL, "parameter 'block_size' is not finite");
        prm.block_size = (v == 2.0 || v == 3.0) ? (int)v : 1;
    } else if (k == "check_period") prm.check_period = as_int(1, 1 << 20);
    else if (k == "true_residual") prm.true_residual = as_int(0, 1);
    else if (k == "profile_spmv") prm.profile_spmv = as_int(0, 1 << 20);
    else if (k == "blocks_per_cu") {
        prm.blocks_per_cu = as_int(1, 16);
        int g = num_cus_ * prm.blocks_per_cu;
        g = (g + 7) & ~7;
        if (g > kMaxPartials) g = kMaxPartials;
        Lmax_.grid = g;
        L_.grid = g;
        if (A.n > 0) refit_launch();
    } else if (k == "vec_blocks_per_cu") {
        prm.vec_blocks_per_cu = as_int(1, 16);
        if (A.n > 0) refit_launch();
    } else if (k == "spmv_blocks_per_cu") {
        prm.spmv_blocks_per_cu = as_int(1, 16);
        spmv_grid_user_set_ = true;
        int g = num_cus_ * prm.spmv_blocks_per_cu;
        g = (g + 7) & ~7;
        if (g > kMaxPartials) g = kMaxPartials;
        Lmax_.spmv_grid = g;
        L_.spmv_grid = g;
        if (A.n > 0) refit_launch();
    } else if (k == "spmv_xcd_map") {
        prm.spmv_xcd_map = as_int(0, 2);
        L_.spmv_xcd_map = prm.spmv_xcd_map;
        Lmax_.spmv_xcd_map = prm.spmv_xcd_map;
    } else if (k == "spmv_kernel") {
        prm.spmv_kernel = as_int(-1, 3);
        L_.spmv_kernel = Lmax_.spmv_kernel = prm.spmv_kernel;
        if (A.n > 0) refit_launch(); // the dictionary kernel and the row-block kernels want different grids
    } else if (k == "spmv_nt") {
        prm.spmv_nt = as_int(-1, 1);
        L_.spmv_nt = Lmax_.spmv_nt = prm.spmv_nt;
        if (A.n > 0) refit_launch();
    } else if (k == "vec_policy") {
        prm.vec_policy = as_int(0, 15);
        L_.vec_policy = Lmax_.vec_policy = prm.vec_policy;
    } else if (k == "spmv_nt_mbytes") {
        prm.spmv_nt_mbytes = as_int(0, 1 << 20);
        L_.spmv_nt_bytes = Lmax_.spmv_nt_bytes = (int64_t)prm.spmv_nt_mbytes << 20;
        if (A.n > 0) refit_launch();
    } else if (k == "spmv_chunk_rows") {
        prm.spmv_chunk_rows = as_int(256, 1 << 24);
        L_.spmv_chunk_rows = prm.spmv_chunk_rows;
        Lmax_.spmv_chunk_rows = prm.spmv_chunk_rows;
    } else if (k == "spmv_rows_per_block") {
        const int r = as_int(0, 256);
        PS_REQUIRE(r == 0 || (r >= 8 && (r & (r - 1)) == 0), PSOLVE_HIP_EINVAL, "spmv_rows_per_block: 0 (auto) or a power of two in [8, 256]");
        prm.spmv_rows_per_block = r;
        if (A.n > 0) {
            A.rows_per_block = r ? r : spmv_rows_per_block((double)A.nnz / A.n);
            if (A.col16_R != A.rows_per_block) { // (the 16-bit columns were encoded for the other row-block height)
                A.col16 = nullptr;
                A.rb_base = nullptr;
                A.col16_R = 0;
            }
            refit_launch();
        }
    } else if (k == "dist_overlap") prm.dist_overlap = as_int(0, 1);
    else if (k == "dist_single_reduction") prm.dist_single_reduction = as_int(0, 1);
    else if (k == "dist_collectives") {
        prm.dist_collectives = as_int(0, 1);
        comm_.set_peer_collectives(prm.dist_collectives == 1);
    }
    else if (k == "dist_single_reduction_max_rows") prm.dist_single_reduction_max_rows = as_int(0, INT32_MAX);
    else if (k == "use_bsr3") prm.use_bsr3 = as_int(0, 1);
    else if (k == "bsr3_variant") { // lab: spmv_bsr3_dma's gathers before the barrier in every epilogue (1), in none (0), -1: by epilogue
        Lmax_.bsr3_variant = as_int(-1, 7);
        L_.bsr3_variant = Lmax_.bsr3_variant;
    }
    else if (k == "spmv_col16") prm.spmv_col16 = as_int(0, 1);
    else if (k == "spmv_value_dict") prm.spmv_value_dict = as_int(0, 1);
    else if (k == "use_graph") prm.use_graph = as_int(0, 1);
    else if (k == "reorder") prm.reorder = as_int(0, 2);
    else if (k == "reorder_min_rows") prm.reorder_min_rows = as_int(0, 1 << 30);
    else if (k == "reorder_reverse") prm.reorder_reverse = as_int(0, 1);
    else if (k == "reorder_min_spread") {
        PS_REQUIRE(std::isfinite(v) && v >= 0, PSOLVE_HIP_EINVAL, "parameter 'reorder_min_spread' out of range");
        prm.reorder_min_spread = v;
    }
    else if (k == "fault.solve_rank") prm.fault_solve_rank = as_int(-1, 63);
    else if (k == "amg.max_levels") prm.amg.max_levels = as_int(1, 32);
    else if (k == "amg.coarse_enough") prm.amg.coarse_enough = as_int(1, 1 << 30);
    else if (k == "amg.ncycle") prm.amg.ncycle = as_int(1, 4);
    else if (k == "amg.npre") prm.amg.npre = as_int(0, 8);
    else if (k == "amg.npost") prm.amg.npost = as_int(0, 8);
    else if (k == "amg.eps_strong") prm.amg.eps_strong = v;
    else if (k == "amg.sa_relax") prm.amg.sa_relax = v;
    else if (k == "amg.estimate_spectral_radius") prm.amg.estimate_spectral_radius = as_int(0, 1);
    else if (k == "amg.sa_power_iters") prm.amg.sa_power_iters = as_int(0, 10000);
    else if (k == "amg.cheb_degree") prm.amg.cheb_degree = as_int(1, 64);
    else if (k == "amg.cheb_power_iters") prm.amg.cheb_power_iters = as_int(0, 10000);
    else if (k == "amg.cheb_higher") prm.amg.cheb_higher = v;
    else if (k == "amg.cheb_lower") prm.amg.cheb_lower = v;
    else if (k == "amg.reuse") prm.amg.reuse = as_int(0, 1);
    else if (k == "amg.device_setup") prm.amg.device_setup = as_int(0, 1);
    else if (k == "amg.matrix_fp32") prm.amg.matrix_fp32 = as_int(0, 1);
    else if (k == "amg.stream_nt") prm.amg.stream_nt = as_int(-1, 0);
    else if (k == "amg.sell") prm.amg.sell = as_int(0, 2);
    else if (k == "amg.block_levels") prm.amg.block_levels = as_int(0, 1);
    else if (k == "amg.dist_global") prm.amg.dist_global = as_int(0, 2);
    else if (k == "amg.dist_replicate_rows") prm.amg.dist_replicate_rows = as_int(0, 1 << 30);
    else if (k == "amg.dist_global_max_mbytes") prm.amg.dist_global_max_mbytes = as_int(0, 1 << 30);
    else if (k == "amg.renumber") prm.amg.renumber = as_int(0, 1);
    else if (k == "amg.renumber_min_rows") prm.amg.renumber_min_rows = as_int(0, 1 << 30);
    else if (k == "amg.device_aggregation") prm.amg.device_aggregation = as_int(0, 1);
    else if (k == "amg.aggregation_rounds") prm.amg.aggregation_rounds = as_int(0, 1);
    else if (k == "amg.aggregation_max_rounds") prm.amg.aggregation_max_rounds = as_int(1, 1 << 24);
    else if (k == "amg.aggregation_min_rows") prm.amg.aggregation_min_rows = as_int(0, 1 << 30);
    else if (k == "amg.overlap_smoothers") prm.amg.overlap_smoothers = as_int(0, 1);
    else if (k == "amg.aggregation") prm.amg.aggregation = as_int(0, 2);
    else if (k == "amg.coarsening") prm.amg.coarsening = as_int(0, 1);
    else if (k == "amg.over_interp") prm.amg.over_interp = v;
    else if (k == "amg.relax_type") prm.amg.relax_type = as_int(0, 4);
    else if (k == "amg.ilu_damping") prm.amg.ilu_damping = v;
    else if (k == "amg.class") prm.amg.precond_class = as_int(0, 1);
    else if (k == "amg.damping") prm.amg.damping = v;
    else if (k == "amg.cheb_scale") prm.amg.cheb_scale = as_int(0, 1);
    else if (k == "amg.direct_coarse") prm.amg.direct_coarse = as_int(0, 1);
    else if (k == "amg.coarse_dense") prm.amg.coarse_dense = as_int(0, 4096);
    else if (k == "amg.refresh_power_iters") prm.amg.refresh_power_iters = as_int(-1, 10000);
    // The "lab.*" knobs belong to THIS handle (round 6; process-wide globals until round 5): kept in the handle's two Launch
    // objects, from which every derived Launch (the AMG levels', the setup's) is copied -- the levels of a hierarchy see a
    // changed knob from the next factorize on.
    else if (k.compare(0, 4, "lab.") == 0 && k != "lab.alloc_cache_poison" && k != "lab.alloc_cache_mb") {
        LabKnobs &lab = Lmax_.lab;
        if (k == "lab.dma_tile_max") lab.dma_tile_max = as_int(512, 8192) & ~255;
        else if (k == "lab.var_row_blocks") lab.var_row_blocks = as_int(0, 1);
        else if (k == "lab.symbolic_bitmap") lab.symbolic_bitmap = as_int(0, 1);
        else if (k == "lab.agg_two_pass_assign") lab.agg_two_pass_assign = as_int(0, 1);
        else if (k == "lab.cheb_split") lab.cheb_split = as_int(-1, 1);
        else if (k == "lab.kind_unroll") lab.kind_unroll = as_int(1, 4);
        else if (k == "lab.kind_sched") lab.kind_sched = as_int(-1, 1);
        else if (k == "lab.kind_probe") lab.kind_probe = as_int(0, 7);
        else if (k == "lab.kind_slots") lab.kind_slots = as_int(0, 1);
        else if (k == "lab.bsr3_kinds") lab.bsr3_kinds = as_int(0, 1);
        else if (k == "lab.alternate") lab.alternate = as_int(0, 15);
        else if (k == "lab.stage_kb") lab.stage_kb = as_int(0, 1 << 30);
        else if (k == "lab.verbose") lab.verbose = as_int(0, 9);
        else throw Error(PSOLVE_HIP_EINVAL, "unknown parameter '" + k + "'");
        L_.lab = lab;
    } else if (k == "lab.alloc_cache_poison") meter_->poison = as_int(0, 2);
    else if (k == "lab.alloc_cache_mb") {
        meter_->cache_mb = as_int(0, 1 << 20);
        if (meter_->cache_mb == 0) meter_->trim();
    }
    else throw Error(PSOLVE_HIP_EINVAL, "unknown parameter '" + k + "'");
}

// the plain parameters (everything set_param accepts): shared by get_param and by the handle-free
// psolve_hip_default_param, which reads a default-constructed Params
bool param_value(const Params &prm, const std::string &k, double *out)
{
    double v;
    if (k == "max_iter") v = prm.max_iter;
    else if (k == "tolerance" || k == "relative_tolerance") v = prm.rel_tol;
    else if (k == "absolute_tolerance") v = prm.abs_tol;
    else if (k == "precond") v = prm.precond;
    else if (k == "schwarz.levels") v = prm.schwarz_levels;
    else if (k == "ic.initial_shift") v = prm.ic_initial_shift;
    else if (k == "ic.ordering") v = prm.ic_ordering;
    else if (k == "block_size") v = prm.block_size;
    else if (k == "check_period") v = prm.check_period;
    else if (k == "true_residual") v = prm.true_residual;
    else if (k == "profile_spmv") v = prm.profile_spmv;
    else if (k == "blocks_per_cu") v = prm.blocks_per_cu;
    else if (k == "vec_blocks_per_cu") v = prm.vec_blocks_per_cu;
    else if (k == "spmv_blocks_per_cu") v = prm.spmv_blocks_per_cu;
    else if (k == "spmv_xcd_map") v = prm.spmv_xcd_map;
    else if (k == "spmv_chunk_rows") v = prm.spmv_chunk_rows;
    else if (k == "spmv_kernel") v = prm.spmv_kernel;
    else if (k == "spmv_nt") v = prm.spmv_nt;
    else if (k == "spmv_nt_mbytes") v = prm.spmv_nt_mbytes;
    else if (k == "vec_policy") v = prm.vec_policy;
    else if (k == "spmv_rows_per_block") v = prm.spmv_rows_per_block;
    else if (k == "dist_overlap") v = prm.dist_overlap;
    else if (k == "dist_single_reduction") v = prm.dist_single_reduction;
    else if (k == "dist_collectives") v = prm.dist_collectives;
    else if (k == "dist_single_reduction_max_rows") v = prm.dist_single_reduction_max_rows;
    else if (k == "use_bsr3") v = prm.use_bsr3;
    else if (k == "spmv_col16") v = prm.spmv_col16;
    else if (k == "spmv_value_dict") v = prm.spmv_value_dict;
    else if (k == "use_graph") v = prm.use_graph;
    else if (k == "reorder") v = prm.reorder;
    else if (k == "reorder_min_spread") v = prm.reorder_min_spread;
    else if (k == "reorder_min_rows") v = prm.reorder_min_rows;
    else if (k == "reorder_reverse") v = prm.reorder_reverse;
    else if (k == "amg.max_levels") v = prm.amg.max_levels;
    else if (k == "amg.coarse_enough") v = prm.amg.coarse_enough;
    else if (k == "amg.ncycle") v = prm.amg.ncycle;
    else if (k == "amg.npre") v = prm.amg.npre;
    else if (k == "amg.npost") v = prm.amg.npost;
    else if (k == "amg.eps_strong") v = prm.amg.eps_strong;
    else if (k == "amg.sa_relax") v = prm.amg.sa_relax;
    else if (k == "amg.estimate_spectral_radius") v = prm.amg.estimate_spectral_radius;
    else if (k == "amg.sa_power_iters") v = prm.amg.sa_power_iters;
    else if (k == "amg.cheb_degree") v = prm.amg.cheb_degree;
    else if (k == "amg.cheb_power_iters") v = prm.amg.cheb_power_iters;
    else if (k == "amg.cheb_higher") v = prm.amg.cheb_higher;
    else if (k == "amg.cheb_lower") v = prm.amg.cheb_lower;
    else if (k == "amg.reuse") v = prm.amg.reuse;
    else if (k == "amg.device_setup") v = prm.amg.device_setup;
    else if (k == "amg.matrix_fp32") v = prm.amg.matrix_fp32;
    else if (k == "amg.stream_nt") v = prm.amg.stream_nt;
    else if (k == "amg.sell") v = prm.amg.sell;
    else if (k == "amg.block_levels") v = prm.amg.block_levels;
    else if (k == "amg.dist_global") v = prm.amg.dist_global;
    else if (k == "amg.dist_replicate_rows") v = prm.amg.dist_replicate_rows;
    else if (k == "amg.dist_global_max_mbytes") v = prm.amg.dist_global_max_mbytes;
    else if (k == "amg.renumber") v = prm.amg.renumber;
    else if (k == "amg.renumber_min_rows") v = prm.amg.renumber_min_rows;
    else if (k == "amg.device_aggregation") v = prm.amg.device_aggregation;
    else if (k == "amg.aggregation_rounds") v = prm.amg.aggregation_rounds;
    else if (k == "amg.aggregation_max_rounds") v = prm.amg.aggregation_max_rounds;
    else if (k == "amg.aggregation_min_rows") v = prm.amg.aggregation_min_rows;
    else if (k == "amg.overlap_smoothers") v = prm.amg.overlap_smoothers;
    else if (k == "amg.aggregation") v = prm.amg.aggregation;
    else if (k == "amg.coarsening") v = prm.amg.coarsening;
    else if (k == "amg.over_interp") v = prm.amg.over_interp;
    else if (k == "amg.relax_type") v = prm.amg.relax_type;
    else if (k == "amg.ilu_damping") v = prm.amg.ilu_damping;
    else if (k == "amg.class") v = prm.amg.precond_class;
    else if (k == "amg.damping") v = prm.amg.damping;
    else if (k == "amg.cheb_scale") v = prm.amg.cheb_scale;
    else if (k == "amg.direct_coarse") v = prm.amg.direct_coarse;
    else if (k == "amg.coarse_dense") v = prm.amg.coarse_dense;
    else if (k == "amg.refresh_power_iters") v = prm.amg.refresh_power_iters;
    else return false;
    *out = v;
    return true;
}

double Context::get_param(const std::string &k) const
{
    // derived / read-only values first (some shadow a parameter with what is actually in use)
    if (k == "grid") return L_.grid;
    if (k == "spmv_grid") return L_.spmv_grid;
    if (k == "spmv_rows_per_block") return A.rows_per_block;
    if (k == "bsr3_active") return A.bsr3 ? 1 : 0;
    if (k == "bsr3_nb") return A.bsr3 ? (double)A.bsr3->nb : 0.0;       // block rows / stored 3x3 blocks of the block copy
    if (k == "bsr3_nnzb") return A.bsr3 ? (double)A.bsr3->nnzb : 0.0;
    if (k == "spmv_patterns") return A.pat ? A.pat->npat : 0; // > 0: PCG's product runs without the column stream
    if (k == "bsr3_row_kinds") return (A.bsr3 && A.bsr3->kinds) ? A.bsr3->kinds->nk : 0; // > 0: the block products stream no matrix
    if (k == "bsr3_kind_blocks") return (A.bsr3 && A.bsr3->kinds) ? A.bsr3->kinds->nblk : 0;
    if (k == "pcg_kind_diag") return L_.kd_kind ? 1 : 0; // 1: Jacobi-PCG's vector kernels read 1 / diag as table[kind[row]]
    if (k == "spmv_slots") return (A.pat && A.pat->kind) ? A.pat->nslot : 0; // > 0: ... in the slot form (spmv_csr_slots)
    if (k == "spmv_row_kinds") return (A.pat && A.pat->kind) ? A.pat->nkind : 0; // > 0: ... and without the value stream
    if (k == "sell_active") return A.sell ? 1 : 0;
    if (k == "col16_active") return A.col16 ? 1 : 0; // PCG's product streams 16-bit columns
    if (k == "num_cus") return num_cus_;
    if (k == "dist.comm_aborted") return comm_.aborted() ? 1 : 0;
    if (k == "dist.world") return comm_.active() ? comm_.world() : 1;
    // ranks of a REAL RCCL communicator this handle is part of (0: none / the in-process loopback group): a SCALE record
    // that says 1 here measured no collective over xGMI
    if (k == "dist.rccl_ranks_seen") return comm_.is_rccl() ? comm_.world() : 0;
    if (k == "dist.peer_available") return comm_.peer_attached() ? 1 : 0; // this handle's devices map each other's memory
    if (k == "dist.peer_in_use") return (comm_.peer_on() && comm_.peer_halo_ready()) ? 1 : 0;
    if (k == "dist.n_halo") return (double)n_halo();                   // shards: halo entries of this shard's vectors
    if (k == "reorder.active") return reordered_ ? 1 : 0;              // the factorized system is renumbered
    if (k == "reorder.levels") return ro_info_.levels;                 // breadth-first levels of the search
    if (k == "reorder.components") return ro_info_.components;
    if (k == "reorder.isolated") return ro_info_.isolated;
    if (k == "reorder.leftover") return ro_info_.leftover;
    if (k == "reorder.spread_before") return ro_spread_before_;        // device_gather_spread of the caller's numbering
    if (k == "reorder.spread_after") return ro_spread_after_;          // ... of the new one (0: not computed)
    if (k == "reorder.seconds") return ro_seconds_;                    // of the last factorize: search (first time) + permuted copy
    if (k == "schwarz.levels_built") return schwarz_ ? schwarz_->levels() : 0;
    if (k == "ic.shift") return ic_ ? ic_->shift() : 0.0;             // the shift the factorization ended with
    if (k == "ic.attempts") return ic_ ? ic_->attempts() : 0;         // 1 + restarts with a larger shift
    if (k == "ic.levels_backward") return ic_ ? ic_->levels_backward() : 0;
    if (k == "ic.levels") return ic_ ? ic_->levels_forward() : 0;     // dependency depth of the forward solve
    if (k == "amg.last_setup_reused") return damg_ ? (damg_->last_setup_reused() ? 1 : 0) : (amg_ ? (amg_->last_setup_reused() ? 1 : 0) : 0);
    if (k == "amg.levels_aggregated_on_device") return amg_ ? amg_->levels_aggregated_on_device() : 0;
    if (k == "amg.packed_row_block_operators") return amg_ ? amg_->operators_with_packed_row_blocks() : 0;
    if (k == "amg.dist_mode_used") return dist_mode_used_; // what "amg.dist_global" came to at the last factorize on shards
    if (k == "amg.distributed_levels") return damg_ ? damg_->distributed_levels() : 0; // levels whose rows are partitioned
    if (k == "stats.update_r_ms_avg") return k2_ms_avg_;   // sampled iterations of the last solve: pcg_update_r_kernel ...
    if (k == "stats.update_xp_ms_avg") return k3_ms_avg_;  // ... and pcg_update_xp_kernel (Jacobi-PCG's fused loop, one device)
    if (k == "stats.allreduce_us_avg") return ar_us_avg_;   // shards, sampled iterations of the last solve ("profile_spmv")
    if (k == "stats.allreduce_samples") return ar_samples_;
    if (k == "stats.halo_us_avg") return halo_us_avg_;
    if (k == "stats.halo_samples") return halo_samples_;
    if (k == "stats.h2d_bytes") return (double)stats.h2d_bytes;
    if (k == "stats.d2h_bytes") return (double)stats.d2h_bytes;
    if (k == "stats.pattern_uploads") return (double)stats.pattern_uploads; // ... of them with the 4 (n + 1 + nnz) bytes of pattern
    if (k == "stats.matrix_uploads") return (double)stats.matrix_uploads;
    if (k == "stats.reorder_searches") return (double)stats.reorder_searches;
    if (k == "stats.amg_setups") return (double)stats.amg_setups;
    if (k == "stats.amg_refreshes") return (double)stats.amg_refreshes;
    if (k == "stats.solves") return (double)stats.solves;
    if (k == "stats.device_bytes") return (double)meter.bytes.load();      // held by this handle's buffers right now
    if (k == "stats.device_bytes_cached") {
        std::lock_guard<std::mutex> lk(meter_->mu);
        return (double)meter_->cached;
    }
    if (k == "stats.device_bytes_peak") return (double)meter.peak.load();  // ... at most since the handle was created
    double v = 0.0;
    if (param_value(prm, k, &v)) return v;
    throw Error(PSOLVE_HIP_EINVAL, "unknown parameter '" + k + "'");
}

// ---------------------------------------------------------------------------------------------
// analyze_pattern / factorize
// ---------------------------------------------------------------------------------------------
static void check_sizes(int64_t n, int64_t nnz)
{
    PS_REQUIRE(n > 0 && nnz >= 0, PSOLVE_HIP_EINVAL, "empty matrix");
    // int32 per shard, like MAS (BSRMatrix.cu:438-442); +4 leaves room for the SpMV's 4-wide loads
    PS_REQUIRE(n < INT32_MAX - 1024 && nnz < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE,
               "matrix shard exceeds int32 indexing (n or nnz >= 2^31): partition it over more GPUs");
}

void Context::check_sizes_public(int64_t n, int64_t nnz) { check_sizes(n, nnz); }

void Context::analyze_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int precond_num)
{
    const double t0 = wall_seconds();
    use_device();
    check_sizes(n, nnz);
    PS_REQUIRE(outer && (inner || nnz == 0), PSOLVE_HIP_EINVAL, "analyze_pattern: null pattern arrays");
    PS_REQUIRE(outer[0] == 0 && outer[n] == nnz, PSOLVE_HIP_EINVAL,
               "analyze_pattern: outer[0] != 0 or outer[n] != nnz (matrix must be compressed)");
    analyzed_n_ = n;
    analyzed_nnz_ = nnz;
    precond_num_ = precond_num;
    // pre-size the device storage so factorize() only moves bytes
    rowptr_own_.ensure((size_t)n + 1);
    col_own_.ensure((size_t)nnz + 4);
    val_own_.ensure((size_t)nnz + 4);
    info.time_analyze = wall_seconds() - t0;
}

void Context::factorize_host(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner,
                             const double *values)
{
    const double t0 = wall_seconds();
    use_device();
    check_sizes(n, nnz);
    PS_REQUIRE(outer && inner && values, PSOLVE_HIP_EINVAL, "factorize: null matrix arrays");
    PS_REQUIRE(outer[0] == 0 && outer[n] == nnz, PSOLVE_HIP_EINVAL,
               "factorize: outer[0] != 0 or outer[n] != nnz (matrix must be compressed)");
    PS_REQUIRE(!(comm_.active() && comm_.world() > 1), PSOLVE_HIP_EINVAL,
               "factorize(host arrays) is the single-GPU contract; shards use factorize_device");
    factorized_ = false;
    // The values travel first; meanwhile a few host threads hash the pattern.  The same pattern as the one this handle
    // still holds on the device (Newton: every iteration, Newton.cpp:189-193): its 4 (n + 1 + nnz) bytes stay where
    // they are and only the 8 nnz bytes of values cross PCIe ("stats.h2d_bytes").
    HostPatternHash h;
    std::thread hasher([&] { h = hash_host_pattern(n, nnz, outer, inner); });
    try {
        val_own_.ensure((size_t)nnz + 4);
        PS_HIP_CHECK(hipMemcpyAsync(val_own_.ptr, values, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, stream));
    } catch (...) {
        hasher.join();
        throw;
    }
    hasher.join();
    const bool same_pattern = host_pat_resident_ && host_pat_n_ == n && host_pat_nnz_ == nnz && h == host_pat_hash_ &&
                              rowptr_own_.ptr && col_own_.ptr;
    stats.h2d_bytes += nnz * 8;
    if (!same_pattern) {
        host_pat_resident_ = false;
        rowptr_own_.ensure((size_t)n + 1);
        col_own_.ensure((size_t)nnz + 4);
        PS_HIP_CHECK(hipMemcpyAsync(rowptr_own_.ptr, outer, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        PS_HIP_CHECK(hipMemcpyAsync(col_own_.ptr, inner, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += (int64_t)(n + 1) * 4 + nnz * 4;
        ++stats.pattern_uploads;
        host_pat_hash_ = h;
        host_pat_n_ = n;
        host_pat_nnz_ = nnz;
    }
    ++stats.matrix_uploads;
    row_begin_ = 0;
    row_end_ = n;
    n_global_ = n;
    from_host_ = true;
    try {
        factorize_device(n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
    } catch (...) {
        from_host_ = false;
        host_pat_resident_ = false;
        throw;
    }
    from_host_ = false;
    host_pat_resident_ = rowptr_own_.ptr != nullptr && col_own_.ptr != nullptr; // (released where the device has no room for both numberings)
    info.time_factorize = wall_seconds() - t0;
}

void Context::factorize_host_rows(int64_t n_global, int64_t row_begin, int64_t row_end, const int32_t *outer,
                                  const int32_t *inner, const double *values)
{
    const double t0 = wall_seconds();
    use_device();
    PS_REQUIRE(outer && inner && values, PSOLVE_HIP_EINVAL, "factorize: null matrix arrays");
    set_partition(n_global, row_begin, row_end);
    const int64_t n = row_end - row_begin, k0 = outer[row_begin], nnz = (int64_t)outer[row_end] - k0;
    check_sizes(n, nnz);
    rowptr_own_.ensure((size_t)n + 1);
    col_own_.ensure((size_t)nnz + 4);
    val_own_.ensure((size_t)nnz + 4);
    std::vector<int32_t> ptr((size_t)n + 1); // row pointers of the slice, rebased to 0
    for (int64_t i = 0; i <= n; ++i) ptr[(size_t)i] = (int32_t)(outer[row_begin + i] - k0);
    PS_HIP_CHECK(hipMemcpyAsync(rowptr_own_.ptr, ptr.data(), (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipMemcpyAsync(col_own_.ptr, inner + k0, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipMemcpyAsync(val_own_.ptr, values + k0, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream)); // `ptr` is pageable and dies with this frame
    stats.h2d_bytes += (int64_t)(n + 1) * 4 + nnz * 12;
    ++stats.matrix_uploads;
    factorize_device(n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
    info.time_factorize = wall_seconds() - t0;
}

void Context::factorize_device(int64_t n_local, int64_t nnz_local, const int32_t *d_rowptr, const int32_t *d_col,
                               const double *d_values, bool owned)
{
    const double t0 = wall_seconds();
    use_device();
    check_sizes(n_local, nnz_local);
    // PSOLVE_TIMING: where a factorize spends its time outside the preconditioner's setup (every lap synchronises)
    const bool timing = std::getenv("PSOLVE_TIMING") != nullptr;
    double t_lap = t0;
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(stream);
        const double t1 = wall_seconds();
        std::fprintf(stderr, "[psolve timing] factorize %-28s %.4f s\n", what, t1 - t_lap);
        t_lap = t1;
    };
    PS_REQUIRE(d_rowptr && d_col && d_values, PSOLVE_HIP_EINVAL, "factorize_device: null device arrays");
    PS_REQUIRE(((uintptr_t)d_col % 16) == 0 && ((uintptr_t)d_values % 16) == 0, PSOLVE_HIP_EINVAL,
               "factorize_device: col/values must be 16-byte aligned");
    factorized_ = false;
    if (loop_graph_) { // the captured loop bakes in matrix / workspace pointers
        (void)hipGraphExecDestroy(loop_graph_);
        loop_graph_ = nullptr;
    }
    const bool dist = comm_.active();
    if (!from_host_) host_pat_resident_ = false; // (the own buffers are about to hold, or have held, something else)
    // "reorder": the products, the preconditioner and the PCG vectors live in a locality numbering; b and x are permuted
    // on the way in and out (solve_device).  Shards keep the caller's numbering (the partition is by its rows).
    reordered_ = false;
    ro_called_ = false;
    if (prm.reorder > 0 && !dist && reorder_matrix(n_local, nnz_local, d_rowptr, d_col, d_values)) {
        reordered_ = true;
        if (owned && d_rowptr == rowptr_own_.ptr && d_col == col_own_.ptr && d_values == val_own_.ptr) {
            // the handle's own copy in the caller's numbering (uploaded or generated) has served.  A generated system keeps
            // one copy of the matrix resident, not two.  A system that came through the HOST contract keeps the caller's
            // numbering as well where the device has room to spare (a quarter of what is free): the next factorize of
            // the same pattern then uploads 8 nnz bytes into the buffer that is already there instead of 12 nnz bytes
            // into three fresh allocations (hipFree synchronises the device).
            size_t free_b = 0, total_b = 0;
            const bool keep = from_host_ && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                              (size_t)(12 * nnz_local + 4 * n_local) < free_b / 4;
            if (!keep) {
                rowptr_own_.release();
                col_own_.release();
                val_own_.release();
            }
        }
        d_rowptr = ro_ptr_.ptr;
        d_col = ro_col_.ptr;
        d_values = ro_val_.ptr;
    } else { // (a renumbered copy kept from an earlier factorize)
        ro_ptr_.release();
        ro_col_.release();
        ro_val_.release();
        ro_b_.release();
        ro_x_.release();
    }
    if (!owned) {
        rowptr_own_.release();
        val_own_.release();
        if (!dist) col_own_.release(); // shards keep it: the local-id copy of the adopted column array
    }
    if (!dist && (row_end_ - row_begin_ != n_local || n_global_ != n_local)) {
        row_begin_ = 0;
        row_end_ = n_local;
        n_global_ = n_local;
    }
    PS_REQUIRE(row_end_ - row_begin_ == n_local, PSOLVE_HIP_EINVAL,
               "factorize_device: n_local does not match the partition set by set_partition");
    lap("reorder decision / copy");
    A.n = (int)n_local;
    A.nnz = nnz_local;
    A.rowptr = d_rowptr;
    A.col = d_col;
    A.val = d_values;
    A.rows_per_block = prm.spmv_rows_per_block ? prm.spmv_rows_per_block
                                               : spmv_rows_per_block((double)nnz_local / (double)n_local);
    refit_launch();
    setup_halo(d_col, owned);
    ensure_workspace();
    if (dist) classify_row_blocks();
    lap("launch fit, halo, workspace");
    // the pattern's identity, once for everybody who keeps symbolic work (see pattern_id_of_A)
    if (!dist) {
        const bool sizes = a_hash_n_ == n_local && a_hash_nnz_ == nnz_local && a_hash_reordered_ == reordered_;
        if (ro_called_ && reordered_ && ro_same_last_ && sizes && a_hash_ != 0) {
            a_same_ = true; // same caller's pattern, same kept order: the renumbered pattern is the one we hashed before
        } else {
            unsigned long long h[2];
            if (ro_called_ && !reordered_) { // A IS the caller's arrays, which reorder_matrix has just hashed
                h[0] = ro_hash_last_[0];
                h[1] = ro_hash_last_[1];
            } else {
                ro_hash_dev_.ensure(2);
                PS_HIP_CHECK(hipMemsetAsync(ro_hash_dev_.ptr, 0, 2 * sizeof(unsigned long long), stream));
                launch_hash_i32(L_, (int64_t)A.n + 1, A.rowptr, ro_hash_dev_.ptr);
                launch_hash_i32(L_, A.nnz, A.col, ro_hash_dev_.ptr + 1);
                PS_HIP_CHECK(hipMemcpyAsync(h, ro_hash_dev_.ptr, sizeof(h), hipMemcpyDeviceToHost, stream));
                PS_HIP_CHECK(hipStreamSynchronize(stream));
            }
            unsigned long long id = h[0] * 0x9E3779B97F4A7C15ull + h[1];
            if (id == 0) id = 1;
            a_same_ = sizes && id == a_hash_;
            a_hash_ = id;
        }
        a_hash_n_ = n_local;
        a_hash_nnz_ = nnz_local;
        a_hash_reordered_ = reordered_;
    } else {
        a_same_ = false;
        a_hash_ = 0;
    }

    lap("pattern identity");
    // Jacobi: Eigen::DiagonalPreconditioner::factorize semantics; a non-finite diagonal is a
    // factorization failure (-> std::runtime_error in the adapter, caught by Newton.cpp:195)
    // (round 6: only where Jacobi is the preconditioner -- the pass reads every column index, 1.7 ms of configs[2]'s 22 ms
    // refresh; another preconditioner checks what IT inverts, and a later switch to Jacobi computes it at the first use)
    invdiag_valid_ = false;
    if (prm.precond == 1 || dist) {
        PS_HIP_CHECK(hipMemsetAsync(flags_.ptr, 0, 4 * sizeof(int), stream));
        launch_diag_inverse(L_, A, invdiag_.ptr, flags_.ptr);
        int bad = 0;
        PS_HIP_CHECK(hipMemcpyAsync(&bad, flags_.ptr, sizeof(int), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        shards_agree(bad == 0, PSOLVE_HIP_ENUMERIC, "factorize: " + std::to_string(bad) + " non-finite diagonal entries");
        invdiag_valid_ = true;
    }

    lap("inverse diagonal");
    A.bsr3 = nullptr;
    if (prm.block_size == 3 && prm.use_bsr3 && !dist) build_bsr3();
    // rows that repeat a few column-offset patterns -- stencils, FEM on structured meshes --: the products drop the
    // column stream (8 nnz + 22 n bytes instead of 12 nnz + 20 n); operators without such a dictionary (unstructured
    // meshes) keep the plain CSR stream.  "spmv_kernel" 0 / 1 / 2 switch it off
    A.pat = nullptr;
    const bool want_pat = !A.bsr3 && A.rows_per_block >= 64 && (prm.spmv_kernel == 3 || prm.spmv_kernel < 0);
    // (the dictionary is a function of the pattern: kept across factorizes of the same one -- Newton.cpp:189-193 --,
    // valid or not: an operator that had none does not grow one with new values)
    // (round-4 advice: every kept piece of symbolic work carries the id of the pattern it was built for -- "the same
    // pattern as the PREVIOUS factorize call" is not that: a call that failed, or did not rebuild this cache, lies between)
    bool pat_kept = false;
    if (want_pat && a_hash_ != 0 && pat_id_ == a_hash_ && pat_n_ == A.n && pat_tried_) {
        if (pat_.valid) A.pat = &pat_.view;
        pat_kept = pat_.valid;
    } else {
        pat_.reset();
        pat_tried_ = false;
        pat_id_ = 0;
        if (want_pat) {
            if (pat_.build(L_, A)) A.pat = &pat_.view;
            pat_tried_ = true;
            pat_n_ = A.n;
            pat_id_ = a_hash_;
        }
    }
    // ... and rows that also repeat their VALUES bit for bit (a constant-coefficient stencil): row kinds, no matrix stream at
    // all.  A function of the values: rebuilt by every factorize, dropped when the rows do not repeat
    if (A.pat) {
        Launch Lk = L_;
        Lk.stream = stream;
        // (rows of any length the dictionary takes: spmv_csr_kind gives a row to a lane whatever the row-block height of the
        // streaming kernels -- a 27-point operator's 64-row blocks are theirs, not its)
        if (!(prm.spmv_value_dict && prm.spmv_kernel < 0 && pat_.build_values(Lk, A, pat_kept)))
            pat_.drop_values();
    }
    // wide rows without a block copy and without a dictionary (>= 12 stored entries per row: Q1 elasticity as CSR,
    // higher-order FEM): PCG's product runs on a SELL-64-sigma copy; "spmv_kernel" 2 forces it
    A.sell = nullptr;
    if (!dist && (prm.spmv_kernel == 2 || (prm.spmv_kernel < 0 && !A.pat && !A.bsr3 && A.n >= 4096 && A.nnz >= 12ll * A.n))) {
        sell_.build(L_, A, bsr_scratch_, prm.spmv_kernel == 2 ? 4.0 : 1.25);
        if (sell_.valid) A.sell = &sell_.view;
    }
    // every other operator in a local numbering (a renumbered unstructured mesh, a grid whose dictionary is off, a
    // shard): 16-bit columns through eight 8192-column windows per row-block, where every row-block fits them
    A.col16 = nullptr;
    A.rb_base = nullptr;
    A.col16_R = 0;
    if (prm.spmv_col16 && !A.pat && !A.bsr3 && !A.sell && A.n >= 4096 && prm.spmv_kernel != 0) {
        Launch Lc = L_;
        Lc.stream = stream;
        if (col16_.build(Lc, A)) col16_.attach(A);
        else col16_.reset();
    } else {
        col16_.reset();
    }
    // Jacobi's inverse diagonal is constant within every row kind (rows of a kind have the same diagonal entry): the fused
    // vector kernels read it as table[kind[row]] (Launch::kd_*), 2 bytes per row instead of 8; verified against every row
    kdinv_valid_ = false;
    if (A.pat && A.pat->kind && prm.spmv_value_dict && invdiag_valid_) {
        Launch Lk = L_;
        Lk.stream = stream;
        kdinv_valid_ = pat_.build_row_table(Lk, A.n, invdiag_.ptr, kdinv_);
    }
    refit_launch(); // the product kernel is known now: the dictionary kernel takes a larger grid
    lap("block copy / dictionary / sell / col16");

    info.amg_levels = 0;
    // the preconditioner setup may fail on one shard only (a singular diagonal block, ...): agree before returning
    int fail_code = 0;
    std::string fail_msg;
    try {
    if (prm.precond == 2) {
        if (!amg_) amg_.reset(new AmgHierarchy());
        prm.amg.block_size = prm.block_size;
        prm.amg.col16 = prm.spmv_col16;
        PS_REQUIRE(prm.block_size == 1 || A.n % prm.block_size == 0, PSOLVE_HIP_EINVAL,
                   "block_size does not divide the matrix size");
        bool global_done = false;
        int dist_mode = (dist && comm_.world() > 1) ? prm.amg.dist_global : 0;
        if (dist_mode == 1 && prm.block_size > 1) dist_mode = 2; // (the replicated setup serves scalar systems)
        if (dist_mode == 2 && prm.block_size > 1)
            for (int64_t o : plan_.row_offsets) // (known to every rank alike: they all decide the same)
                if (o % prm.block_size != 0) dist_mode = 0; // a partition that cuts through a node: per-shard hierarchies
        // the distributed setup serves eps_strong = 0; a scalar system with another threshold takes the replicated
        // single-device hierarchy where that fits (round 2's construction), and only then one hierarchy per shard
        const bool want_replicated = dist_mode == 1 || (dist_mode == 2 && prm.amg.eps_strong != 0.0 && prm.block_size <= 1);
        if (want_replicated) {
            // the replicated setup gathers the WHOLE matrix on every rank: only for systems that fit comfortably --
            // decided from the global size (every rank computes the same sum), before anybody gathers anything
            scal_.ensure(S_COUNT);
            scal_host_.ensure(S_COUNT);
            const double mine = 12.0 * (double)A.nnz + 4.0 * (double)A.n;
            PS_HIP_CHECK(hipMemcpyAsync(scal_.ptr + S_TMP + 2, &mine, sizeof(double), hipMemcpyHostToDevice, stream));
            comm_.allreduce_sum(scal_.ptr + S_TMP + 2, 1, stream);
            PS_HIP_CHECK(hipMemcpyAsync(scal_host_.ptr + 2, scal_.ptr + S_TMP + 2, sizeof(double), hipMemcpyDeviceToHost, stream));
            PS_HIP_CHECK(hipStreamSynchronize(stream));
            const double gbytes = scal_host_.ptr[2];
            const bool fits = !(gbytes > (double)prm.amg.dist_global_max_mbytes * 1048576.0 || gbytes / 12.0 >= 2.0e9);
            if (dist_mode == 1) dist_mode = fits ? 1 : 2;
            else dist_mode = fits ? 1 : 0;
        }
        if (dist_mode == 2 && prm.amg.eps_strong != 0.0) dist_mode = 0;
        dist_mode_used_ = dist_mode;
        if (dist_mode != 2) damg_.reset();
        else if (!damg_) damg_.reset(new DistAmg());
        if (dist_mode == 2) {
            prm.amg.block_size = prm.block_size;
            damg_->setup(*this, prm.amg);
            amg_.reset();
            info.amg_levels = damg_->levels();
            ++(damg_->last_setup_reused() ? stats.amg_refreshes : stats.amg_setups);
            global_done = true;
        } else if (dist_mode == 1) {
            // shards, scalar systems: ONE hierarchy for the whole matrix.  Every rank gathers the matrix and runs the
            // single-device setup on it -- the aggregates, P and the Galerkin operators are then exactly the
            // single-device ones, and so are the iteration counts --, keeps its rows of level 0 (A with halo, P_0,
            // the transpose as its share of R_0) and all of the coarser levels, replicated.  Level 0, 60 % of the
            // cycle's work, scales with the ranks; the coarser levels cost every rank what they cost one device.
            int64_t gnnz = 0;
            gather_global_matrix(glob_ptr_, glob_col_, glob_val_, gnnz);
            CsrDev Ag;
            Ag.n = (int)n_global_;
            Ag.n_ext = (int)n_global_;
            Ag.nnz = gnnz;
            Ag.rowptr = glob_ptr_.ptr;
            Ag.col = glob_col_.ptr;
            Ag.val = glob_val_.ptr;
            Ag.rows_per_block = spmv_rows_per_block((double)gnnz / (double)std::max<int64_t>(1, n_global_));
            amg_->setup_global(*this, Ag, (int)row_begin_, A.n, prm.amg);
            global_done = amg_->global_on_shards();
        }
        if (damg_) {
        } else if (global_done) {
        } else if (dist) {
            // shards: non-overlapping additive Schwarz -- every rank builds the AMG hierarchy of ITS diagonal
            // block (halo columns dropped) and applies it to its slice of the residual, no communication in
            // the preconditioner; block-diagonal of SPD pieces, so PCG stays valid
            Launch L = L_;
            L.stream = stream;
            const int64_t lnnz = device_diagonal_block(L, A, loc_ptr_, loc_col_, loc_val_, bsr_scratch_);
            CsrDev Aloc;
            Aloc.n = A.n;
            Aloc.n_ext = A.n;
            Aloc.nnz = lnnz;
            Aloc.rowptr = loc_ptr_.ptr;
            Aloc.col = loc_col_.ptr;
            Aloc.val = loc_val_.ptr;
            Aloc.rows_per_block = spmv_rows_per_block((double)lnnz / (double)std::max(1, A.n));
            amg_->setup(*this, Aloc, prm.amg);
        } else {
            amg_->setup(*this, A, prm.amg);
        }
        if (!damg_) {
            info.amg_levels = amg_->levels();
            ++(amg_->last_setup_reused() ? stats.amg_refreshes : stats.amg_setups);
        }
    } else {
        // a hierarchy kept from an earlier factorize describes another matrix: level 0 aliases arrays that
        // may be gone and the level vectors have the old size.  Selecting precond = amg later must not find it.
        amg_.reset();
        damg_.reset();
    }
    if (prm.precond == 4) {
        // incomplete Cholesky (ic.hpp).  On shards: of the shard's diagonal block (block Jacobi of incomplete factors:
        // no communication in the preconditioner, SPD, so PCG stays valid)
        if (!ic_) ic_.reset(new IcPrecond());
        if (dist) {
            Launch L = L_;
            L.stream = stream;
            const int64_t lnnz = device_diagonal_block(L, A, loc_ptr_, loc_col_, loc_val_, bsr_scratch_);
            CsrDev Aloc;
            Aloc.n = A.n;
            Aloc.n_ext = A.n;
            Aloc.nnz = lnnz;
            Aloc.rowptr = loc_ptr_.ptr;
            Aloc.col = loc_col_.ptr;
            Aloc.val = loc_val_.ptr;
            ic_->setup(*this, Aloc, prm.ic_initial_shift, prm.ic_ordering);
        } else {
            ic_->setup(*this, A, prm.ic_initial_shift, prm.ic_ordering);
        }
    } else {
        ic_.reset();
    }
    if (prm.precond == 3) {
        // multilevel additive Schwarz on 64-unknown domains (schwarz.hip).  On shards the domains and their
        // coarse levels stay inside the shard: halo columns are ignored by the assembly.
        if (!schwarz_) schwarz_.reset(new SchwarzPrecond());
        schwarz_->setup(*this, A, prm.schwarz_levels, prm.block_size);
    } else {
        schwarz_.reset();
    }
    } catch (const Error &e) {
        if (!dist) throw;
        fail_code = e.code;
        fail_msg = e.what();
    }
    if (dist) shards_agree(fail_code == 0, fail_code, fail_msg);
    lap("preconditioner setup");
    factorized_ = true;
    info.time_factorize = wall_seconds() - t0;
    if (timing) std::fprintf(stderr, "[psolve timing] factorize total %.4f s\n", info.time_factorize);
}

// block_size 3: a zero-filled 3x3-block copy of the matrix, built on the device (block columns by the
// row-set kernels of amg_symbolic.hip, values by a kernel), so that the fine-level products run on 76 B
// per 9 entries instead of 108 B
void Context::build_bsr3()
{
    PS_REQUIRE(A.n % 3 == 0, PSOLVE_HIP_EINVAL, "block_size does not divide the matrix size");
    Launch L = L_;
    L.stream = stream;
    // the block graph is symbolic work: kept while the pattern stays the same (Newton.cpp:189-193)
    const bool keep = a_hash_ != 0 && bsr_graph_id_ == a_hash_ && bsr_graph_n_ == A.n && bsr_graph_.b == 3 &&
                      bsr_graph_.nb == A.n / 3 && bsr_graph_.ptr.ptr && bsr_graph_.col.ptr && bsr_graph_.nnzb > 0;
    if (!keep) bsr_graph_id_ = 0; // (a rebuild that throws half-way leaves no graph that claims a pattern)
    const int64_t nnzb = keep ? bsr_graph_.nnzb : device_block_graph(L, A, 3, bsr_graph_, bsr_scratch_);
    bsr_graph_n_ = A.n;
    bsr_graph_id_ = a_hash_;
    PS_REQUIRE(nnzb * 9 < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "BSR-3 copy exceeds int32 indexing");
    device_block_values(L, A, bsr_graph_);
    bsr_.nb = bsr_graph_.nb;
    bsr_.nnzb = nnzb;
    bsr_.rowptr = bsr_graph_.ptr.ptr;
    bsr_.col = bsr_graph_.col.ptr;
    bsr_.val = bsr_graph_.val.ptr;
    bsr_.brows_per_group = bsr3_brows_per_group((double)nnzb / (double)std::max(1, bsr_graph_.nb));
    // block rows that repeat offsets and values bit for bit (constant-coefficient elasticity on a structured mesh): block-row
    // kinds, no matrix stream.  A function of the values: rebuilt by every factorize, absent where block rows do not repeat
    bsr_.kinds = nullptr;
    if (prm.spmv_value_dict && prm.spmv_kernel < 0 && bsr_kinds_.build(L, bsr_, keep)) bsr_.kinds = &bsr_kinds_.view;
    A.bsr3 = &bsr_;
}

void Context::shards_agree(bool ok, int code, const std::string &msg)
{
    if (comm_.active() && comm_.world() > 1) {
        scal_.ensure(S_COUNT);
        scal_host_.ensure(S_COUNT);
        const double mine = ok ? 0.0 : 1.0;
        PS_HIP_CHECK(hipMemcpyAsync(scal_.ptr + S_TMP + 1, &mine, sizeof(double), hipMemcpyHostToDevice, stream));
        comm_.allreduce_sum(scal_.ptr + S_TMP + 1, 1, stream);
        PS_HIP_CHECK(hipMemcpyAsync(scal_host_.ptr + 1, scal_.ptr + S_TMP + 1, sizeof(double), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        if (ok && scal_host_.ptr[1] > 0.0) throw Error(PSOLVE_HIP_ECOMM, "factorize failed on another shard", true);
        if (!ok) throw Error(code, msg, true); // (every rank leaves here: the collective sequence stays aligned)
    }
    if (!ok) throw Error(code, msg);
}

void Context::refit_launch()
{
    L_ = fit_launch(Lmax_, A.n, A.rows_per_block, (spmv_grid_user_set_ || A.n <= 0) ? 0.0 : (double)A.nnz / A.n);
    // PCG's own vector kernels run on their own, smaller persistent grid ("vec_blocks_per_cu", solver.hpp)
    L_.grid = std::max(8, std::min(L_.grid, (num_cus_ * prm.vec_blocks_per_cu + 7) & ~7));
    if (A.pat && (prm.spmv_kernel < 0 || prm.spmv_kernel == 3) && !spmv_grid_user_set_) {
        // the dictionary kernel: 16 KiB + the dictionary of LDS per workgroup instead of 24.6 KiB, 8 workgroups per
        // CU are its optimum (256^3: 0.227 / 0.219 / 0.263 ms with 6 / 8 / 9) -- unless the caller has chosen a grid
        Launch m = Lmax_;
        m.spmv_grid = std::min(kMaxPartials, (num_cus_ * 8 + 7) & ~7);
        L_.spmv_grid = fit_launch(m, A.n, A.rows_per_block).spmv_grid;
    }
    // one verdict for the whole iteration: when the operator is streamed non-temporally (too large for the
    // Infinity Cache), so are the vectors of the fused kernels -- otherwise the dirty lines one kernel leaves
    // behind are written back in the middle of the next kernel's read stream -- once the five vectors an iteration
    // touches no longer fit the cache together (8 n >= 64 MiB; round 4: 216^3, 77 MiB each, Jacobi-PCG 115-117 -> 109-110 ms;
    // 192^3, 54 MiB each: even; profiles/r04_nt_crossover.txt)
    L_.kd_kind = nullptr;
    L_.kd_tab = L_.kd_for = nullptr;
    L_.kd_n = 0;
    if (kdinv_valid_ && A.pat && A.pat->kind && A.pat->nkind <= kKindTabMax) {
        L_.kd_kind = A.pat->kind;
        L_.kd_tab = kdinv_.ptr;
        L_.kd_for = invdiag_.ptr;
        L_.kd_n = A.pat->nkind;
    }
    const int64_t bytes = A.nnz * 12 + 20ll * A.n;
    L_.vec_nt = prm.spmv_nt == 1 ||
                (prm.spmv_nt < 0 && bytes > ((int64_t)prm.spmv_nt_mbytes << 20) && 8ll * A.n >= (64ll << 20));
}

void Context::ensure_workspace()
{
    const size_t n = (size_t)A.n, ne = (size_t)A.n_ext;
    invdiag_.ensure(n);
    r_.ensure(n + 2);
    q_.ensure(n + 2);
    p_ext_.ensure(ne + 2);
    t_ext_.ensure(ne + 2);
    if (prm.precond >= 2) z_.ensure(n + 2);
    partials_.ensure((size_t)P_COUNT * kMaxPartials);
    scal_.ensure(S_COUNT);
    state_.ensure(1);
    flags_.ensure(8);
    state_host_.ensure(4);
    scal_host_.ensure(S_COUNT);
}

// ---------------------------------------------------------------------------------------------
// Distributed: partition, halo plan, halo exchange
// ---------------------------------------------------------------------------------------------
void Context::comm_init(int rank, int world, const char *id, const char *rccl_path)
{
    use_device();
    comm_.init(rank, world, id, rccl_path);
    factorized_ = false;
}

void Context::comm_init_local(LocalGroup *g, int rank)
{
    use_device();
    comm_.init_local(g, rank);
    factorized_ = false;
}

void Context::set_partition(int64_t n_global, int64_t row_begin, int64_t row_end)
{
    PS_REQUIRE(n_global > 0 && row_begin >= 0 && row_begin < row_end && row_end <= n_global, PSOLVE_HIP_EINVAL,
               "set_partition: need 0 <= row_begin < row_end <= n_global");
    PS_REQUIRE(n_global < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "global size exceeds int32 column ids");
    n_global_ = n_global;
    row_begin_ = row_begin;
    row_end_ = row_end;
    factorized_ = false;
}

// Shards: the column ids arrive GLOBAL and the kernels want LOCAL ids in [0, n + n_halo).  The remap is
// written into storage this handle owns -- in place for matrices it uploaded or generated itself, into a
// private copy for arrays adopted from the caller (psolve_hip_factorize_device), which stay untouched, so
// that a second factorize with the same arrays (Newton, constant pattern) sees global ids again.
void Context::setup_halo(const int32_t *d_col, bool owned)
{
    const bool dist = comm_.active();
    const int row0 = (int)row_begin_, row1 = (int)row_end_;
    flags_.ensure(8);
    PS_HIP_CHECK(hipMemsetAsync(flags_.ptr, 0, 8 * sizeof(int), stream));
    launch_offrange_count(L_, A.nnz, d_col, row0, row1, flags_.ptr);
    int n_off = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&n_off, flags_.ptr, sizeof(int), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    plan_ = HaloPlan();
    if (!dist) {
        PS_REQUIRE(n_off == 0, PSOLVE_HIP_EINVAL,
                   "factorize: " + std::to_string(n_off) + " column ids outside [0, n) (and no partition/communicator set)");
        A.n_ext = A.n;
        return;
    }
    const int world = comm_.world(), rank = comm_.rank();
    // 1. everyone learns the partition
    DeviceBuffer<int64_t> d_i64;
    d_i64.ensure((size_t)world * (world + 2));
    std::vector<int64_t> h_i64((size_t)world * (world + 2));
    int64_t my_begin = row_begin_;
    PS_HIP_CHECK(hipMemcpyAsync(d_i64.ptr + world, &my_begin, sizeof(int64_t), hipMemcpyHostToDevice, stream));
    comm_.allgather_i64(d_i64.ptr + world, d_i64.ptr, 1, stream);
    PS_HIP_CHECK(hipMemcpyAsync(h_i64.data(), d_i64.ptr, (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    plan_.rank = rank;
    plan_.world = world;
    plan_.row_offsets.assign(h_i64.begin(), h_i64.begin() + world);
    plan_.row_offsets.push_back(n_global_);
    PS_REQUIRE(plan_.row_offsets[rank] == row_begin_ && plan_.row_offsets[rank + 1] == row_end_, PSOLVE_HIP_EINVAL,
               "set_partition: partitions of the ranks are not contiguous in rank order");

    // 2. off-shard column ids -> sorted unique halo list + owner counts
    std::vector<int32_t> off((size_t)n_off);
    if (n_off > 0) {
        DeviceBuffer<int> d_off;
        d_off.ensure((size_t)n_off);
        launch_offrange_collect(L_, A.nnz, d_col, row0, row1, d_off.ptr, flags_.ptr + 1);
        PS_HIP_CHECK(hipMemcpyAsync(off.data(), d_off.ptr, (size_t)n_off * sizeof(int), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
    }
    if (prm.block_size > 1 && n_global_ % prm.block_size == 0) {
        // block value types: the halo consists of whole nodes (all block_size scalar columns of a node; the partition
        // is cut at block multiples, so they share an owner) -- block views of the shard's operator stay aligned
        const int bs = prm.block_size;
        std::vector<int32_t> whole;
        whole.reserve(off.size() * (size_t)bs);
        for (int32_t g : off)
            for (int c = 0; c < bs; ++c) whole.push_back(g / bs * bs + c);
        off.swap(whole);
    }
    plan_halo(rank, world, plan_.row_offsets.data(), (int64_t)off.size(), off.data(), plan_.halo, plan_.recv_counts);
    const int n_halo = (int)plan_.halo.size();
    plan_.recv_offsets.assign((size_t)world, 0);
    for (int q = 1; q < world; ++q) plan_.recv_offsets[q] = plan_.recv_offsets[q - 1] + plan_.recv_counts[q - 1];
    halo_dev_.ensure((size_t)n_halo + 1);
    if (n_halo)
        PS_HIP_CHECK(hipMemcpyAsync(halo_dev_.ptr, plan_.halo.data(), (size_t)n_halo * sizeof(int), hipMemcpyHostToDevice, stream));
    int32_t *d_loc = col_own_.ptr;
    if (!owned) {
        col_own_.ensure((size_t)A.nnz + 4);
        d_loc = col_own_.ptr;
        PS_HIP_CHECK(hipMemcpyAsync(d_loc, d_col, (size_t)A.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        A.col = d_loc;
    }
    PS_REQUIRE(d_loc != nullptr && A.col == d_loc, PSOLVE_HIP_EINVAL, "setup_halo: no owned column storage");
    launch_remap_cols(L_, A.nnz, d_loc, row0, row1, A.n, halo_dev_.ptr, n_halo);
    A.n_ext = A.n + n_halo;

    // 3. counts[src * world + dst] = entries src needs from dst
    PS_HIP_CHECK(hipMemcpyAsync(d_i64.ptr + (size_t)world * world, plan_.recv_counts.data(), (size_t)world * sizeof(int64_t),
                                hipMemcpyHostToDevice, stream));
    comm_.allgather_i64(d_i64.ptr + (size_t)world * world, d_i64.ptr, world, stream);
    PS_HIP_CHECK(hipMemcpyAsync(h_i64.data(), d_i64.ptr, (size_t)world * world * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    plan_.send_counts.assign((size_t)world, 0);
    plan_.send_offsets.assign((size_t)world, 0);
    for (int q = 0; q < world; ++q) plan_.send_counts[q] = h_i64[(size_t)q * world + rank];
    for (int q = 1; q < world; ++q) plan_.send_offsets[q] = plan_.send_offsets[q - 1] + plan_.send_counts[q - 1];
    plan_.n_send = plan_.send_offsets[world - 1] + plan_.send_counts[world - 1];

    // 4. tell every owner which of its rows we need; receive the same from our dependants
    send_idx_.ensure((size_t)plan_.n_send + 1);
    send_buf_.ensure((size_t)plan_.n_send + 1);
    comm_.exchange_i32(halo_dev_.ptr, plan_.recv_counts, plan_.recv_offsets, send_idx_.ptr, plan_.send_counts,
                       plan_.send_offsets, stream);
    std::vector<int32_t> req((size_t)plan_.n_send);
    if (plan_.n_send) {
        PS_HIP_CHECK(hipMemcpyAsync(req.data(), send_idx_.ptr, (size_t)plan_.n_send * sizeof(int), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        for (int32_t &g : req) {
            PS_REQUIRE(g >= row0 && g < row1, PSOLVE_HIP_ECOMM, "halo request for a row this rank does not own");
            g -= row0;
        }
        PS_HIP_CHECK(hipMemcpyAsync(send_idx_.ptr, req.data(), (size_t)plan_.n_send * sizeof(int), hipMemcpyHostToDevice, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
    }
    // peer-mapped halo exchange of the in-process handle: every shard publishes where it expects whose entries
    if (comm_.peer_attached()) comm_.peer_prepare_halo(plan_, stream);
}

void Context::gather_global_matrix(DeviceBuffer<int> &gptr, DeviceBuffer<int> &gcol, DeviceBuffer<double> &gval, int64_t &gnnz)
{
    const int W = comm_.world(), me = comm_.rank();
    PS_REQUIRE(comm_.active() && (int)plan_.row_offsets.size() == W + 1, PSOLVE_HIP_EINVAL, "gather_global_matrix: no partition");
    // 1. stored entries per rank
    DeviceBuffer<int64_t> d_cnt;
    d_cnt.ensure((size_t)W + 1);
    std::vector<int64_t> cnt((size_t)W), off((size_t)W + 1, 0);
    const int64_t mine = A.nnz;
    PS_HIP_CHECK(hipMemcpyAsync(d_cnt.ptr + W, &mine, sizeof(int64_t), hipMemcpyHostToDevice, stream));
    comm_.allgather_i64(d_cnt.ptr + W, d_cnt.ptr, 1, stream);
    PS_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.ptr, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    for (int q = 0; q < W; ++q) off[(size_t)q + 1] = off[(size_t)q] + cnt[(size_t)q];
    gnnz = off[(size_t)W];
    PS_REQUIRE(gnnz < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE,
               "amg.dist_global: the whole matrix does not fit int32 indexing on one device; set amg.dist_global = 0");
    gptr.ensure((size_t)n_global_ + 1);
    gcol.ensure((size_t)gnnz + 4);
    gval.ensure((size_t)gnnz + 4);
    // 2. my rows into place: row pointers shifted to global positions, columns back to global ids
    PS_HIP_CHECK(hipMemcpyAsync(gptr.ptr + row_begin_, A.rowptr, (size_t)A.n * sizeof(int), hipMemcpyDeviceToDevice, stream));
    launch_add_offset_i32(L_, A.n, gptr.ptr + row_begin_, (int)off[(size_t)me]);
    launch_unmap_cols(L_, A.nnz, A.col, (int)row_begin_, A.n, halo_dev_.ptr, gcol.ptr + off[(size_t)me]);
    PS_HIP_CHECK(hipMemcpyAsync(gval.ptr + off[(size_t)me], A.val, (size_t)A.nnz * sizeof(double), hipMemcpyDeviceToDevice, stream));
    const int last = (int)gnnz;
    PS_HIP_CHECK(hipMemcpyAsync(gptr.ptr + n_global_, &last, sizeof(int), hipMemcpyHostToDevice, stream));
    // 3. everybody sends its slice to everybody (in place: the regions are disjoint)
    std::vector<int64_t> sc((size_t)W, 0), so((size_t)W, 0), rc((size_t)W, 0), ro((size_t)W, 0);
    for (int q = 0; q < W; ++q) {
        if (q == me) continue;
        sc[(size_t)q] = A.n;
        so[(size_t)q] = row_begin_;
        rc[(size_t)q] = plan_.row_offsets[(size_t)q + 1] - plan_.row_offsets[(size_t)q];
        ro[(size_t)q] = plan_.row_offsets[(size_t)q];
    }
    comm_.exchange_i32(gptr.ptr, sc, so, gptr.ptr, rc, ro, stream);
    for (int q = 0; q < W; ++q) {
        if (q == me) continue;
        sc[(size_t)q] = A.nnz;
        so[(size_t)q] = off[(size_t)me];
        rc[(size_t)q] = cnt[(size_t)q];
        ro[(size_t)q] = off[(size_t)q];
    }
    comm_.exchange_i32(gcol.ptr, sc, so, gcol.ptr, rc, ro, stream);
    comm_.exchange_f64(gval.ptr, sc, so, gval.ptr, rc, ro, stream);
    PS_HIP_CHECK(hipStreamSynchronize(stream));
}

void Context::export_halo_link(HaloLink &out)
{
    out.plan = plan_;
    out.n_local = A.n;
    const size_t nh = plan_.halo.size(), ns = (size_t)plan_.n_send;
    out.halo_dev.ensure(nh + 1);
    out.send_idx.ensure(ns + 1);
    out.send_buf.ensure(ns + 1);
    out.send_buf_i.ensure(ns + 1);
    if (nh) PS_HIP_CHECK(hipMemcpyAsync(out.halo_dev.ptr, halo_dev_.ptr, nh * sizeof(int), hipMemcpyDeviceToDevice, stream));
    if (ns) PS_HIP_CHECK(hipMemcpyAsync(out.send_idx.ptr, send_idx_.ptr, ns * sizeof(int), hipMemcpyDeviceToDevice, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
}

void Context::exchange_halo(double *d_ext) { exchange_halo_on(d_ext, stream); }

void Context::exchange_halo_on(double *d_ext, hipStream_t s)
{
    if (!comm_.active()) return;
    Launch L = L_;
    L.stream = s;
    launch_gather(L, (int)plan_.n_send, send_idx_.ptr, d_ext, send_buf_.ptr);
    if (comm_.peer_on() && comm_.peer_halo_ready()) {
        comm_.peer_exchange_halo(send_buf_.ptr, d_ext + A.n, s);
        return;
    }
    comm_.exchange_f64(send_buf_.ptr, plan_.send_counts, plan_.send_offsets, d_ext + A.n, plan_.recv_counts,
                       plan_.recv_offsets, s);
}

// Row-blocks whose rows reference halo columns ("boundary") vs the rest ("interior"): the interior
// SpMV runs while the halo is on the wire.
void Context::classify_row_blocks()
{
    const int R = A.rows_per_block, nrb = (A.n + R - 1) / R;
    DeviceBuffer<int> flags;
    flags.ensure((size_t)nrb);
    PS_HIP_CHECK(hipMemsetAsync(flags.ptr, 0, (size_t)nrb * sizeof(int), stream));
    launch_classify_row_blocks(L_, A, flags.ptr);
    std::vector<int> h((size_t)nrb), in, bd;
    PS_HIP_CHECK(hipMemcpyAsync(h.data(), flags.ptr, (size_t)nrb * sizeof(int), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    for (int rb = 0; rb < nrb; ++rb) (h[rb] ? bd : in).push_back(rb);
    n_rb_interior_ = (int)in.size();
    n_rb_boundary_ = (int)bd.size();
    rb_interior_.ensure(in.size() + 1);
    rb_boundary_.ensure(bd.size() + 1);
    if (!in.empty())
        PS_HIP_CHECK(hipMemcpyAsync(rb_interior_.ptr, in.data(), in.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    if (!bd.empty())
        PS_HIP_CHECK(hipMemcpyAsync(rb_boundary_.ptr, bd.data(), bd.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    if (!comm_stream_) {
        PS_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
        PS_HIP_CHECK(hipEventCreateWithFlags(&ev_p_ready_, hipEventDisableTiming));
        PS_HIP_CHECK(hipEventCreateWithFlags(&ev_halo_done_, hipEventDisableTiming));
    }
}

const double *Context::extend(const double *d_v, double *d_ext)
{
    if (!comm_.active()) return d_v;
    if (d_v != d_ext)
        PS_HIP_CHECK(hipMemcpyAsync(d_ext, d_v, (size_t)A.n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    exchange_halo(d_ext);
    return d_ext;
}

// ---------------------------------------------------------------------------------------------
// solve
// ---------------------------------------------------------------------------------------------
void Context::solve_host(const double *b, double *x)
{
    const double t0 = wall_seconds();
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "[HIP] solve before factorize (size mismatch?)");
    PS_REQUIRE(b && x, PSOLVE_HIP_EINVAL, "solve: null vector");
    const size_t n = (size_t)A.n;
    b_dev_.ensure(n + 2);
    x_dev_.ensure(n + 2);
    // vectors of a few pages go through a pinned staging buffer (10 us less than a copy from the caller's pageable array);
    // from 1 MiB on the direct copy wins, 2-3 x at every size measured (16 MiB: 1.2 against 3.2 ms for b, x in and x out;
    // 128 MiB: 8.4 against 22 ms -- the two memcpy passes of the staged path are what costs; round 4, scripts/r4/host_solve_lab.py.
    // Until then the limit was 32 MiB, a round-1 measurement of pinning costs that this runtime no longer shows)
    const bool staged = n * sizeof(double) <= ((size_t)L_.lab.stage_kb << 10);
    if (staged) {
        stage_.ensure(2 * n);
        std::memcpy(stage_.ptr, b, n * sizeof(double));
        std::memcpy(stage_.ptr + n, x, n * sizeof(double));
        PS_HIP_CHECK(hipMemcpyAsync(b_dev_.ptr, stage_.ptr, n * sizeof(double), hipMemcpyHostToDevice, stream));
        PS_HIP_CHECK(hipMemcpyAsync(x_dev_.ptr, stage_.ptr + n, n * sizeof(double), hipMemcpyHostToDevice, stream));
    } else {
        PS_HIP_CHECK(hipMemcpyAsync(b_dev_.ptr, b, n * sizeof(double), hipMemcpyHostToDevice, stream));
        PS_HIP_CHECK(hipMemcpyAsync(x_dev_.ptr, x, n * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    stats.h2d_bytes += 2 * (int64_t)n * 8;
    stats.d2h_bytes += (int64_t)n * 8;
    solve_device(b_dev_.ptr, x_dev_.ptr);
    if (staged) {
        PS_HIP_CHECK(hipMemcpyAsync(stage_.ptr, x_dev_.ptr, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        std::memcpy(x, stage_.ptr, n * sizeof(double));
    } else {
        PS_HIP_CHECK(hipMemcpyAsync(x, x_dev_.ptr, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
    }
    info.time_solve = wall_seconds() - t0;
}

// K1 -> K2 -> K3 of one iteration of the fused (Jacobi / identity) loop on one GPU
void Context::enqueue_fused_iteration(int par, const double *invd, double *d_x)
{
    const int n = A.n, G = L_.grid, GS = L_.spmv_grid;
    double *part = partials_.ptr;
    double *part_pq = part + P_PQ * kMaxPartials, *part_rr = part + P_RR * kMaxPartials;
    double *part_rz = part + P_RZ * kMaxPartials;
    PcgState *S = state_.ptr;
    tl_spmv_kernel_record = 1; // (PCG's own product: what psolve_hip_last_spmv_kernel reports)
    launch_spmv(L_, A, SPMV_DOT, p_ext_.ptr, nullptr, q_.ptr, part_pq, &S->done[par]);
    tl_spmv_kernel_record = 0;
    last_spmv_kernel_ = tl_spmv_kernel_name;
    launch_pcg_update_r(L_, n, par, S, part_pq, GS, invd, q_.ptr, r_.ptr, part_rr, part_rz);
    launch_pcg_update_xp(L_, n, par, S, part_pq, GS, part_rr, part_rz, G, invd, r_.ptr, p_ext_.ptr, d_x, prm.max_iter);
}

// y = A v (v's halo exchanged first, on the comm stream while the interior row-blocks are multiplied when
// the overlap is on), part[0 .. return value) = partial sums of v . y
void Context::comm_mark(char kind, hipStream_t s)
{
    if (comm_ev_.size() <= comm_ev_used_) {
        hipEvent_t e;
        PS_HIP_CHECK(hipEventCreate(&e));
        comm_ev_.push_back(e);
        comm_kind_.push_back(kind);
    }
    comm_kind_[comm_ev_used_] = kind;
    PS_HIP_CHECK(hipEventRecord(comm_ev_[comm_ev_used_], s));
    ++comm_ev_used_;
}

int Context::dist_spmv_dot(double *v_ext, double *y, double *part, const int *done_flag)
{
    const int GS = L_.spmv_grid;
    const bool overlap = comm_.active() && prm.dist_overlap && n_rb_boundary_ > 0 && n_rb_interior_ > 0 &&
                         GS + 8 <= kMaxPartials;
    const bool mark = prof_now_ && comm_.active() && comm_.world() > 1;
    if (!overlap) {
        if (mark) comm_mark('h', stream);
        exchange_halo(v_ext);
        if (mark) comm_mark('H', stream);
        launch_spmv(L_, A, SPMV_DOT, v_ext, nullptr, y, part, done_flag);
        return GS;
    }
    PS_HIP_CHECK(hipEventRecord(ev_p_ready_, stream));
    PS_HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_p_ready_, 0));
    if (mark) comm_mark('h', comm_stream_);
    exchange_halo_on(v_ext, comm_stream_);
    if (mark) comm_mark('H', comm_stream_);
    PS_HIP_CHECK(hipEventRecord(ev_halo_done_, comm_stream_));
    SpmvExtra ex;
    ex.rb_list = rb_interior_.ptr;
    ex.n_list = n_rb_interior_;
    launch_spmv(L_, A, SPMV_DOT, v_ext, nullptr, y, part, done_flag, &ex);
    PS_HIP_CHECK(hipStreamWaitEvent(stream, ev_halo_done_, 0));
    Launch L2 = L_;
    L2.spmv_grid = std::min(std::min(GS, kMaxPartials - GS), std::max(8, (n_rb_boundary_ + 7) & ~7));
    ex.rb_list = rb_boundary_.ptr;
    ex.n_list = n_rb_boundary_;
    launch_spmv(L2, A, SPMV_DOT, v_ext, nullptr, y, part + GS, done_flag, &ex);
    return GS + L2.spmv_grid;
}

// Sharded Jacobi / identity PCG with ONE all-reduce per iteration (kernels.hip: cg1_update_kernel).  Same
// stopping rule as the fused loop (recurrence residual against ||b||), same state block, same polling.
// Iterations: update kernel -> halo of u + w = A u (+ u.w) -> fold of the three local sums -> all-reduce(3).
void Context::cg1_loop(const double *d_b, double *d_x, size_t &prof_used)
{
    const int n = A.n, G = L_.grid, GS = L_.spmv_grid;
    const double *invd = prm.precond == 1 ? invdiag_.ptr : nullptr;
    double *part = partials_.ptr;
    double *part_pq = part + P_PQ * kMaxPartials, *part_rr = part + P_RR * kMaxPartials;
    double *part_rz = part + P_RZ * kMaxPartials, *part_bb = part + P_BB * kMaxPartials;
    double *scal = scal_.ptr;
    PcgState *S = state_.ptr;
    double *u = p_ext_.ptr, *r = r_.ptr, *w = q_.ptr;
    cg1_p_.ensure((size_t)n + 2);
    cg1_s_.ensure((size_t)n + 2);
    double *p = cg1_p_.ptr, *s = cg1_s_.ptr;
    double *red3 = scal + S_INIT; // (gamma, rr, delta) of the current residual; [3] = ||b||^2 at start

    // r0 = b - A x0 ; u0 = M^-1 r0 ; w0 = A u0 ; one all-reduce of (r0.u0, r0.r0, w0.u0, b.b)
    const double *xin = extend(d_x, t_ext_.ptr);
    launch_spmv(L_, A, SPMV_RESIDUAL, xin, d_b, r, part_rr, nullptr);
    launch_dot(L_, n, d_b, d_b, part_bb);
    launch_sum_partials(L_, part_rr, GS, kMaxPartials, red3 + 1, 1);
    launch_sum_partials(L_, part_bb, G, kMaxPartials, red3 + 3, 1);
    launch_pcg_init_dir(L_, n, invd, r, u, part_rz);
    launch_sum_partials(L_, part_rz, G, kMaxPartials, red3 + 0, 1);
    const int npq0 = dist_spmv_dot(u, w, part_pq, nullptr);
    launch_sum_partials(L_, part_pq, npq0, kMaxPartials, red3 + 2, 1);
    comm_.allreduce_sum(red3, 4, stream);
    launch_pcg_init_state(L_, S, red3 + 1, red3 + 3, red3 + 0, 1, 1, 1, prm.rel_tol, prm.abs_tol);

    const int period = prm.check_period;
    int it = 0, chunk = 0;
    int it_at_copy[2] = {0, 0};
    bool finished = false;
    PcgState *hs = state_host_.ptr;
    while (!finished) {
        const int end = std::min(it + period, prm.max_iter);
        for (; it < end; ++it) {
            const int par = it & 1;
            launch_cg1_update(L_, n, par, it == 0 ? 1 : 0, S, red3, invd, u, w, p, s, d_x, r, part_rz, part_rr);
            const bool prof = prm.profile_spmv > 0 && (it % prm.profile_spmv) == 0;
            if (prof) {
                if (prof_ev_.size() < prof_used + 2) {
                    hipEvent_t a, b2;
                    PS_HIP_CHECK(hipEventCreate(&a));
                    PS_HIP_CHECK(hipEventCreate(&b2));
                    prof_ev_.push_back(a);
                    prof_ev_.push_back(b2);
                }
                PS_HIP_CHECK(hipEventRecord(prof_ev_[prof_used], stream));
            }
            prof_now_ = prof;
            if (it == 0) tl_spmv_kernel_record = 1; // (the single-reduction loop reports its product's instantiation too)
            const int npq = dist_spmv_dot(u, w, part_pq, &S->done[par ^ 1]);
            if (it == 0) {
                tl_spmv_kernel_record = 0;
                last_spmv_kernel_ = tl_spmv_kernel_name;
            }
            prof_now_ = false;
            if (prof) {
                PS_HIP_CHECK(hipEventRecord(prof_ev_[prof_used + 1], stream));
                prof_used += 2;
            }
            launch_cg1_fold(L_, part_rz, part_rr, G, part_pq, npq, red3);
            const bool mark = prof && comm_.world() > 1;
            if (mark) comm_mark('a', stream);
            comm_.allreduce_sum(red3, 3, stream);
            if (mark) comm_mark('A', stream);
        }
        const bool last = it >= prm.max_iter;
        if (last) // the residual of the last iterate has been reduced but not looked at yet
            launch_cg1_update(L_, n, it & 1, 2, S, red3, invd, u, w, p, s, d_x, r, part_rz, part_rr);
        const int slot = chunk & 1;
        PS_HIP_CHECK(hipMemcpyAsync(&hs[slot], S, sizeof(PcgState), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipEventRecord(poll_ev_[slot], stream));
        it_at_copy[slot] = it;
        int look = -1;
        if (last) look = slot;
        else if (chunk >= 1) look = slot ^ 1;
        if (look >= 0) {
            PS_HIP_CHECK(hipEventSynchronize(poll_ev_[look]));
            if (comm_.peer_on()) comm_.peer_poll(); // a waiting kernel that gave up ends the solve here, not at max_iter
            if (hs[look].done[it_at_copy[look] & 1]) finished = true;
        }
        if (last) finished = true;
        ++chunk;
    }
}

// ---------------------------------------------------------------------------------------------
// "reorder": Cuthill-McKee renumbering at factorize (reorder.hpp)
// ---------------------------------------------------------------------------------------------
bool Context::reorder_matrix(int64_t n, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_col, const double *d_values)
{
    // auto: identity / Jacobi (PCG's iterates do not depend on the numbering) and amg (the aggregation sweep follows the
    // numbering: the hierarchy is then AMGCL's hierarchy of the renumbered matrix -- 216^3 under a random numbering:
    // 125 -> 39 ms, same iteration count); the elimination order of ic and the domains of schwarz ARE the
    // preconditioner's definition (Eigen's NaturalOrdering; 64 consecutive unknowns): renumbered on request only
    // (round 6: likewise amg with an ORDERED relaxation -- gauss_seidel / ilu0 sweep in the numbering they are given, which is
    // amgcl's only in the caller's)
    if (prm.reorder == 2 && (prm.precond > 2 || n < prm.reorder_min_rows || (prm.precond == 2 && prm.amg.relax_type >= 3))) return false;
    const double t0 = wall_seconds();
    Launch L = Lmax_;
    L.stream = stream;
    const int b = (prm.block_size > 1 && n % prm.block_size == 0) ? prm.block_size : 1;
    // the order is kept while the pattern stays the same (Newton: Newton.cpp:189-193 factorizes a new Hessian of the
    // same pattern every iteration; MAS keeps its partition the same way, MASSolver.cu:304-321)
    ro_hash_dev_.ensure(2);
    scal_host_.ensure(S_COUNT);
    PS_HIP_CHECK(hipMemsetAsync(ro_hash_dev_.ptr, 0, 2 * sizeof(unsigned long long), stream));
    launch_hash_i32(L, n + 1, d_rowptr, ro_hash_dev_.ptr);
    launch_hash_i32(L, nnz, d_col, ro_hash_dev_.ptr + 1);
    unsigned long long h[2];
    PS_HIP_CHECK(hipMemcpyAsync(h, ro_hash_dev_.ptr, sizeof(h), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    ro_called_ = true;
    ro_hash_last_[0] = h[0];
    ro_hash_last_[1] = h[1];
    const bool same = ro_n_ == n && ro_nnz_ == nnz && ro_block_ == b && h[0] == ro_hash_[0] && h[1] == ro_hash_[1] &&
                      ro_mode_ == prm.reorder && ro_min_spread_ == prm.reorder_min_spread && ro_reverse_ == prm.reorder_reverse;
    ro_same_last_ = same;
    const int groups = (int)((n + 63) / 64), stride = std::max(1, groups / 4096);
    try {
    if (!same) {
        ro_n_ = -1;
        ro_map_valid_ = false;
        ro_decision_ = false;
        ro_info_ = ReorderInfo();
        ro_spread_after_ = 0.0;
        ro_spread_before_ = device_gather_spread(L, (int)n, d_rowptr, d_col, b, stride, bsr_scratch_);
        if (prm.reorder == 1 || ro_spread_before_ > prm.reorder_min_spread) {
            ++stats.reorder_searches;
            ro_order_.ensure((size_t)n + 1);
            ro_new_of_old_.ensure((size_t)n + 1);
            if (b == 1) {
                device_cuthill_mckee(L, (int)n, d_rowptr, d_col, ro_order_.ptr, ro_new_of_old_.ptr, ro_scratch_,
                                     bsr_scratch_, &ro_info_);
                if (prm.reorder_reverse) launch_reverse_order(L, (int)n, ro_order_.ptr, ro_new_of_old_.ptr);
            } else { // block value types: whole nodes move (the b x b blocks stay blocks)
                CsrDev T;
                T.n = (int)n;
                T.n_ext = (int)n;
                T.nnz = nnz;
                T.rowptr = d_rowptr;
                T.col = d_col;
                T.val = d_values;
                BlockGraph G;
                device_block_graph(L, T, b, G, bsr_scratch_);
                const int nb = (int)(n / b);
                ro_node_order_.ensure((size_t)nb + 1);
                ro_node_new_.ensure((size_t)nb + 1);
                device_cuthill_mckee(L, nb, G.ptr.ptr, G.col.ptr, ro_node_order_.ptr, ro_node_new_.ptr, ro_scratch_,
                                     bsr_scratch_, &ro_info_);
                if (prm.reorder_reverse) launch_reverse_order(L, nb, ro_node_order_.ptr, ro_node_new_.ptr);
                launch_expand_node_order(L, nb, b, ro_node_order_.ptr, ro_order_.ptr, ro_new_of_old_.ptr);
            }
            ro_decision_ = true;
            ro_scratch_.claim.release(); // (8 n bytes of search state: not needed until the pattern changes)
            ro_scratch_.cnt.release();
            ro_scratch_.tsum.release();
            ro_node_order_.release();
            ro_node_new_.release();
        }
        ro_n_ = n;
        ro_nnz_ = nnz;
        ro_block_ = b;
        ro_hash_[0] = h[0];
        ro_hash_[1] = h[1];
        ro_mode_ = prm.reorder;
        ro_min_spread_ = prm.reorder_min_spread;
        ro_reverse_ = prm.reorder_reverse;
    }
    if (ro_decision_) {
        // round 6: a factorize of the SAME pattern (Newton: every iteration) moves the values only -- a gather through the
        // map the first permutation left behind (20 bytes per entry) instead of sorting every row's columns again
        // (permute_rows_lds_kernel: 4.0 ms of a 44 ms refresh of configs[2] under a random numbering)
        if (same && ro_map_valid_ && ro_map_.count >= (size_t)nnz && ro_val_.count >= (size_t)nnz && nnz < (int64_t)INT32_MAX) {
            launch_gather(L, (int)nnz, ro_map_.ptr, d_values, ro_val_.ptr);
        } else {
            device_permute_csr(L, (int)n, nnz, d_rowptr, d_col, d_values, ro_new_of_old_.ptr, ro_new_of_old_.ptr, ro_ptr_,
                               ro_col_, &ro_val_, bsr_scratch_, &ro_map_);
            ro_map_valid_ = ro_map_.ptr != nullptr && ro_map_.count >= (size_t)nnz;
        }
        if (!same) {
            ro_spread_after_ = device_gather_spread(L, (int)n, ro_ptr_.ptr, ro_col_.ptr, b, stride, bsr_scratch_);
            // auto: a numbering the search does not improve by a tenth stays as the caller made it
            if (prm.reorder == 2 && ro_spread_after_ > 0.9 * ro_spread_before_) ro_decision_ = false;
        }
    }
    if (ro_decision_) {
        ro_b_.ensure((size_t)n + 2);
        ro_x_.ensure((size_t)n + 2);
    }
    } catch (const Error &e) {
        // auto: a device without room for the second copy of the matrix keeps the caller's numbering
        if (prm.reorder != 2 || e.code != PSOLVE_HIP_EDEVICE) throw;
        (void)hipGetLastError();
        ro_decision_ = false;
        ro_n_ = -1;
    }
    if (!ro_decision_) {
        ro_ptr_.release();
        ro_col_.release();
        ro_val_.release();
        ro_map_.release();
        ro_map_valid_ = false;
        ro_b_.release();
        ro_x_.release();
    }
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    ro_seconds_ = wall_seconds() - t0;
    return ro_decision_;
}

bool Context::order_host_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, std::vector<int32_t> &order,
                                 std::vector<int32_t> &new_of_old, ReorderInfo &rinfo, double &spread_before,
                                 double &spread_after)
{
    use_device();
    check_sizes(n, nnz);
    Launch L = Lmax_;
    L.stream = stream;
    const int b = (prm.block_size > 1 && n % prm.block_size == 0) ? prm.block_size : 1;
    // the pattern only: 4 (n + nnz) bytes on this device, whatever the size of the values
    DeviceBuffer<int> d_ptr, d_col, p_ptr, p_col;
    d_ptr.ensure((size_t)n + 1);
    d_col.ensure((size_t)nnz + 4);
    PS_HIP_CHECK(hipMemcpyAsync(d_ptr.ptr, outer, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipMemcpyAsync(d_col.ptr, inner, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    stats.h2d_bytes += (int64_t)(n + 1) * 4 + nnz * 4;
    const int groups = (int)((n + 63) / 64), stride = std::max(1, groups / 4096);
    rinfo = ReorderInfo();
    spread_after = 0.0;
    spread_before = device_gather_spread(L, (int)n, d_ptr.ptr, d_col.ptr, b, stride, bsr_scratch_);
    if (prm.reorder != 1 && spread_before <= prm.reorder_min_spread) return false;
    ro_order_.ensure((size_t)n + 1);
    ro_new_of_old_.ensure((size_t)n + 1);
    if (b == 1) {
        device_cuthill_mckee(L, (int)n, d_ptr.ptr, d_col.ptr, ro_order_.ptr, ro_new_of_old_.ptr, ro_scratch_, bsr_scratch_, &rinfo);
        if (prm.reorder_reverse) launch_reverse_order(L, (int)n, ro_order_.ptr, ro_new_of_old_.ptr);
    } else {
        CsrDev T;
        T.n = (int)n;
        T.n_ext = (int)n;
        T.nnz = nnz;
        T.rowptr = d_ptr.ptr;
        T.col = d_col.ptr;
        BlockGraph G;
        device_block_graph(L, T, b, G, bsr_scratch_);
        const int nb = (int)(n / b);
        ro_node_order_.ensure((size_t)nb + 1);
        ro_node_new_.ensure((size_t)nb + 1);
        device_cuthill_mckee(L, nb, G.ptr.ptr, G.col.ptr, ro_node_order_.ptr, ro_node_new_.ptr, ro_scratch_, bsr_scratch_, &rinfo);
        if (prm.reorder_reverse) launch_reverse_order(L, nb, ro_node_order_.ptr, ro_node_new_.ptr);
        launch_expand_node_order(L, nb, b, ro_node_order_.ptr, ro_order_.ptr, ro_new_of_old_.ptr);
    }
    device_permute_csr(L, (int)n, nnz, d_ptr.ptr, d_col.ptr, nullptr, ro_new_of_old_.ptr, ro_new_of_old_.ptr, p_ptr, p_col,
                       nullptr, bsr_scratch_);
    spread_after = device_gather_spread(L, (int)n, p_ptr.ptr, p_col.ptr, b, stride, bsr_scratch_);
    const bool take = prm.reorder == 1 || spread_after <= 0.9 * spread_before;
    if (take) {
        order.resize((size_t)n);
        new_of_old.resize((size_t)n);
        PS_HIP_CHECK(hipMemcpyAsync(order.data(), ro_order_.ptr, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipMemcpyAsync(new_of_old.data(), ro_new_of_old_.ptr, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        stats.d2h_bytes += 8 * n;
    }
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    ro_order_.release();
    ro_new_of_old_.release();
    ro_node_order_.release();
    ro_node_new_.release();
    ro_scratch_.claim.release();
    ro_scratch_.cnt.release();
    ro_scratch_.tsum.release();
    ro_version_ = 0;
    return take;
}

void Context::factorize_host_rows_packed(int64_t n_global, int64_t row_begin, int64_t row_end, const int32_t *ptr,
                                         const int32_t *col, const double *val, const int32_t *new_of_old,
                                         uint64_t order_version)
{
    const double t0 = wall_seconds();
    use_device();
    PS_REQUIRE(ptr && col && val && new_of_old, PSOLVE_HIP_EINVAL, "factorize: null matrix arrays");
    set_partition(n_global, row_begin, row_end);
    const int64_t n = row_end - row_begin, nnz = ptr[n];
    check_sizes(n, nnz);
    Launch L = Lmax_;
    L.stream = stream;
    if (ro_version_ != order_version || ro_new_of_old_.count < (size_t)n_global) {
        ro_new_of_old_.ensure((size_t)n_global + 1);
        PS_HIP_CHECK(hipMemcpyAsync(ro_new_of_old_.ptr, new_of_old, (size_t)n_global * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        stats.h2d_bytes += 4 * n_global;
        ro_version_ = order_version;
    }
    // staging: the packed rows as the caller numbered their columns
    ro_ptr_.ensure((size_t)n + 1);
    ro_col_.ensure((size_t)nnz + 4);
    ro_val_.ensure((size_t)nnz + 4);
    PS_HIP_CHECK(hipMemcpyAsync(ro_ptr_.ptr, ptr, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipMemcpyAsync(ro_col_.ptr, col, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    PS_HIP_CHECK(hipMemcpyAsync(ro_val_.ptr, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, stream));
    stats.h2d_bytes += (int64_t)(n + 1) * 4 + nnz * 12;
    ++stats.matrix_uploads;
    factorized_ = false;
    device_permute_csr(L, (int)n, nnz, ro_ptr_.ptr, ro_col_.ptr, ro_val_.ptr, nullptr, ro_new_of_old_.ptr, rowptr_own_, col_own_,
                       &val_own_, bsr_scratch_);
    PS_HIP_CHECK(hipStreamSynchronize(stream)); // the caller's packed arrays may go now
    factorize_device(n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
    info.time_factorize = wall_seconds() - t0;
}

const double *Context::to_new(const double *d_v, double *buf)
{
    launch_gather(L_, A.n, ro_order_.ptr, d_v, buf);
    return buf;
}

void Context::to_old(const double *buf, double *d_v) { launch_gather(L_, A.n, ro_new_of_old_.ptr, buf, d_v); }

bool Context::reorder_perm(int *new_of_old)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "reorder_perm before factorize");
    if (!reordered_) return false;
    PS_HIP_CHECK(hipMemcpyAsync(new_of_old, ro_new_of_old_.ptr, (size_t)A.n * sizeof(int), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    return true;
}

void Context::solve_device(const double *d_b, double *d_x)
{
    if (!reordered_) {
        solve_device_inner(d_b, d_x);
        return;
    }
    const double t0 = wall_seconds();
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "[HIP] solve before factorize");
    PS_REQUIRE(d_b && d_x, PSOLVE_HIP_EINVAL, "solve: null vector");
    to_new(d_b, ro_b_.ptr);
    to_new(d_x, ro_x_.ptr); // the initial guess (Solver.hpp:119-127)
    solve_device_inner(ro_b_.ptr, ro_x_.ptr);
    to_old(ro_x_.ptr, d_x);
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    info.time_solve_device = wall_seconds() - t0;
    info.time_solve = info.time_solve_device;
}

// Jacobi selected after a factorize under another preconditioner: its diagonal now (single device; shards always have it)
void Context::ensure_jacobi_diagonal()
{
    if (prm.precond != 1 || invdiag_valid_) return;
    ensure_workspace();
    PS_HIP_CHECK(hipMemsetAsync(flags_.ptr, 0, 4 * sizeof(int), stream));
    launch_diag_inverse(L_, A, invdiag_.ptr, flags_.ptr);
    int bad = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&bad, flags_.ptr, sizeof(int), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    PS_REQUIRE(bad == 0, PSOLVE_HIP_ENUMERIC, "solve: " + std::to_string(bad) + " non-finite diagonal entries");
    invdiag_valid_ = true;
}

void Context::solve_device_inner(const double *d_b, double *d_x)
{
    const double t0 = wall_seconds();
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "[HIP] solve before factorize");
    PS_REQUIRE(d_b && d_x, PSOLVE_HIP_EINVAL, "solve: null vector");
    ensure_jacobi_diagonal();
    PS_REQUIRE(((uintptr_t)d_b % 16) == 0 && ((uintptr_t)d_x % 16) == 0, PSOLVE_HIP_EINVAL,
               "solve_device: vectors must be 16-byte aligned");
    PS_REQUIRE(prm.precond != 2 || amg_ || damg_, PSOLVE_HIP_EINVAL, "precond=amg was selected after factorize; factorize again");
    PS_REQUIRE(prm.precond != 3 || (schwarz_ && schwarz_->rows() == A.n), PSOLVE_HIP_EINVAL,
               "precond=schwarz was selected after factorize; factorize again");
    PS_REQUIRE(prm.precond != 4 || (ic_ && ic_->rows() == A.n), PSOLVE_HIP_EINVAL,
               "precond=ic was selected after factorize; factorize again");
    if (prm.fault_solve_rank >= 0) { // one shard leaves the collective sequence of a solve before its first collective
        const bool me = comm_.active() && comm_.rank() == prm.fault_solve_rank;
        prm.fault_solve_rank = -1;
        if (me) throw Error(PSOLVE_HIP_EDEVICE, "injected fault (fault.solve_rank)");
    }
    ++stats.solves;
    ensure_workspace();
    const int n = A.n, G = L_.grid, GS = L_.spmv_grid; // partial counts: vector kernels / SpMV
    const bool dist = comm_.active();
    const bool fused = prm.precond < 2;
    const double *invd = prm.precond == 1 ? invdiag_.ptr : nullptr;
    double *part = partials_.ptr;
    double *part_pq = part + P_PQ * kMaxPartials, *part_rr = part + P_RR * kMaxPartials;
    double *part_rz = part + P_RZ * kMaxPartials, *part_bb = part + P_BB * kMaxPartials;
    double *scal = scal_.ptr;
    PcgState *S = state_.ptr;
    double *p = p_ext_.ptr, *r = r_.ptr, *q = q_.ptr;

    size_t prof_used = 0;
    prof2_used_ = 0;
    k2_ms_avg_ = k3_ms_avg_ = 0.0;
    // one all-reduce per iteration instead of two costs 16 n more bytes per iteration (the single-reduction step
    // updates five vectors): worth it on small shards, where the all-reduce latency is the iteration (256^3 over 8
    // GPUs), not on large ones (256^3 PER GPU, one-rank communicator: 0.597 ms per iteration against 0.445 ms).
    // Every rank decides from the global size, so they all take the same loop.
    const bool single_reduction = dist && fused && prm.dist_single_reduction &&
                                  n_global_ / std::max(1, comm_.world()) <= (int64_t)prm.dist_single_reduction_max_rows;
    if (single_reduction) {
        cg1_loop(d_b, d_x, prof_used);
    } else {
    // ---- r0 = b - A x0 ; ||b||^2 ; p0 = M^-1 r0 ; rz0 -------------------------------------------
    const double *xin = extend(d_x, t_ext_.ptr);
    launch_spmv(L_, A, SPMV_RESIDUAL, xin, d_b, r, part_rr, nullptr);
    launch_dot(L_, n, d_b, d_b, part_bb);
    if (fused) {
        launch_pcg_init_dir(L_, n, invd, r, p, part_rz);
    } else {
        apply_generic_precond(r, z_.ptr, nullptr);
        launch_dot(L_, n, r, z_.ptr, part_rz);
        PS_HIP_CHECK(hipMemcpyAsync(p, z_.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    if (dist) {
        // rr came from the SpMV grid; P_RZ, P_BB are adjacent arrays of the vector grid -> scal[0..2]
        launch_sum_partials(L_, part_rr, GS, kMaxPartials, scal + S_INIT, 1);
        launch_sum_partials(L_, part_rz, G, kMaxPartials, scal + S_INIT + 1, 2);
        comm_.allreduce_sum(scal + S_INIT, 3, stream);
        launch_pcg_init_state(L_, S, scal + S_INIT, scal + S_INIT + 2, scal + S_INIT + 1, 1, 1, 1, prm.rel_tol, prm.abs_tol);
    } else {
        launch_pcg_init_state(L_, S, part_rr, part_bb, part_rz, GS, G, G, prm.rel_tol, prm.abs_tol);
    }

    // ---- the loop ---------------------------------------------------------------------------------
    // generic (AMG) path: poll every iteration, but one iteration behind, so that the GPU always has the
    // next iteration queued; everything queued behind the converged iteration returns at once (latch)
    const int period = fused ? prm.check_period : 1;
    const bool run_ahead = true;
    int it = 0, chunk = 0;
    int it_at_copy[2] = {0, 0};
    bool finished = false;
    PcgState *hs = state_host_.ptr;
    // Launch-bound regime (small systems: three ~10 us launches per iteration): replay one hipGraph
    // per polling chunk instead of 3 x period eager launches.  The parity pattern repeats every two
    // iterations, so one captured chunk (even period, starting at an even iteration) serves the whole solve.
    // (measured on the AMG path too: ~35 launches per iteration replayed as a graph are no faster than eager
    // launches polled one iteration behind, so only the fused loop is captured)
    // (round 6: the captured loop pays where the iteration is launch-bound -- 16^3 ... 64^3: 3-8 % -- is neutral from 96^3 to
    // 200^3 and costs 2.6 % at 256^3 on the row-kinds storage (145.7 against 141.9 ms, profiles/r06_graph_by_size.jsonl): kept
    // for systems of up to two million rows)
    const bool graphable = fused && !dist && prm.use_graph && prm.profile_spmv == 0 && (period % 2) == 0 && A.n <= 2000000;
    if (graphable) {
        GraphKey key;
        key.x = d_x; key.val = A.val; key.invd = invd; key.n = n; key.grid = G; key.spmv_grid = GS;
        key.period = period; key.R = A.rows_per_block; key.xcd = L_.spmv_xcd_map; key.chunk = L_.spmv_chunk_rows;
        if (!loop_graph_ || !(key == loop_graph_key_)) {
            if (loop_graph_) {
                (void)hipGraphExecDestroy(loop_graph_);
                loop_graph_ = nullptr;
            }
            hipGraph_t g = nullptr;
            PS_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            try {
                for (int k = 0; k < period; ++k) enqueue_fused_iteration(k & 1, invd, d_x);
            } catch (...) {
                (void)hipStreamEndCapture(stream, &g);
                if (g) (void)hipGraphDestroy(g);
                throw;
            }
            PS_HIP_CHECK(hipStreamEndCapture(stream, &g));
            PS_HIP_CHECK(hipGraphInstantiate(&loop_graph_, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            loop_graph_key_ = key;
        }
    }
    while (!finished) {
        const int end = std::min(it + period, prm.max_iter);
        if (graphable && end - it == period && (it % 2) == 0) {
            PS_HIP_CHECK(hipGraphLaunch(loop_graph_, stream));
            it = end;
        }
        for (; it < end; ++it) {
            const int par = it & 1;
            const bool overlap = dist && prm.dist_overlap && n_rb_boundary_ > 0 && n_rb_interior_ > 0 &&
                                 GS + 8 <= kMaxPartials;
            if (dist && !overlap) exchange_halo(p);
            const bool prof = prm.profile_spmv > 0 && (it % prm.profile_spmv) == 0;
            if (prof) {
                if (prof_ev_.size() < prof_used + 2) {
                    hipEvent_t a, b2;
                    PS_HIP_CHECK(hipEventCreate(&a));
                    PS_HIP_CHECK(hipEventCreate(&b2));
                    prof_ev_.push_back(a);
                    prof_ev_.push_back(b2);
                }
            }
            // (round 6) a sampled product that is ONE launch carries its two events itself (Launch::ev_start / ev_stop: the
            // kernel's own begin and end, as rocprofv3's trace has them); the overlapped product of a shard -- two launches around
            // a halo wait -- is still bracketed by recorded events
            const bool kprof = prof && !overlap;
            if (prof && !kprof) PS_HIP_CHECK(hipEventRecord(prof_ev_[prof_used], stream));
            int n_pq = GS; // partials the SpMV leaves in part_pq
            if (overlap) {
                // halo of p travels on the comm stream while the interior row-blocks are multiplied
                PS_HIP_CHECK(hipEventRecord(ev_p_ready_, stream));
                PS_HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_p_ready_, 0));
                const bool hmark = prof && comm_.world() > 1;
                if (hmark) comm_mark('h', comm_stream_);
                exchange_halo_on(p, comm_stream_);
                if (hmark) comm_mark('H', comm_stream_);
                PS_HIP_CHECK(hipEventRecord(ev_halo_done_, comm_stream_));
                SpmvExtra ex;
                ex.rb_list = rb_interior_.ptr;
                ex.n_list = n_rb_interior_;
                tl_spmv_kernel_record = 1;
                launch_spmv(L_, A, SPMV_DOT, p, nullptr, q, part_pq, &S->done[par], &ex);
                tl_spmv_kernel_record = 0;
                last_spmv_kernel_ = tl_spmv_kernel_name;
                PS_HIP_CHECK(hipStreamWaitEvent(stream, ev_halo_done_, 0));
                Launch L2 = L_;
                L2.spmv_grid = std::min(std::min(GS, kMaxPartials - GS), std::max(8, (n_rb_boundary_ + 7) & ~7));
                ex.rb_list = rb_boundary_.ptr;
                ex.n_list = n_rb_boundary_;
                launch_spmv(L2, A, SPMV_DOT, p, nullptr, q, part_pq + GS, &S->done[par], &ex);
                n_pq = GS + L2.spmv_grid;
            } else {
                tl_spmv_kernel_record = 1;
                if (kprof) {
                    L_.ev_start = prof_ev_[prof_used];
                    L_.ev_stop = prof_ev_[prof_used + 1];
                }
                launch_spmv(L_, A, SPMV_DOT, p, nullptr, q, part_pq, &S->done[par]);
                L_.ev_start = L_.ev_stop = nullptr;
                tl_spmv_kernel_record = 0;
                last_spmv_kernel_ = tl_spmv_kernel_name;
            }
            if (prof) {
                if (!kprof) PS_HIP_CHECK(hipEventRecord(prof_ev_[prof_used + 1], stream));
                prof_used += 2;
            }
            const double *c_pq = part_pq;
            int np_pq = n_pq, np = G;
            if (dist) {
                launch_sum_partials(L_, part_pq, n_pq, kMaxPartials, scal + S_PQ, 1);
                const bool mark = prof && comm_.world() > 1;
                if (mark) comm_mark('a', stream);
                comm_.allreduce_sum(scal + S_PQ, 1, stream);
                if (mark) comm_mark('A', stream);
                c_pq = scal + S_PQ;
                np_pq = 1;
                np = 1;
            }
            if (fused) {
                if (it == 0) tl_spmv_kernel_record = 1; // (the names of the two vector kernels, once per solve)
                // (sampled iterations of an undistributed solve: the two vector kernels timed too, each by its own pair of
                // kernel-timestamp events -- bench.py names the kernel that takes most of the iteration, whichever it is)
                const bool prof23 = prof && !dist;
                if (prof23) {
                    while (prof_ev2_.size() < prof2_used_ + 4) {
                        hipEvent_t e;
                        PS_HIP_CHECK(hipEventCreate(&e));
                        prof_ev2_.push_back(e);
                    }
                    L_.ev_start = prof_ev2_[prof2_used_];
                    L_.ev_stop = prof_ev2_[prof2_used_ + 1];
                }
                launch_pcg_update_r(L_, n, par, S, c_pq, np_pq, invd, q, r, part_rr, part_rz);
                L_.ev_start = L_.ev_stop = nullptr;
                const double *c_rr = part_rr, *c_rz = part_rz;
                if (dist) {
                    launch_sum_partials(L_, part_rr, G, kMaxPartials, scal + S_RR, 2); // rr, rz adjacent
                    comm_.allreduce_sum(scal + S_RR, 2, stream);
                    c_rr = scal + S_RR;
                    c_rz = scal + S_RZ;
                }
                if (prof23) {
                    L_.ev_start = prof_ev2_[prof2_used_ + 2];
                    L_.ev_stop = prof_ev2_[prof2_used_ + 3];
                }
                launch_pcg_update_xp(L_, n, par, S, c_pq, np_pq, c_rr, c_rz, np, invd, r, p, d_x, prm.max_iter);
                L_.ev_start = L_.ev_stop = nullptr;
                if (it == 0) {
                    tl_spmv_kernel_record = 0;
                    last_vec_kernel_[0] = tl_vec_kernel_name[0];
                    last_vec_kernel_[1] = tl_vec_kernel_name[1];
                }
                if (prof23) prof2_used_ += 4;
            } else {
                launch_pcg_update_xr(L_, n, par, S, c_pq, np_pq, p, q, d_x, r, part_rr);
                const double *c_rr = part_rr, *c_rz = part_rz;
                if (dist) {
                    launch_sum_partials(L_, part_rr, G, kMaxPartials, scal + S_RR, 1);
                    comm_.allreduce_sum(scal + S_RR, 1, stream);
                    c_rr = scal + S_RR;
                }
                launch_pcg_check(L_, par, S, c_rr, np, prm.max_iter);
                apply_generic_precond(r, z_.ptr, &S->done[par ^ 1]);
                launch_dot(L_, n, r, z_.ptr, part_rz);
                if (dist) {
                    launch_sum_partials(L_, part_rz, G, kMaxPartials, scal + S_RZ, 1);
                    comm_.allreduce_sum(scal + S_RZ, 1, stream);
                    c_rz = scal + S_RZ;
                }
                launch_pcg_update_p(L_, n, par, S, c_rz, np, z_.ptr, p);
            }
        }
        // poll: async copy of the state after this chunk; decide on the previous chunk's copy so the
        // GPU always has one chunk queued
        const int slot = chunk & 1;
        PS_HIP_CHECK(hipMemcpyAsync(&hs[slot], S, sizeof(PcgState), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipEventRecord(poll_ev_[slot], stream));
        it_at_copy[slot] = it;
        const bool last = it >= prm.max_iter;
        int look = -1;
        if (!run_ahead || last) look = slot;
        else if (chunk >= 1) look = slot ^ 1;
        if (look >= 0) {
            PS_HIP_CHECK(hipEventSynchronize(poll_ev_[look]));
            if (comm_.peer_on()) comm_.peer_poll(); // a waiting kernel that gave up ends the solve here, not at max_iter
            if (hs[look].done[it_at_copy[look] & 1]) finished = true;
        }
        if (last) finished = true;
        ++chunk;
    }
    } // standard (two-reduction) loop
    PcgState *hs = state_host_.ptr;
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    PS_HIP_CHECK(hipMemcpyAsync(&hs[2], S, sizeof(PcgState), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    const PcgState st = hs[2];

    if (st.zero_rhs) // Eigen: rhsNorm2 == 0 -> x = 0
        PS_HIP_CHECK(hipMemsetAsync(d_x, 0, (size_t)n * sizeof(double), stream));

    // peer-mapped collectives: a waiting kernel that gave up (a peer never arrived) left garbage behind -- say so
    if (dist && comm_.peer_on()) comm_.peer_check(stream);

    const bool converged = st.status != PSOLVE_HIP_RUNNING;
    info.num_iterations = st.passes;
    info.solver_iter = converged ? (st.passes > 0 ? st.passes - 1 : 0) : st.passes;
    info.rhs_norm = std::sqrt(st.rhs_norm2);
    info.solver_error = st.zero_rhs ? 0.0 : std::sqrt(st.rn2 / st.rhs_norm2);
    info.final_res_norm = info.solver_error;
    info.solver_status = converged ? st.status : PSOLVE_HIP_REACH_MAX_ITERATIONS;

    // sampled SpMV timings
    info.spmv_ms_avg = 0.0;
    info.spmv_samples = 0;
    if (prof_used) {
        double tot = 0.0;
        int64_t cnt = 0;
        const int live = std::min<int>((int)(prof_used / 2), prm.profile_spmv > 0 ? (st.passes + prm.profile_spmv - 1) / prm.profile_spmv : 0);
        for (int k = 0; k < live; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, prof_ev_[2 * k], prof_ev_[2 * k + 1]) == hipSuccess) {
                tot += ms;
                ++cnt;
            }
        }
        info.spmv_samples = cnt;
        info.spmv_ms_avg = cnt ? tot / cnt : 0.0;
        // ... and of the two vector kernels behind them (quadruples of prof_ev2_ line up with the pairs of prof_ev_)
        double t2 = 0.0, t3 = 0.0;
        int64_t c23 = 0;
        const int live2 = std::min<int>(live, (int)(prof2_used_ / 4));
        for (int k = 0; k < live2; ++k) {
            float a = 0.f, b2 = 0.f;
            if (hipEventElapsedTime(&a, prof_ev2_[4 * k], prof_ev2_[4 * k + 1]) == hipSuccess &&
                hipEventElapsedTime(&b2, prof_ev2_[4 * k + 2], prof_ev2_[4 * k + 3]) == hipSuccess) {
                t2 += a;
                t3 += b2;
                ++c23;
            }
        }
        k2_ms_avg_ = c23 ? t2 / c23 : 0.0;
        k3_ms_avg_ = c23 ? t3 / c23 : 0.0;
    }

    ar_us_avg_ = halo_us_avg_ = 0.0;
    ar_samples_ = halo_samples_ = 0;
    if (comm_ev_used_) {
        double tot[2] = {0.0, 0.0};
        int cnt[2] = {0, 0};
        for (size_t k = 0; k + 1 < comm_ev_used_; ++k) {
            const char a = comm_kind_[k], b2 = comm_kind_[k + 1];
            if (!((a == 'a' && b2 == 'A') || (a == 'h' && b2 == 'H'))) continue;
            float ms = 0.f;
            if (hipEventSynchronize(comm_ev_[k + 1]) == hipSuccess && hipEventElapsedTime(&ms, comm_ev_[k], comm_ev_[k + 1]) == hipSuccess) {
                tot[a == 'h'] += ms;
                ++cnt[a == 'h'];
            }
        }
        ar_samples_ = cnt[0];
        halo_samples_ = cnt[1];
        ar_us_avg_ = cnt[0] ? 1e3 * tot[0] / cnt[0] : 0.0;
        halo_us_avg_ = cnt[1] ? 1e3 * tot[1] / cnt[1] : 0.0;
        comm_ev_used_ = 0;
    }

    // true residual (the reference tests check ||Ax - b|| themselves; we report it)
    info.true_residual = -1.0;
    if (prm.true_residual && !st.zero_rhs) {
        const double *xf = extend(d_x, t_ext_.ptr);
        launch_spmv(L_, A, SPMV_RESIDUAL, xf, d_b, r, part_rr, nullptr);
        launch_sum_partials(L_, part_rr, GS, kMaxPartials, scal + S_TMP, 1);
        if (dist) comm_.allreduce_sum(scal + S_TMP, 1, stream);
        PS_HIP_CHECK(hipMemcpyAsync(scal_host_.ptr, scal + S_TMP, sizeof(double), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        info.true_residual = std::sqrt(scal_host_.ptr[0] / st.rhs_norm2);
    }
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    info.time_solve_device = wall_seconds() - t0;
    info.time_solve = info.time_solve_device;
}

void Context::amg_level_info(int level, int64_t *rows, int64_t *nnz, double *rho) const
{
    if (damg_) { // the hierarchy built on the shards: global rows, this shard's stored entries
        damg_->level_shape(level, rows, nullptr, nnz, rho);
        return;
    }
    PS_REQUIRE(amg_ && level >= 0 && level < amg_->levels(), PSOLVE_HIP_EINVAL, "amg_level_info: no such level");
    amg_->level_shape(level, rows, nnz, rho);
}

void Context::amg_level_matrix_shape(int level, int what, int64_t out[3]) const
{
    PS_REQUIRE(amg_ != nullptr, PSOLVE_HIP_EINVAL, "amg_level_matrix: no AMG hierarchy");
    amg_->level_matrix_shape(level, what, out);
}

void Context::matrix_copy(int32_t *rowptr, int32_t *col, double *val)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "matrix_copy: not factorized");
    if (rowptr) PS_HIP_CHECK(hipMemcpyAsync(rowptr, A.rowptr, ((size_t)A.n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (col && A.nnz) PS_HIP_CHECK(hipMemcpyAsync(col, A.col, (size_t)A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    if (val && A.nnz) PS_HIP_CHECK(hipMemcpyAsync(val, A.val, (size_t)A.nnz * sizeof(double), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
}

void Context::amg_level_matrix_copy(int level, int what, int *rowptr, int *col, double *val)
{
    use_device();
    PS_REQUIRE(amg_ != nullptr, PSOLVE_HIP_EINVAL, "amg_level_matrix: no AMG hierarchy");
    amg_->level_matrix_copy(stream, level, what, rowptr, col, val);
}

void Context::amg_time_level_ops(int level, int reps, double out_us[5])
{
    use_device();
    PS_REQUIRE(amg_ != nullptr && factorized_, PSOLVE_HIP_EINVAL, "amg_time_level_ops: no AMG hierarchy");
    amg_->time_level_ops(*this, level, reps, out_us);
}

bool Context::amg_level_perm(int level, int *perm)
{
    use_device();
    PS_REQUIRE(amg_ != nullptr, PSOLVE_HIP_EINVAL, "amg_level_perm: no AMG hierarchy");
    return amg_->level_perm_copy(stream, level, perm);
}

// ---------------------------------------------------------------------------------------------
// single kernels (parity tests, roofline bench)
// ---------------------------------------------------------------------------------------------
void Context::spmv(const double *d_x, double *d_y)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "spmv before factorize");
    if (reordered_) { // the caller's numbering outside: y = Pi^T (Pi A Pi^T) Pi x
        launch_spmv(L_, A, SPMV_PLAIN, to_new(d_x, ro_x_.ptr), nullptr, ro_b_.ptr, nullptr, nullptr);
        to_old(ro_b_.ptr, d_y);
        return;
    }
    const double *xin = extend(d_x, t_ext_.ptr);
    launch_spmv(L_, A, SPMV_PLAIN, xin, nullptr, d_y, nullptr, nullptr);
}

double Context::spmv_dot(const double *d_x, double *d_y)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "spmv before factorize");
    const double *xin = reordered_ ? to_new(d_x, ro_x_.ptr) : extend(d_x, t_ext_.ptr);
    double *part = partials_.ptr + P_TMP * kMaxPartials;
    launch_spmv(L_, A, SPMV_DOT, xin, nullptr, reordered_ ? ro_b_.ptr : d_y, part, nullptr);
    if (reordered_) to_old(ro_b_.ptr, d_y);
    launch_sum_partials(L_, part, L_.spmv_grid, kMaxPartials, scal_.ptr + S_TMP, 1);
    if (comm_.active()) comm_.allreduce_sum(scal_.ptr + S_TMP, 1, stream);
    PS_HIP_CHECK(hipMemcpyAsync(scal_host_.ptr, scal_.ptr + S_TMP, sizeof(double), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    return scal_host_.ptr[0];
}

double Context::dot(int64_t n, const double *a, const double *b)
{
    use_device();
    PS_REQUIRE(n >= 0 && n < INT32_MAX, PSOLVE_HIP_ERANGE, "dot: n out of range");
    PS_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0, PSOLVE_HIP_EINVAL, "dot: 16-byte alignment");
    partials_.ensure((size_t)P_COUNT * kMaxPartials);
    scal_.ensure(S_COUNT);
    scal_host_.ensure(S_COUNT);
    double *part = partials_.ptr + P_TMP * kMaxPartials;
    launch_dot(L_, (int)n, a, b, part);
    launch_sum_partials(L_, part, L_.grid, kMaxPartials, scal_.ptr + S_TMP, 1);
    if (comm_.active()) comm_.allreduce_sum(scal_.ptr + S_TMP, 1, stream);
    PS_HIP_CHECK(hipMemcpyAsync(scal_host_.ptr, scal_.ptr + S_TMP, sizeof(double), hipMemcpyDeviceToHost, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
    return scal_host_.ptr[0];
}

void Context::axpby(int64_t n, double a, const double *x, double b, double *y)
{
    use_device();
    PS_REQUIRE(n >= 0 && n < INT32_MAX, PSOLVE_HIP_ERANGE, "axpby: n out of range");
    launch_axpby(L_, (int)n, a, x, b, y);
}

void Context::apply_generic_precond(const double *d_r, double *d_z, const int *done_flag)
{
    if (prm.precond == 3) schwarz_->apply(*this, d_r, d_z, done_flag);
    else if (prm.precond == 4) ic_->apply(*this, d_r, d_z, done_flag);
    else if (damg_) damg_->apply(*this, d_r, d_z, done_flag);
    else amg_->apply(*this, d_r, d_z, done_flag);
}

void Context::precond_apply(const double *d_r, double *d_z)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "precond_apply before factorize");
    double *d_z_user = nullptr;
    if (reordered_) {
        d_r = to_new(d_r, ro_x_.ptr);
        d_z_user = d_z;
        d_z = ro_b_.ptr;
    }
    if (prm.precond >= 2) {
        PS_REQUIRE(prm.precond != 2 || amg_ != nullptr || damg_ != nullptr, PSOLVE_HIP_EINVAL,
                   "precond=amg was selected after factorize; factorize again");
        PS_REQUIRE(prm.precond != 3 || (schwarz_ && schwarz_->rows() == A.n), PSOLVE_HIP_EINVAL,
                   "precond=schwarz was selected after factorize; factorize again");
        PS_REQUIRE(prm.precond != 4 || (ic_ && ic_->rows() == A.n), PSOLVE_HIP_EINVAL,
                   "precond=ic was selected after factorize; factorize again");
        apply_generic_precond(d_r, d_z, nullptr);
    } else {
        ensure_jacobi_diagonal();
        launch_vmul(L_, A.n, prm.precond == 1 ? invdiag_.ptr : nullptr, d_r, d_z);
    }
    if (d_z_user) to_old(d_z, d_z_user);
}

double Context::time_spmv(const double *d_x, double *d_y, int reps)
{
    use_device();
    PS_REQUIRE(factorized_ && reps > 0, PSOLVE_HIP_EINVAL, "time_spmv: not factorized / reps <= 0");
    double *d_y_user = nullptr;
    if (reordered_) { // timed: the product in the numbering it runs in
        d_y_user = d_y;
        d_y = ro_b_.ptr;
    }
    const double *xin = reordered_ ? to_new(d_x, ro_x_.ptr) : extend(d_x, t_ext_.ptr);
    hipEvent_t a, b;
    PS_HIP_CHECK(hipEventCreate(&a));
    PS_HIP_CHECK(hipEventCreate(&b));
    double *part = partials_.ptr + P_TMP * kMaxPartials;
    launch_spmv(L_, A, SPMV_DOT, xin, nullptr, d_y, part, nullptr); // warm-up
    PS_HIP_CHECK(hipEventRecord(a, stream));
    for (int i = 0; i < reps; ++i) {
        SpmvExtra ex;
        ex.reverse = (L_.lab.alternate & 1) ? (i & 1) : 0;
        launch_spmv(L_, A, SPMV_DOT, xin, nullptr, d_y, part, nullptr, &ex);
    }
    PS_HIP_CHECK(hipEventRecord(b, stream));
    PS_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    PS_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (d_y_user) to_old(ro_b_.ptr, d_y_user);
    return (double)ms / reps;
}

void Context::time_vecops(int reps, double *ms_update, double *ms_direction)
{
    use_device();
    PS_REQUIRE(factorized_ && reps > 0, PSOLVE_HIP_EINVAL, "time_vecops: not factorized / reps <= 0");
    ensure_workspace();
    ensure_jacobi_diagonal();
    const int n = A.n, G = L_.grid;
    double *part = partials_.ptr;
    // a state that never converges and keeps the vectors bounded (r = 0 => beta = 0, p = 0)
    PcgState hs;
    std::memset(&hs, 0, sizeof(hs));
    hs.threshold = -1.0;
    hs.rz[0] = 1.0;
    PS_HIP_CHECK(hipMemcpyAsync(state_.ptr, &hs, sizeof(hs), hipMemcpyHostToDevice, stream));
    launch_fill(L_, G, 1.0, part + P_PQ * kMaxPartials);
    launch_fill(L_, n, 0.0, r_.ptr);
    launch_fill(L_, n, 0.0, q_.ptr);
    launch_fill(L_, n, 0.0, p_ext_.ptr);
    launch_fill(L_, n, 0.0, t_ext_.ptr);
    const double *invd = prm.precond == 1 ? invdiag_.ptr : nullptr;
    hipEvent_t e[3];
    for (auto &x : e) PS_HIP_CHECK(hipEventCreate(&x));
    PS_HIP_CHECK(hipEventRecord(e[0], stream));
    for (int i = 0; i < reps; ++i)
        launch_pcg_update_r(L_, n, 0, state_.ptr, part + P_PQ * kMaxPartials, G, invd, q_.ptr, r_.ptr,
                            part + P_RR * kMaxPartials, part + P_RZ * kMaxPartials);
    PS_HIP_CHECK(hipEventRecord(e[1], stream));
    for (int i = 0; i < reps; ++i) {
        // parity 0 always: rz[0] stays 0, done[0] stays 0 (the kernel only writes index 1)
        launch_pcg_update_xp(L_, n, 0, state_.ptr, part + P_PQ * kMaxPartials, G, part + P_RR * kMaxPartials,
                             part + P_RZ * kMaxPartials, G, invd, r_.ptr, p_ext_.ptr, t_ext_.ptr, 1 << 30);
    }
    PS_HIP_CHECK(hipEventRecord(e[2], stream));
    PS_HIP_CHECK(hipEventSynchronize(e[2]));
    float a = 0.f, b = 0.f;
    PS_HIP_CHECK(hipEventElapsedTime(&a, e[0], e[1]));
    PS_HIP_CHECK(hipEventElapsedTime(&b, e[1], e[2]));
    for (auto &x : e) (void)hipEventDestroy(x);
    *ms_update = (double)a / reps;
    *ms_direction = (double)b / reps;
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs
// ---------------------------------------------------------------------------------------------
void Context::generate_poisson7(int nx, int ny, int nz, int z0, int z1)
{
    use_device();
    PS_REQUIRE(nx > 0 && ny > 0 && nz > 0 && z0 >= 0 && z0 < z1 && z1 <= nz, PSOLVE_HIP_EINVAL,
               "generate_poisson7: bad grid / plane range");
    const int64_t plane = (int64_t)nx * ny;
    const int64_t n_global = plane * nz, row0 = plane * z0, row1 = plane * z1, n_local = row1 - row0;
    PS_REQUIRE(n_global < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "global size exceeds int32 column ids");
    const int64_t nnz_local = poisson7_nnz_before(nx, ny, nz, row1) - poisson7_nnz_before(nx, ny, nz, row0);
    check_sizes(n_local, nnz_local);
    const bool dist = comm_.active();
    PS_REQUIRE(dist || (z0 == 0 && z1 == nz), PSOLVE_HIP_EINVAL,
               "generate_poisson7: a partial plane range needs comm_init");
    factorized_ = false;
    rowptr_own_.ensure((size_t)n_local + 1);
    col_own_.ensure((size_t)nnz_local + 4);
    val_own_.ensure((size_t)nnz_local + 4);
    launch_poisson7_generate(L_, nx, ny, nz, z0, z1, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr);
    n_global_ = n_global;
    row_begin_ = row0;
    row_end_ = row1;
    gen_nx_ = nx; gen_ny_ = ny; gen_nz_ = nz; gen_z0_ = z0; gen_z1_ = z1;
    factorize_device(n_local, nnz_local, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
}

void Context::generate_rhs(uint64_t seed, double *d_b, double *d_xstar)
{
    use_device();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "generate_rhs before factorize");
    if (reordered_) { // x* by the caller's row index, b = A x* through the renumbered product
        double *xu = q_.ptr;
        launch_splitmix(L_, A.n, seed, row_begin_, xu);
        spmv(xu, d_b);
        if (d_xstar)
            PS_HIP_CHECK(hipMemcpyAsync(d_xstar, xu, (size_t)A.n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        return;
    }
    double *xs = t_ext_.ptr;
    launch_splitmix(L_, A.n, seed, row_begin_, xs);
    const int n_halo = A.n_ext - A.n;
    if (n_halo > 0) launch_splitmix_indexed(L_, n_halo, seed, halo_dev_.ptr, xs + A.n);
    launch_spmv(L_, A, SPMV_PLAIN, xs, nullptr, d_b, nullptr, nullptr);
    if (d_xstar)
        PS_HIP_CHECK(hipMemcpyAsync(d_xstar, xs, (size_t)A.n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    PS_HIP_CHECK(hipStreamSynchronize(stream));
}

} // namespace psolve
