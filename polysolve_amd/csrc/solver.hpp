// solver.hpp -- the backend object behind one psolve_hip_t handle.
//
// Role in the reference: the `MASSolverImpl`-style pimpl behind a `Solver` subclass
// (/root/reference/src/polysolve/linear/MASSolver.cu:133-219, MASSolver.hpp:69-71), with the solve
// semantics of EigenIterative<ConjugateGradient> (EigenSolver.tpp:101-114): x is the initial guess,
// the stopping rule is on the recurrence residual relative to ||b||.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "amg_symbolic.hpp"
#include "common.hpp"
#include "dist.hpp"
#include "host_hash.hpp"
#include "kernels.hpp"
#include "pattern.hpp"
#include "reorder.hpp"
#include "sell.hpp"

namespace psolve {

class AmgHierarchy; // amg.hip
class DistAmg;      // amg_dist.hip
struct HaloLink;    // amg_dist.hpp
class SchwarzPrecond; // schwarz.hip
class IcPrecond;      // ic.hip

struct AmgParams {
    // names/defaults: AMGCL.cpp:32-65; ncycle = 1 (V-cycle, BASELINE.json north_star) instead of 2
    int max_levels = 6;
    int coarse_enough = 3000;
    int ncycle = 1;
    int npre = 1, npost = 1;
    double eps_strong = 0.0;
    double sa_relax = 1.0;
    int estimate_spectral_radius = 1;
    int sa_power_iters = 0;
    int cheb_degree = 16;
    int cheb_power_iters = 100;
    double cheb_higher = 2.0;
    double cheb_lower = 0.008333333333;
    int block_size = 1; // copied from Params::block_size at factorize
    int reuse = 1;      // same pattern at the next factorize: keep aggregates/patterns, redo the numbers on the device
    int device_setup = 1; // patterns and numbers built on the device (0: all-host hierarchy, uploaded)
    int matrix_fp32 = 0;  // the cycle's operators stream single-precision values (arithmetic stays double)
    int stream_nt = -1;   // products inside the cycle: -1 follow the solver's spmv_nt / spmv_kernel policy, 0 never non-temporal
    int col16 = 0;        // copied from Params::spmv_col16 at factorize: the cycle's CSR operators stream 16-bit columns where encodable
    int block_levels = 1; // block_size 3: every operator of the cycle (A_l of levels >= 1, P_l, R_l) multiplies through a 3x3-block copy and the block-scaled Chebyshev step is an epilogue of the block product; 0: round 3's cycle (scalar CSR below level 0, residual product + a node-local update launch per step)
    int sell = 0;         // operators of levels >= 1 multiply through a SELL-64-sigma copy: 0 never (measured neutral inside the cycle), 1 wide rows (>= 12 entries per row), 2 always
    int renumber = 0;     // scalar systems, device setup: levels >= 1 of at least renumber_min_rows rows are renumbered for locality after the setup (same hierarchy, the nodes of a coarse aggregate consecutive; amg_renumber.hip).  Off by default: on the 256^3 hierarchy the level-1 products gain 12-15 us each and the level-0 prolongation, whose gathers follow the coarse numbering, loses 43 (profiles/r03_amg.md)
    int renumber_min_rows = 65536;
    int dist_global = 2;  // shards, scalar systems: 2 = one hierarchy built ON the shards (aggregates confined to a shard, Galerkin products with exchanged halo rows, levels under dist_replicate_rows x ranks rows gathered and replicated: amg_dist.hpp); 1 = the single-device hierarchy, built by every rank from the gathered matrix (level 0 applied on the shard, coarser levels replicated: exact single-device iteration counts, memory and setup do not scale); 0 = one hierarchy per shard (additive Schwarz)
    int dist_replicate_rows = 50000; // dist_global 2: a level of fewer than this many rows PER RANK is gathered and the rest of the hierarchy replicated
    int dist_global_max_mbytes = 4096; // dist_global 1: matrices above this size (12 nnz + 4 n bytes, global) take dist_global 2 instead of being gathered
    int device_aggregation = 1;       // the aggregation sweep on the device (same aggregates as the sequential loop)
    int aggregation_rounds = 0;       // 0: one kernel in which every vertex waits for the earlier ones it depends on; 1: dependency rounds (two kernels per round)
    int aggregation_max_rounds = 10000; // beyond this depth (or pace; 10 us per round for the waiting kernel) the host sweep takes over
    int aggregation_min_rows = 100000;  // smaller levels are swept faster by the host
    // round 5
    int overlap_smoothers = 1; // the smoothers' power iterations run on a second stream beside the aggregation sweep / the Galerkin products
    int aggregation = 0;    // 0 "amgcl": the sequential greedy sweep of plain_aggregates, reproduced exactly; 1 "parallel": a distance-2 maximal independent set by hashed priorities (oracle: orc_parallel_aggregates), same membership rule; 2 "compact" (round 6): one-hop aggregates around two generations of such sets, the rest by most connections (oracle: orc_compact_aggregates) -- the parallel mode for block / 27-point node graphs
    int coarsening = 0;     // 0 smoothed_aggregation, 1 aggregation (P = P_tent, Galerkin operator scaled by 1 / over_interp) -- amgcl::runtime::coarsening
    double over_interp = 0; // coarsening "aggregation": amgcl's over_interp (0: its default, 1.5 for scalar and 2.0 for block value types)
    int relax_type = 0;     // 0 chebyshev, 1 damped_jacobi, 2 spai0, 3 gauss_seidel, 4 ilu0 (round 6: ordered sweeps, amg_sweep.hip) -- amgcl::runtime::relaxation
    double ilu_damping = 1.0; // ilu0: x += damping (LU)^-1 (rhs - A x), amgcl::relaxation::ilu0::params::damping
    int precond_class = 0;  // 0 amg, 1 relaxation (/AMGCL/precond/class: amgcl::relaxation::as_preconditioner -- the smoother of the system matrix alone, relax.apply)
    double damping = 0.72;  // damped_jacobi: amgcl's default
    int cheb_scale = 1;     // chebyshev.scale (AMGCL.cpp:57: true)
    int refresh_power_iters = -1; // -1: a refresh estimates the smoothers' radii like a first factorize (cheb_power_iters steps from amgcl's random vector); k >= 0 (opt-in, not amgcl's estimate): it continues from the vector the previous factorize ended with for k steps (0: keeps the previous radii)
    int coarse_dense = 1024; // a RELAXED coarsest level of at most this many rows is applied as one dense operator built at factorize (the smoother's recurrence run on the identity): one launch per visit instead of (npre + npost) x degree; 0: off
    int direct_coarse = 0;  // 1: the coarsest level is solved by a dense Cholesky factorization (amgcl: skyline_lu) instead of being relaxed
};

struct Params {
    int max_iter = 10000;          // /MAS/max_iter (linear-solver-spec.json:481-484)
    double rel_tol = 1e-8;         // on ||r|| / ||b||  (BASELINE.json metric)
    double abs_tol = 0.0;          // on ||r||
    int precond = 1;               // 0 identity, 1 jacobi, 2 amg, 3 multilevel additive Schwarz on 64-unknown domains, 4 incomplete Cholesky (Eigen::IncompleteCholesky in the natural ordering)
    double ic_initial_shift = 1e-3; // precond 4: Eigen's setInitialShift
    int ic_ordering = 1;            // precond 4: 1 = approximate minimum degree (Eigen::AMDOrdering<int>: the default of IncompleteCholesky<double>, so of the reference), 0 = natural
    int schwarz_levels = 1;        // precond 3: levels of 64-fold coarsening (1 = block Jacobi with dense 64 x 64 inverses;
                                   // more levels pay only when 64 consecutive unknowns form a compact cluster)
    int block_size = 1;
    int check_period = 16;
    int true_residual = 1;
    int profile_spmv = 0;
    int blocks_per_cu = 8;         // persistent grid of the vector kernels (the bound the AMG levels' and the setup kernels' grids are fitted under)
    int vec_blocks_per_cu = 2;     // ... of PCG's OWN fused vector kernels (pcg_update_r / _xp, the single-reduction update): round 6 --
                                   // a three- to six-stream update runs faster on 2 workgroups per CU than on 8, at every size (Jacobi-PCG
                                   // 128^3 27.0 -> 23.9 ms, 256^3 287 -> 282 ms, 384^3 1135 -> 1059 ms, 512^3 3022 -> 2812 ms: K2 0.91 ->
                                   // 0.70 ms = 0.59 -> 0.77 of peak; profiles/r06_large_single.md)
    int spmv_blocks_per_cu = 6;    // persistent grid of the SpMV (its 24.6 KB LDS tile admits 6 workgroups per CU)
    int spmv_kernel = -1;          // 1: LDS-DMA staged kernel (round 2), 0: register-staged pipeline (round 1), -1: by operator size
    int spmv_nt = -1;              // non-temporal stream + stores: -1 auto by operator size, 0 off, 1 on
    int vec_policy = 7;            // non-temporal streams of the fused vector kernels: bit 0 loads, 1 r, 2 x, 3 p stores
    int spmv_nt_mbytes = 384;      // auto: operators above this many MiB are streamed non-temporally (1.5 x the Infinity Cache)
    int spmv_xcd_map = 2;          // 0 round-robin, 1 contiguous eighths, 2 chunks of rows dealt to the XCDs
    int spmv_chunk_rows = 8192;    // xcd_map 2: rows per chunk
    int spmv_rows_per_block = 0;   // 0 = auto from nnz / n
    int dist_overlap = 1;          // shards: SpMV of the interior rows overlaps the halo exchange
    int dist_single_reduction = 1; // shards: Chronopoulos-Gear recurrences, one all-reduce per iteration instead of two
    int dist_single_reduction_max_rows = 3000000; // ... on shards of at most this many rows (global rows / ranks); larger ones keep Eigen's recurrence with two all-reduces
    int spmv_value_dict = 1;       // rows that repeat pattern AND values bit for bit (constant-coefficient stencils; PatDev::kind):
                                   // the products stream 16-bit row kinds instead of the matrix -- the same sums; rebuilt from
                                   // the values of every factorize, absent where rows do not repeat (the usual FEM matrix)
    int spmv_col16 = 0;            // operators without a dictionary / block / SELL copy whose row-blocks touch at most eight
                                   // 8192-column windows (any local numbering: grids, breadth-first orders, coarse AMG levels)
                                   // stream 16-bit columns: 10 instead of 12 bytes per entry, same columns, same sums.  Off by
                                   // default: it pays only where the product is HBM-bound (256^3 renumbered: 312 -> 287 us per
                                   // product, 237 -> 229 ms per solve) and nothing on cache-resident operators and the cycle's
                                   // coarse levels, which wait for their gathers, not for the stream (profiles/r03_col16.json)
    int use_bsr3 = 1;              // block_size 3: run the fine-level products on a 3x3-block copy
    int use_graph = 1;             // replay a hipGraph per polling chunk of the fused loop (single GPU)
    int reorder = 2;               // single device: renumber the system at factorize for the locality of the gathers (Cuthill-McKee
                                   // by breadth-first levels, reorder.hpp; the order is kept while the pattern stays the same).
                                   // 0 off (the caller's numbering, row sums bit-equal to the oracle's), 1 always, 2 auto: with
                                   // identity / Jacobi (PCG's iterates do not depend on the numbering) and amg (the hierarchy
                                   // of the renumbered matrix), not with ic / schwarz (whose definition is the numbering),
                                   // on systems of at least reorder_min_rows rows whose
                                   // numbering spreads the gathers of 64 consecutive rows over more than reorder_min_spread
                                   // times the fewest cache lines they could occupy, and only if the search improves that
    int reorder_min_rows = 131072;
    double reorder_min_spread = 2.5;
    int dist_collectives = 0;      // in-process multi-device handle: 0 = RCCL (all-reduce, grouped send / recv), 1 = peer-mapped: the per-iteration all-reduce of the CG scalars and the halo exchange of PCG's vector by stores into the peers' memory + epoch flags (dist_peer.hip); setup-time exchanges and the hierarchy's level halos stay on RCCL
    int reorder_reverse = 1;       // the breadth-first order read backwards (reverse Cuthill-McKee): same bandwidth, the aggregation sweep of amg then runs against the search direction
    int fault_solve_rank = -1;     // fault injection (tests of the multi-device abort path): the shard of this rank fails at the start of its next solve, once
    AmgParams amg;
};

// new index of every original row under generate_poisson7_permuted's renumbering (host; generators.hip)
void permutation_host(int64_t n, int mode, int64_t window, uint64_t seed, int32_t *out);

// value of a plain parameter of `prm` (every key set_param accepts); false for an unknown key
bool param_value(const Params &prm, const std::string &key, double *out);

class Context {
public:
    std::shared_ptr<AllocMeter> meter_ = std::make_shared<AllocMeter>(); // ("stats.device_bytes"; shared with the buffers it counts)
    AllocMeter &meter = *meter_;
    explicit Context(int device_id);
    ~Context();

    void set_stream(void *s);
    void synchronize();
    void set_param(const std::string &key, double v);
    double get_param(const std::string &key) const;

    void analyze_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int precond_num);
    void factorize_host(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, const double *values);
    // rows [row_begin, row_end) of a host matrix of n_global rows (global column ids): upload the slice and
    // factorize it as this handle's shard (the in-process multi-device handle, multi.cpp)
    void factorize_host_rows(int64_t n_global, int64_t row_begin, int64_t row_end, const int32_t *outer,
                             const int32_t *inner, const double *values);
    void factorize_device(int64_t n_local, int64_t nnz_local, const int32_t *d_rowptr, const int32_t *d_col,
                          const double *d_values, bool owned);
    // The in-process multi-device handle renumbers BEFORE it partitions ("reorder" on several devices, multi.cpp):
    // the order of a host pattern, searched on this handle's device.  Returns true and fills order / new_of_old where
    // the system is to be renumbered ("reorder" 1, or 2 and the numbering is scattered and the search improves it).
    bool order_host_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, std::vector<int32_t> &order,
                            std::vector<int32_t> &new_of_old, ReorderInfo &info, double &spread_before, double &spread_after);
    // rows [row_begin, row_end) of the RENUMBERED matrix, packed by the caller in the new row order with the caller's
    // (old, global) column ids: upload, rename the columns through new_of_old (uploaded when `order_version` changes),
    // sort every row, factorize as this handle's shard
    void factorize_host_rows_packed(int64_t n_global, int64_t row_begin, int64_t row_end, const int32_t *ptr,
                                    const int32_t *col, const double *val, const int32_t *new_of_old, uint64_t order_version);
    void solve_host(const double *b, double *x);
    void solve_device(const double *d_b, double *d_x);

    void generate_poisson7(int nx, int ny, int nz, int z0, int z1);
    void generate_elasticity_q1(int M, double E, double nu); // generators.hip
    // ... with the NODES renumbered pseudo-randomly (mode / window / seed as generate_poisson7_permuted; 0: grid numbering)
    void generate_elasticity_q1_permuted(int M, double E, double nu, int mode, int64_t window, uint64_t seed);
    // 7-point Poisson under a symmetric pseudo-random renumbering (mode 1: all rows, 2: inside windows): no pattern
    // dictionary, real gathers -- the unstructured leg of the bench (generators.hip)
    void generate_poisson7_permuted(int nx, int ny, int nz, int mode, int64_t window, uint64_t seed);
    void generate_rhs(uint64_t seed, double *d_b, double *d_xstar);
    static void check_sizes_public(int64_t n, int64_t nnz);

    void spmv(const double *d_x, double *d_y);
    double spmv_dot(const double *d_x, double *d_y);
    double dot(int64_t n, const double *a, const double *b);
    void axpby(int64_t n, double a, const double *x, double b, double *y);
    void precond_apply(const double *d_r, double *d_z);
    double time_spmv(const double *d_x, double *d_y, int reps);
    void time_vecops(int reps, double *ms_update, double *ms_direction);
    void box_probe(double *out, int n_out); // probe.hip: dependent-load latencies and gather rates of this box

    void comm_init(int rank, int world, const char *id, const char *rccl_path);
    void comm_init_local(LocalGroup *g, int rank);
    void set_partition(int64_t n_global, int64_t row_begin, int64_t row_end);

    void use_device() const;
    Comm &comm() { return comm_; }
    void export_halo_link(HaloLink &out); // a copy of the shard's halo plan (partition, halo ids, send lists)
    void halo_exchange(double *d_ext) { exchange_halo(d_ext); }          // shards: d_ext[n ..) <- the owners' entries
    void allreduce(double *d_buf, int count) { comm_.allreduce_sum(d_buf, count, stream); }
    Launch launch_config() const { return L_; }
    Launch launch_max() const { return Lmax_; }
    // the b x b block copy of the factorized matrix, when factorize built one (block_size 3 + use_bsr3)
    // identity of the pattern of the factorized matrix A (the arrays the products run on: renumbered where "reorder"
    // renumbers), computed once per factorize and shared by everything that keeps symbolic work across factorizes of the
    // same pattern (the hierarchy, the block copy): a hash of rowptr / col -- or, where reorder_matrix has just recognised
    // the caller's pattern and kept its order, the previous value without touching the arrays
    unsigned long long pattern_id_of_A() const { return a_hash_; }
    // the instantiation PCG's own product (q = A p with the fused p.q) ran on in the last solve, as rocprofv3 names it
    const std::string &last_spmv_kernel() const { return last_spmv_kernel_; }
    const std::string &last_vec_kernel(int which) const { return last_vec_kernel_[which & 1]; }
    bool pattern_of_A_unchanged() const { return a_same_; }
    // the block-row kinds of the shared block graph (level 0 of a block hierarchy), or nullptr
    const Bsr3KindDev *shared_block_kinds() const { return (A.bsr3 && A.bsr3->kinds) ? A.bsr3->kinds : nullptr; }
    BlockGraph *shared_block_graph(int b) { return (A.bsr3 && b == 3 && bsr_graph_.b == 3) ? &bsr_graph_ : nullptr; }
    void matrix_copy(int32_t *rowptr, int32_t *col, double *val); // D2H of the factorized matrix (any pointer may be null)
    void amg_level_info(int level, int64_t *rows, int64_t *nnz, double *rho) const;
    void amg_level_matrix_shape(int level, int what, int64_t out[3]) const;
    void amg_level_matrix_copy(int level, int what, int *rowptr, int *col, double *val);
    bool amg_level_perm(int level, int *perm);
    void amg_time_level_ops(int level, int reps, double out_us[5]);
    // "reorder": is the factorized system renumbered, and new_of_old[i] = row that row i of the caller's numbering became
    bool reordered() const { return reordered_; }
    bool reorder_perm(int *new_of_old);

    psolve_hip_info info{};
    std::string last_error;
    // what crossed PCIe / was rebuilt, since the handle was created ("stats.*" keys of get_param): lets a
    // caller verify that prefactorize + many solves (FEMSolver.cpp:269-342) move only b and x
    struct Stats {
        int64_t h2d_bytes = 0, d2h_bytes = 0; // bulk transfers of the host entry points (matrix, b, x)
        int64_t matrix_uploads = 0;           // factorize(host arrays) calls that uploaded a matrix
        int64_t pattern_uploads = 0;          // ... of them with the pattern (the others recognised the one on the device)
        int64_t amg_setups = 0, amg_refreshes = 0; // hierarchies built from scratch / refreshed numerically
        int64_t reorder_searches = 0;              // Cuthill-McKee searches (a factorize of the pattern it holds keeps the order)
        int64_t solves = 0;
    } stats;
    int device = 0;
    hipStream_t stream = nullptr;
    Params prm;
    CsrDev A; // the factorized operator (with "reorder": in the new numbering)
    int64_t n_halo() const { return (int64_t)plan_.halo.size(); }

private:
    void ensure_workspace();
    void solve_device_inner(const double *d_b, double *d_x);
    // "reorder": the renumbered copy of the matrix the caller handed over; returns false where the system keeps the
    // caller's numbering (off, shards, a numbering that is already local)
    bool reorder_matrix(int64_t n, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_col, const double *d_values);
    const double *to_new(const double *d_v, double *buf);  // buf[k] = v[order[k]]  (the caller's vector in the new numbering)
    void to_old(const double *buf, double *d_v);           // v[i] = buf[new_of_old[i]]
    // shards: a failure only one rank sees must become every rank's failure BEFORE the next collective, or the others
    // block in it for ever (a hung GPU on a real multi-GPU node).  Throws Error(code, msg) where !ok, and an
    // ECOMM "another shard failed" on the ranks that were fine.
    void shards_agree(bool ok, int code, const std::string &msg);
    void refit_launch(); // L_ from Lmax_ and the factorized matrix (grids, non-temporal policy)
    void setup_halo(const int32_t *d_col, bool owned);
    const double *extend(const double *d_v, double *d_ext); // halo exchange into d_ext if distributed
    void exchange_halo(double *d_ext);

    hipStream_t own_stream_ = nullptr;
    Launch L_;    // grids fitted to the factorized matrix
    Launch Lmax_; // grids from the parameters (upper bounds)
    int num_cus_ = 256;
    bool spmv_grid_user_set_ = false; // "spmv_blocks_per_cu" was set by the caller: no per-kernel grid override

    // matrix storage (owned when it came from host arrays or the generator)
    DeviceBuffer<int> rowptr_own_, col_own_;
    DeviceBuffer<double> val_own_;
    bool factorized_ = false;
    // the pattern of the last factorize(host arrays), while rowptr_own_ / col_own_ still hold it (host_hash.hpp)
    HostPatternHash host_pat_hash_;
    int64_t host_pat_n_ = -1, host_pat_nnz_ = -1;
    bool host_pat_resident_ = false, from_host_ = false;
    // "reorder"
    bool reordered_ = false;
    DeviceBuffer<int> ro_order_, ro_new_of_old_, ro_node_order_, ro_node_new_;
    DeviceBuffer<int> ro_ptr_, ro_col_;
    DeviceBuffer<double> ro_val_, ro_b_, ro_x_;
    DeviceBuffer<int> ro_map_;    // source position of every entry of the renumbered copy: a factorize of the same pattern gathers its values
    bool ro_map_valid_ = false;
    ReorderScratch ro_scratch_;
    ReorderInfo ro_info_;
    unsigned long long ro_hash_[2] = {0, 0}; // of the pattern the kept order belongs to
    bool ro_called_ = false, ro_same_last_ = false; // this factorize: reorder_matrix ran / recognised the caller's pattern
    unsigned long long ro_hash_last_[2] = {0, 0};   // ... and the hash of the caller's arrays it computed
    std::string last_spmv_kernel_, last_vec_kernel_[2];
    unsigned long long a_hash_ = 0;                 // pattern_id_of_A
    int64_t a_hash_n_ = -1, a_hash_nnz_ = -1;
    bool a_hash_reordered_ = false, a_same_ = false;
    bool pat_tried_ = false;                        // pat_ is the dictionary (or the absence of one) of the pattern pat_id_
    unsigned long long pat_id_ = 0, bsr_graph_id_ = 0; // pattern ids (pattern_id_of_A) the dictionary / the block graph were built for
    int pat_n_ = -1;
    int bsr_graph_n_ = -1;                          // the block graph in bsr_graph_ belongs to a_hash_ (rows of A then)
    int64_t ro_n_ = -1, ro_nnz_ = -1;
    uint64_t ro_version_ = 0; // of the new_of_old a shard holds (factorize_host_rows_packed)
    int ro_block_ = 1, ro_mode_ = 0;
    double ro_min_spread_ = 0.0;
    int ro_reverse_ = 0;
    bool ro_decision_ = false; // of the kept pattern: renumbered (true) or left as the caller numbered it
    double ro_spread_before_ = 0.0, ro_spread_after_ = 0.0, ro_seconds_ = 0.0;
    DeviceBuffer<unsigned long long> ro_hash_dev_;
    int64_t analyzed_n_ = -1, analyzed_nnz_ = -1;
    int precond_num_ = 0;

    // block_size 3: zero-filled 3x3 block copy of the matrix for the BSR SpMV
    DeviceBuffer<int> loc_ptr_, loc_col_; // shards + AMG: the diagonal block the local hierarchy is built on
    DeviceBuffer<double> loc_val_;
    BlockGraph bsr_graph_;       // the 3x3-block copy (pattern by the row-set kernels, values by a kernel)
    SymbolicScratch bsr_scratch_;
    SellMatrix sell_; // SELL-64-sigma copy of a wide-row operator (see factorize_device)
    DeviceBuffer<double> kdinv_; // 1 / diag per row kind (Launch::kd_tab), valid when kdinv_valid_
    bool kdinv_valid_ = false;
    bool invdiag_valid_ = false; // Jacobi's inverse diagonal holds THIS factorize's values (computed at factorize under precond jacobi, else at the first use)
    void ensure_jacobi_diagonal();
    Bsr3Kinds bsr_kinds_; // block-row kinds of bsr_ (Bsr3Dev::kinds)
    PatMatrix pat_;   // pattern dictionary of a narrow-row operator (see factorize_device)
    Col16 col16_;     // 16-bit column copy of an operator without one (see factorize_device)
    Bsr3Dev bsr_;
    void build_bsr3();

    // Poisson generator metadata (for generate_rhs)
    int gen_nx_ = 0, gen_ny_ = 0, gen_nz_ = 0, gen_z0_ = 0, gen_z1_ = 0;

    // vectors
    DeviceBuffer<double> invdiag_, r_, q_, z_, p_ext_, t_ext_, b_dev_, x_dev_;
    DeviceBuffer<double> partials_; // 6 arrays of kMaxPartials
    DeviceBuffer<double> scal_;     // all-reduce staging / host-visible dots
    DeviceBuffer<PcgState> state_;
    DeviceBuffer<int> flags_;       // misc device ints (bad diag count, cursors)
    PinnedBuffer<PcgState> state_host_;
    PinnedBuffer<double> scal_host_;
    PinnedBuffer<double> stage_; // pinned staging of host vectors of a few pages (solve_host)
    hipEvent_t poll_ev_[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> prof_ev_;
    std::vector<hipEvent_t> prof_ev2_; // sampled iterations: begin / end of pcg_update_r, begin / end of pcg_update_xp (kernel timestamps)
    size_t prof2_used_ = 0;
    double k2_ms_avg_ = 0.0, k3_ms_avg_ = 0.0;
    // shards, sampled iterations ("profile_spmv"): events around one all-reduce of the CG scalars (main stream) and around
    // the halo exchange (the stream it runs on) -- "stats.allreduce_us_avg" / "stats.halo_us_avg" of the last solve
    std::vector<hipEvent_t> comm_ev_;
    std::vector<char> comm_kind_;
    size_t comm_ev_used_ = 0;
    bool prof_now_ = false;
    double ar_us_avg_ = 0.0, halo_us_avg_ = 0.0;
    int ar_samples_ = 0, halo_samples_ = 0;
    void comm_mark(char kind, hipStream_t s);
    // hipGraph of one polling chunk of the fused PCG loop (launch-bound regime: small systems)
    hipGraphExec_t loop_graph_ = nullptr;
    struct GraphKey {
        const void *x = nullptr, *val = nullptr, *invd = nullptr;
        int n = 0, grid = 0, spmv_grid = 0, period = 0, R = 0, xcd = 0, chunk = 0;
        bool operator==(const GraphKey &o) const
        {
            return x == o.x && val == o.val && invd == o.invd && n == o.n && grid == o.grid && spmv_grid == o.spmv_grid &&
                   period == o.period && R == o.R && xcd == o.xcd && chunk == o.chunk;
        }
    } loop_graph_key_;
    void enqueue_fused_iteration(int par, const double *invd, double *d_x);
    int dist_spmv_dot(double *v_ext, double *y, double *part, const int *done_flag);
    void cg1_loop(const double *d_b, double *d_x, size_t &prof_used);
    DeviceBuffer<double> cg1_p_, cg1_s_;

    // distributed
    Comm comm_;
    int64_t n_global_ = -1, row_begin_ = 0, row_end_ = -1;
    HaloPlan plan_;
    DeviceBuffer<int> halo_dev_, send_idx_;
    DeviceBuffer<double> send_buf_;
    // overlap of the halo exchange with the interior rows of the SpMV
    DeviceBuffer<int> rb_interior_, rb_boundary_;
    int n_rb_interior_ = 0, n_rb_boundary_ = 0;
    hipStream_t comm_stream_ = nullptr;
    hipEvent_t ev_p_ready_ = nullptr, ev_halo_done_ = nullptr;
    void classify_row_blocks();
    void exchange_halo_on(double *d_ext, hipStream_t s);
    // shards: every rank assembles the WHOLE matrix (global column ids; rank q's rows at plan_.row_offsets[q]) from
    // the shards by grouped send / recv -- the input of the replicated AMG setup (amg.dist_global)
    void gather_global_matrix(DeviceBuffer<int> &gptr, DeviceBuffer<int> &gcol, DeviceBuffer<double> &gval, int64_t &gnnz);
    DeviceBuffer<int> glob_ptr_, glob_col_;
    DeviceBuffer<double> glob_val_;

    int dist_mode_used_ = 0; // of the last AMG setup on shards ("amg.dist_mode_used")
    std::unique_ptr<AmgHierarchy> amg_;
    std::unique_ptr<DistAmg> damg_; // shards, amg.dist_global 2: the hierarchy built on the shards
    std::unique_ptr<SchwarzPrecond> schwarz_;
    std::unique_ptr<IcPrecond> ic_;
    // z = M^-1 r for the preconditioners that are not fused into the PCG kernels (amg, schwarz)
    void apply_generic_precond(const double *d_r, double *d_z, const int *done_flag);
    friend class AmgHierarchy;
};

} // namespace psolve
