/*
 * psolve_hip.h -- C ABI of libpsolve_hip.so, the MI355X (gfx950) "HIP" linear-solver backend for
 * PolySolve.  This is the drop-in boundary: a `class HIPSolver : public polysolve::linear::Solver`
 * (polysolve_amd/host/HIPSolver.hpp, registered as Solver::create("HIP")) forwards each virtual of
 * the reference interface to exactly one entry point below.  Plain pointers and sizes only; no
 * C++/torch types; no function throws -- every call returns 0 on success or a negative
 * PSOLVE_HIP_E* code, with the message available from psolve_hip_last_error().
 *
 * Matrix layout at the boundary: the three arrays of a compressed
 * Eigen::SparseMatrix<double, ColMajor, int> (polysolve::StiffnessMatrix,
 * /root/reference/src/polysolve/Types.hpp:11-15): outer[n+1], inner[nnz], values[nnz].  For the SPD
 * (symmetric) systems this path serves, CSC arrays == CSR arrays, the same reinterpretation the
 * reference's AMGCL wrapper relies on (src/polysolve/linear/AMGCL.hpp:36-43, AMGCL.cpp:164-166).
 *
 * Threading: like every reference backend, one handle is used by one thread at a time; handles
 * are independent (own stream, own device buffers).
 */
#ifndef PSOLVE_HIP_H
#define PSOLVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSOLVE_HIP_ABI_VERSION 2

typedef struct psolve_hip_ctx *psolve_hip_t;

enum {
    PSOLVE_HIP_OK = 0,
    PSOLVE_HIP_EINVAL = -1,    /* bad argument / call order (e.g. solve before factorize)          */
    PSOLVE_HIP_EDEVICE = -2,   /* HIP runtime error (no device, out of memory, launch failure)     */
    PSOLVE_HIP_ENUMERIC = -3,  /* factorize: zero/NaN diagonal, non-finite values                  */
    PSOLVE_HIP_ECOMM = -4,     /* RCCL error                                                       */
    PSOLVE_HIP_ERANGE = -5     /* nnz or n does not fit int32 on one GPU (cf. BSRMatrix.cu:438-442) */
};

/* solver_status values, mirroring polysolve::linear::MASSolverStatus
 * (/root/reference/src/polysolve/linear/MASSolver.hpp:10-33). */
enum {
    PSOLVE_HIP_RUNNING = 0,
    PSOLVE_HIP_REACH_RELATIVE_TOLERANCE = 1,
    PSOLVE_HIP_REACH_ABSOLUTE_TOLERANCE = 2,
    PSOLVE_HIP_REACH_MAX_ITERATIONS = 3,
    PSOLVE_HIP_NONFINITE_RESIDUAL = 4 /* NaN/Inf in b, x0 or A (or CG breakdown): the loop stops at once;
                                         MAS throws "Invalid initial residual" here (MASSolver.cu:482-486) */
};

/* What get_info() reports.  Covers both key families of the reference:
 * Eigen  (EigenSolver.tpp:86-90):  solver_iter, solver_error
 * AMGCL  (AMGCL.cpp:130-144):      num_iterations, final_res_norm
 * MAS    (MASSolver.cu:214-219):   solver_status                                                   */
typedef struct psolve_hip_info {
    int64_t solver_iter;        /* Eigen's iterations(): completed direction updates                 */
    int64_t num_iterations;     /* AMGCL's count: passes through the CG loop (= SpMVs in the loop)   */
    double solver_error;        /* recurrence ||r|| / ||b|| at exit (Eigen error(), AMGCL final_res) */
    double final_res_norm;      /* same value, AMGCL's name                                          */
    double true_residual;       /* ||b - A x|| / ||b|| recomputed after the loop (-1 if disabled)    */
    double rhs_norm;            /* ||b||                                                             */
    int32_t solver_status;      /* PSOLVE_HIP_REACH_*                                                */
    int32_t amg_levels;         /* 0 unless precond == amg                                           */
    double time_analyze;        /* seconds, host wall                                                */
    double time_factorize;
    double time_solve;          /* includes H2D of b/x and D2H of x for the host entry point         */
    double time_solve_device;   /* device-resident part only                                         */
    double spmv_ms_avg;         /* HIP-event average of the sampled in-loop SpMV launches (0 if off) */
    int64_t spmv_samples;
} psolve_hip_info;

/* ---------------------------------------------------------------------------------------------
 * Lifecycle.  Replaces the backend constructor reached from Solver::create(solver, precond)
 * (/root/reference/src/polysolve/linear/Solver.cpp:307-496; model: the MAS branch :400-405 and
 * MASSolverImpl's ctor, MASSolver.cu:186-196 -- one device, one private stream).
 * ------------------------------------------------------------------------------------------- */
int psolve_hip_abi_version(void);
int psolve_hip_device_count(int *count);
int psolve_hip_create(psolve_hip_t *out, int device_id);
/* One Solver object over 1..8 GPUs of one node, inside the caller's process: the
 * handle owns one device context + one host thread per listed device and an in-process RCCL clique
 * (ncclCommInitAll).  It serves the HOST contract below unchanged -- set_param, analyze_pattern,
 * factorize (splits the rows itself: contiguous ranges balanced by nonzeros, cut at block_size multiples),
 * solve (scatters b / x, gathers x), get_info -- so Solver::create("HIP") with
 * params["HIP"]["devices"] = [0, 1, ...] reaches several GPUs from PolyFEM / Newton with no launcher.
 * Device-pointer entry points (a device pointer belongs to ONE device) return PSOLVE_HIP_EINVAL on
 * such a handle.  Repeated ids (e.g. {0, 0}) put several shards on one GPU through the host-synchronised
 * loopback group instead of RCCL: a test vehicle for boxes with fewer GPUs than shards, not a fast path.
 * n_devices == 1 is psolve_hip_create(out, device_ids[0]). */
int psolve_hip_create_multi(psolve_hip_t *out, const int *device_ids, int n_devices);
/* rows [*row_begin, *row_end) and device of shard `shard` after factorize (single-device handle: shard 0);
 * psolve_hip_get_param(h, "devices", &v) gives the shard count */
int psolve_hip_shard_rows(psolve_hip_t h, int shard, int64_t *row_begin, int64_t *row_end, int *device_id);
void psolve_hip_destroy(psolve_hip_t h);
const char *psolve_hip_last_error(psolve_hip_t h); /* h may be NULL: last create() error */

/* The instantiation PCG's own product (q = A p with the fused p.q) ran on in the last solve_device / solve of a single-device
 * handle, spelled as rocprofv3 prints it ("spmv_csr_pat<256, 1, true>", "spmv_csr_dma<256, 1, double, true, false, true>",
 * "spmv_bsr3_dma<1, 3, false>" ...): the bench's roofline line and the profiles under profiles/ name the same kernel because
 * both read it from the library (round 5).  Empty before the first solve. */
int psolve_hip_last_spmv_kernel(psolve_hip_t h, char *buf, int buf_len);
/* ... and the three kernels of a Jacobi-PCG iteration by number: which = 0 the product (as above), 1 the residual update
 * ("pcg_update_r_kernel<P>": r -= alpha q with the fused r.r and r.z), 2 the iterate / direction update
 * ("pcg_update_xp_kernel<P>": x += alpha p, p = z + beta p).  Their sampled durations in the last solve ("profile_spmv"):
 * psolve_hip_get_info's spmv_ms_avg, get_param "stats.update_r_ms_avg" / "stats.update_xp_ms_avg".  bench.py names the one
 * that takes most of the iteration in its roofline object. */
int psolve_hip_last_pcg_kernel(psolve_hip_t h, int which, char *buf, int buf_len);
/* Hand the device blocks this handle keeps for reuse (released allocations, "stats.device_bytes_cached"; at most
 * "lab.alloc_cache_mb" MiB; default min(16 GiB, device memory / 16)) back to the driver now: another handle, or the caller's own hipMalloc, gets the memory without
 * waiting for this handle's next failed allocation.  Synchronises the handle's stream. */
int psolve_hip_trim(psolve_hip_t h);

/* Adopt the caller's HIP stream (e.g. torch's current stream) instead of the private one.
 * NULL restores the private stream. */
int psolve_hip_set_stream(psolve_hip_t h, void *hip_stream);
int psolve_hip_synchronize(psolve_hip_t h);

/* ---------------------------------------------------------------------------------------------
 * set_parameters(json)  -- Solver.hpp:90; keys follow the reference's own spellings:
 *   "max_iter"            (/MAS/max_iter, EigenSolver.tpp:73-76)          default 10000
 *   "tolerance"           (EigenSolver.tpp:77-80) alias of "relative_tolerance"
 *   "relative_tolerance"  (/MAS/relative_tolerance) on ||r||/||b||          default 1e-8
 *   "absolute_tolerance"  (/MAS/absolute_tolerance) on ||r||                default 0
 *   "precond"             0 none (Eigen::IdentityPreconditioner), 1 jacobi
 *                         (Eigen::DiagonalPreconditioner), 2 amg (AMGCL.cpp:32-65), 3 schwarz: multilevel
 *                         additive Schwarz on 64-unknown dense domains, the wave64 re-think of the reference's
 *                         MAS preconditioner (mas_utils/MASPreconditioner.cu)                default 1
 *                         4 ic: incomplete Cholesky as Eigen::IncompleteCholesky computes it (scaling, shift, as many
 *                         entries per column as the matrix has), in Eigen's default approximate-minimum-degree ordering
 *                         ("ic.ordering"; both restated from the published algorithms, parity unpinned); ordered and
 *                         factorized on the host, applied on the device by two triangular solves in which every row waits
 *                         for the rows it depends on (on shards: of the shard's diagonal block)
 *   "ic.ordering"         precond 4: the ordering the matrix is factored in -- 1 approximate minimum degree
 *                         (Eigen::AMDOrdering<int>, the default of Eigen::IncompleteCholesky<double>), 0 natural   default 1
 *   "ic.initial_shift"    precond 4: Eigen's setInitialShift                                           default 1e-3
 *   "schwarz.levels"      precond 3: levels of 64-fold coarsening, 1..4 (1 = block Jacobi with dense 64 x 64
 *                         inverses; with block_size > 1 the coarse unknowns are per component)       default 1
 *   "block_size"          1 | 2 | 3 (AMGCL.cpp:111-113, /MAS/block_dim); any other value selects 1, the
 *                         scalar path, as the reference does (AMGCL.cpp:111-128)        default 1
 *   "check_period"        iterations enqueued between host polls             default 16
 *   "true_residual"       1: recompute ||b-Ax||/||b|| after the loop         default 1
 *   "profile_spmv"        k>0: HIP-event-time every k-th in-loop SpMV launch default 0
 *   "blocks_per_cu" "spmv_blocks_per_cu" "vec_blocks_per_cu"   persistent-grid sizes (vector kernels 8, SpMV 6; PCG's own fused vector kernels 2)
 *   "spmv_kernel"         1 LDS-DMA staged stream (round 2), 0 register-staged pipeline (round 1), 2 a
 *                         SELL-64-sigma copy (one row per lane), 3 the pattern dictionary (rows that repeat a
 *                         few column-offset patterns -- stencils, structured meshes -- multiply without the
 *                         column stream: 8 nnz + 22 n bytes instead of 12 nnz + 20 n; same columns, same order,
 *                         same sums), -1 by operator: the dictionary where there is one (get_param
 *                         "spmv_patterns" > 0), SELL for wide rows (>= 12 stored entries per row, no block copy),
 *                         DMA for operators streamed non-temporally or with several threads per row, the
 *                         pipeline for the rest                                                     default -1
 *   "spmv_nt"             non-temporal matrix stream + y stores: -1 for operators above "spmv_nt_mbytes" (384) MiB
 *                         -- smaller ones are re-read from the Infinity Cache; the results of a level operator over
 *                         vectors under 64 MiB are stored plainly, the fused vector kernels follow from 64 MiB per
 *                         vector on --, 0 off, 1 on                                                  default -1
 *   "spmv_xcd_map"        SpMV schedule: 0 round-robin row-blocks, 1 contiguous eighth per XCD,
 *                         2 chunks of "spmv_chunk_rows" (8192) rows dealt to the XCDs   default 2
 *   "spmv_rows_per_block" SpMV row-block height, 0 = auto from nnz / n       default 0
 *   "dist_overlap"        shards: interior-row SpMV overlaps the halo exchange  default 1
 *   "dist_single_reduction" shards, Jacobi / identity: Chronopoulos-Gear recurrences, ONE all-reduce of three
 *                         doubles per iteration instead of two all-reduces       default 1
 *                         -- on shards of at most "dist_single_reduction_max_rows" (3000000) rows (global rows /
 *                         ranks): the single-reduction step moves 16 n more bytes per iteration, which only pays
 *                         where the all-reduce latency is the iteration
 *   "dist_collectives"    one handle, several devices (psolve_hip_create_multi): 0 = RCCL for every exchange; 1 = the two
 *                         PER-ITERATION exchanges -- the all-reduce of the CG scalars and the halo of the PCG vector -- by
 *                         stores into the peers' memory (xGMI peer mapping) and epoch flags: one / two small launches, no
 *                         library call; same numbers (contributions added in rank order).  Needs devices that map each
 *                         other ("dist.peer_available"); setup-time exchanges stay on RCCL      default 0
 *   "use_bsr3"            block_size 3: fine-level products on a 3x3-block copy (76 B / 9 entries)  default 1
 *   "spmv_value_dict"     operators whose rows repeat a few column-offset patterns AND their values bit for bit (a
 *                         constant-coefficient stencil, one material on a structured mesh): the products stream a 16-bit
 *                         "row kind" per row and no matrix at all (2 n + the vectors instead of 8 nnz + 22 n bytes), the
 *                         kinds' offsets and values in LDS -- the same values times the same entries of x in the same
 *                         order, bit-equal sums (get_param "spmv_row_kinds" > 0 when active).  Built from the values of
 *                         every factorize; absent when the rows do not repeat (the usual FEM matrix), at the cost of one
 *                         early-ending pass.  "spmv_kernel" 3 keeps the dictionary kernel with the value stream.  With
 *                         block_size 3 the same switch covers BLOCK rows that repeat their block offsets and values
 *                         (one material on a structured mesh): a 16-bit kind per node, (offset, block id) lists and the
 *                         distinct 3x3 blocks in LDS, no block stream (get_param "bsr3_row_kinds", "bsr3_kind_blocks").
 *                         Where kinds exist, Jacobi-PCG's vector kernels read 1 / diag as table[kind[row]] (2 bytes per
 *                         row instead of 8; get_param "pcg_kind_diag"); get_param "spmv_slots" > 0: the product runs in
 *                         the slot form (at most 8 distinct offsets)                                          default 1
 *   "spmv_col16"          operators without a dictionary / block / SELL copy whose row-blocks touch at most eight
 *                         8192-column windows -- any local numbering: a grid, a breadth-first order ("reorder"), a coarse
 *                         AMG level -- stream 16-bit columns (window, offset) instead of 32-bit ones: 10 instead of 12
 *                         bytes per entry, the same columns in the same order, bit-equal sums (get_param
 *                         "col16_active"); "spmv_kernel" 1 keeps the plain 12-byte stream.  Pays where the
 *                         product is HBM-bound (256^3 renumbered: 312 -> 287 us per product), nothing on cache-resident
 *                         operators and coarse AMG levels (latency of the gathers, not the stream)        default 0
 *   "reorder"             single device: renumber the system at factorize for the locality of the products' gathers
 *                         (Cuthill-McKee by breadth-first levels, built on the device; whole nodes move with block_size
 *                         2 / 3).  The renumbered copy, the preconditioner and the PCG vectors live in the new numbering,
 *                         b and x are permuted on the way in and out, every entry point keeps the caller's numbering;
 *                         the order is kept while the pattern stays the same (Newton).  Precedent: MAS permutes the
 *                         system by a graph partition (mas_utils/GraphPartition.cpp:240-243).  0 off -- the caller's
 *                         numbering, row sums bit-equal to the reference loop's --, 1 always, 2 auto: with the identity /
 *                         Jacobi preconditioners (PCG's iterates do not depend on the numbering) and with amg (the
 *                         aggregation sweep follows the numbering: the hierarchy is AMGCL's hierarchy of the renumbered
 *                         matrix), not with ic / schwarz (the elimination order / the domains ARE the numbering:
 *                         renumbered on request), on systems of at least "reorder_min_rows" (131072) rows whose numbering spreads the
 *                         gathers of 64 consecutive rows over more than "reorder_min_spread" (2.5) times the fewest
 *                         cache lines they could occupy, and only if the search improves that figure by a tenth
 *                         (get_param "reorder.active" / ".spread_before" / ".spread_after" / ".levels" / ".seconds";
 *                         psolve_hip_reorder_perm).  A multi-device handle renumbers BEFORE it partitions (the order
 *                         of the whole pattern is searched on its first device; contiguous row ranges of that order are
 *                         slabs of the mesh: a shard's halo is two frontiers of the search instead of most of the vector);
 *                         shards set up by the caller (comm_init + set_partition) keep the caller's numbering  default 2
 *                         (the library reads no environment variable for this; the Python test mirror,
 *                         polysolve_amd/solver.py, presets "reorder" and "reorder_min_rows" 0 from PSOLVE_REORDER so that
 *                         a whole test run can be put under a forced renumbering)
 *   "lab.dma_tile_max" "lab.verbose" "lab.var_row_blocks" "lab.symbolic_bitmap" "lab.agg_two_pass_assign" "lab.kind_*" "lab.bsr3_kinds"
 *   "lab.stage_kb" "lab.alloc_cache_mb" "lab.alloc_cache_poison" "lab.alternate"     knobs of THIS handle since round 6 (process-wide until
 *                         round 5; the levels of an AMG hierarchy see a changed knob from the next factorize on): measurement knobs of profiles/r04_level1.md and of the A/B tests (the largest LDS tile of the
 *                         wide-row product, the entries a row-block may hold when its height is chosen, tile head-room, a
 *                         trace of refresh decisions on stderr; 0 switches off: row-blocks packed to the tile, the LDS
 *                         bitmap of the symbolic products; "lab.alternate" 8: all products of a cycle sweep forward,
 *                         1: psolve_hip_time_spmv alternates the direction, 2: results always stored like the matrix is
 *                         loaded).  Per handle, set-only, not in the /HIP spec and not part of the contract:
 *                         their defaults are the shipped behaviour.  (The library looks at three environment variables, none
 *                         of which changes a result: PSOLVE_TIMING -- when set, factorize prints the wall time of its phases
 *                         on stderr, synchronising after each --, PSOLVE_HIP_FORCE_LOOPBACK, the multi-device handle's test
 *                         vehicle for boxes with one GPU, and PSOLVE_ALLOC_CACHE_POISON=1, the default of
 *                         "lab.alloc_cache_poison" for handles created afterwards: recycled device blocks arrive full of
 *                         0xFF bytes (=2: fresh blocks from the driver too) -- how the test session proves that nothing
 *                         reads an allocation before writing it; PSOLVE_SWEEP_LIMIT_MS / PSOLVE_SWEEP_DEBUG, debugging aids of
 *                         the ordered relaxations (amg_sweep.hip))
 *   "reorder_reverse"     the breadth-first order read backwards (reverse Cuthill-McKee): the same bandwidth and gather
 *                         locality; AMGCL's aggregation sweep, which follows the numbering, builds more regular aggregates
 *                         against the search direction than along it (configs[2] with its nodes in a random order: 40 PCG
 *                         iterations against 52; the grid numbering: 37)  default 1
 *   "amg.max_levels" "amg.coarse_enough" "amg.ncycle" "amg.npre" "amg.npost"
 *   "amg.eps_strong" "amg.sa_relax" "amg.estimate_spectral_radius" "amg.sa_power_iters"
 *   "amg.cheb_degree" "amg.cheb_power_iters" "amg.cheb_higher" "amg.cheb_lower"
 *                         (names and defaults of AMGCL.cpp:32-65, except ncycle = 1 and
 *                         cheb_degree / cheb_power_iters which default to the V-cycle north_star asks for)
 *   "amg.reuse"           same sparsity pattern at the next factorize: keep aggregates and patterns,
 *                         recompute the numbers by kernels                     default 1
 *   "amg.device_setup"    build the hierarchy on the device; 0 = all-host construction, uploaded   default 1
 *   "amg.matrix_fp32"     the operators inside the cycle (A_l, P_l, R_l) stream single-precision VALUES (8 B per
 *                         nonzero instead of 12); vectors and arithmetic stay double, PCG's own product uses
 *                         the original matrix; faster cycle, a slightly different preconditioner   default 0
 *   "amg.block_levels"    block_size 3 (AMGCL_Block<3>, AMGCL.cpp:243-302: the block value type end to end): the operators of
 *                         the cycle below the finest level, the prolongations and the restrictions multiply through 3x3-block
 *                         copies (76 B and 3 gathers per block instead of 108 B and 9), and the block-scaled Chebyshev step is
 *                         an epilogue of the block product (one launch per step, no residual vector); 0: scalar CSR below
 *                         level 0 and residual product + node-local update per step                          default 1
 *   "amg.dist_global"     several devices, scalar systems.  2: the hierarchy is built ON the shards -- aggregates confined
 *                         to a shard, the halo rows of P and A P fetched from their owners for the Galerkin products,
 *                         every operator row-partitioned like the matrix, a level with fewer than
 *                         "amg.dist_replicate_rows" (50000) rows per device gathered and the rest replicated; memory
 *                         and setup scale with the devices, iteration counts stay within ~1.3x of one device's.
 *                         1: the single-device hierarchy, built by every rank from the gathered matrix (level 0 applied
 *                         on the shard, coarser levels replicated): exact single-device iteration counts, but O(global)
 *                         memory per rank -- matrices above "amg.dist_global_max_mbytes" (4096) take 2 instead.
 *                         0: one hierarchy per shard (additive Schwarz).  "amg.eps_strong" != 0 (the distributed setup
 *                         serves 0): scalar systems take 1 where the matrix fits, else 0.  get_param
 *                         "amg.dist_mode_used" reports what the last factorize came to                 default 2
 *   "amg.renumber"        single device, scalar: renumber levels >= 1 of at least "amg.renumber_min_rows" (65536) rows for
 *                         locality after the setup (psolve_hip_amg_level_perm)                         default 0
 *   "amg.device_aggregation" the aggregation sweep on the device (same aggregates as the sequential loop): one
 *                         kernel in which every vertex waits for the earlier vertices it depends on, or with
 *                         "amg.aggregation_rounds" 1 as dependency rounds (two kernels per round); levels under
 *                         "amg.aggregation_min_rows" (100000) rows, or deeper than "amg.aggregation_max_rounds"
 *                         (10000) rounds / 10 us per round of waiting, use the host loop            default 1
 *   round 5 -- amgcl's other runtime classes (the reference forwards these names as free strings: linear-solver-spec.json:
 *   393-397, 423-427, AMGCL.cpp:67-92; codes here, names in the JSON spec and the adapters):
 *   "amg.relax_type"      0 chebyshev, 1 damped_jacobi ("amg.damping", 0.72), 2 spai0, 3 gauss_seidel (a forward sweep    default 0
 *                         before, a backward sweep after the coarse correction), 4 ilu0 ("amg.ilu_damping", 1.0) -- 3 and 4 are
 *                         sweeps in row order on the device: a row waits for the rows it depends on (amg_sweep.hip)
 *   "amg.class"           0 amg, 1 relaxation: the smoother of the system matrix alone is the preconditioner                default 0
 *                         (/AMGCL/precond/class, amgcl::relaxation::as_preconditioner)
 *   "amg.cheb_scale"      chebyshev.scale: 0 = no diagonal scaling of the residual (needs cheb_power_iters > 0)     default 1
 *   "amg.coarsening"      0 smoothed_aggregation, 1 aggregation (P = tentative prolongation, the Galerkin operator scaled by
 *                         1 / "amg.over_interp"; 0 = amgcl's default 1.5 scalar / 2.0 block value types)            default 0
 *   "amg.direct_coarse"   1: the coarsest level (at most 4096 rows; larger ones are refused) is solved -- dense inverse
 *                         built on the device at factorize, one dense product per visit -- instead of relaxed  default 0
 *   "amg.coarse_dense"    a RELAXED coarsest level of at most this many rows is applied as one dense operator built at
 *                         factorize (the smoother's recurrence run on the identity): the same operator up to rounding, one
 *                         launch per visit instead of (npre + npost) x degree; 0 off                           default 1024
 *   round 5 -- NOT amgcl's arithmetic, opt-in:
 *   "amg.aggregation"     0 amgcl: plain_aggregates' sequential sweep, reproduced exactly; 1 parallel: a distance-2 maximal
 *                         independent set by hashed priorities in a dozen synchronous rounds, the sweep's membership rule
 *                         (restated in oracle/amg_oracle.c; scalar stencil-like operators: same iteration counts, a first
 *                         factorize without the sweep's dependency chain; 27-point block operators: 1.7 x the iterations);
 *                         2 compact (round 6): ONE-hop aggregates around two generations of such sets, the rest by most
 *                         connections -- the sweep's aggregate sizes on 27-point node graphs (meant to go with
 *                         "amg.direct_coarse": its hierarchies end a level earlier)                              default 0
 *   "amg.refresh_power_iters" -1: a factorize of the same pattern estimates the smoothers' radii like a first one; k >= 0: it
 *                         continues the power iteration from the vector the previous factorize ended with for k steps
 *                         (0 keeps the radii): a third of a refresh is those iterations                        default -1
 *   "amg.overlap_smoothers" the smoothers' power iterations on a second stream                                   default 1
 *   "fault.solve_rank"    TEST HOOK (set only; not part of the JSON spec): the shard of this rank throws at the start
 *                         of its next solve, once, before its first collective -- how the tests reach the path on
 *                         which a multi-device handle frees the shards blocked in a collective (loopback: wake-up;
 *                         in-process RCCL clique: ncclCommAbort, the clique is made again at the next call)
 * Unknown key -> PSOLVE_HIP_EINVAL.
 * ------------------------------------------------------------------------------------------- */
int psolve_hip_set_param(psolve_hip_t h, const char *key, double value);
int psolve_hip_get_param(psolve_hip_t h, const char *key, double *value);
/* The built-in default of a parameter; needs no handle and no GPU.  This is what the `/HIP` objects of
 * integration/linear-solver-spec.hip.json (the rules a PolySolve build merges into its
 * linear-solver-spec.json) are checked against.  Unknown key -> PSOLVE_HIP_EINVAL. */
int psolve_hip_default_param(const char *key, double *value);

/* ---------------------------------------------------------------------------------------------
 * The reference contract on HOST arrays.
 *   analyze_pattern(A, precond_num)  Solver.hpp:96   -> psolve_hip_analyze_pattern
 *   factorize(A)                     Solver.hpp:99   -> psolve_hip_factorize
 *   solve(b, x)                      Solver.hpp:128  -> psolve_hip_solve   (x: guess in, solution out)
 *   get_info(json&)                  Solver.hpp:93   -> psolve_hip_get_info
 * factorize copies/uploads what it needs (the caller keeps ownership of A, like
 * EigenSolver.tpp:101-105 and BSRMatrix.cu:210-231); it may be called repeatedly with new values
 * and the same or a different pattern (tests/test_linear_solver.cpp:260-295, Newton.cpp:189-193).
 * factorize fails with PSOLVE_HIP_ENUMERIC on a non-finite diagonal entry (the adapter turns that into
 * std::runtime_error, which Newton catches, Newton.cpp:191-202); a ZERO diagonal entry is not an error:
 * Jacobi scales that row by 1, Eigen::DiagonalPreconditioner's rule.  solve returns 0 on
 * non-convergence (inspect get_info), like Eigen/AMGCL.
 * ------------------------------------------------------------------------------------------- */
int psolve_hip_analyze_pattern(psolve_hip_t h, int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner,
                               int precond_num);
int psolve_hip_factorize(psolve_hip_t h, int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner,
                         const double *values);
int psolve_hip_solve(psolve_hip_t h, const double *b, double *x_inout);
int psolve_hip_get_info(psolve_hip_t h, psolve_hip_info *info);

/* ---------------------------------------------------------------------------------------------
 * Device-resident entry points (bench, multi-GPU shards, torch tensors via data_ptr()).
 * All pointers are device pointers on the handle's device.  Rows are the handle's LOCAL rows
 * [row_begin, row_end) of the global system; column ids are GLOBAL (see psolve_hip_set_partition).
 * ------------------------------------------------------------------------------------------- */
/* The CSR arrays are adopted without copy, are never written, and must outlive the handle's use of them.
 * (On a shard the handle keeps a private copy of d_col translated to local ids; the caller's array keeps
 * its global ids, so the same arrays may be factorized again -- Newton with a constant pattern.) */
int psolve_hip_factorize_device(psolve_hip_t h, int64_t n_local, int64_t nnz_local, const int32_t *d_rowptr,
                                const int32_t *d_col, const double *d_values);
int psolve_hip_solve_device(psolve_hip_t h, const double *d_b, double *d_x_inout);

/* Synthetic 7-point Poisson shard: rows of the z-planes [z0, z1) of an
 * nx*ny*nz grid, diag 6 / off-diag -1, generated on the device, then factorized. */
int psolve_hip_generate_poisson7(psolve_hip_t h, int nx, int ny, int nz, int z0, int z1);
/* Synthetic block-3 elasticity system (SURVEY.md 8(d) "Elasticity-Q1(M)", BASELINE.json configs[2]): trilinear
 * hexahedra on an M^3-node unit cube, Young's modulus E, Poisson ratio nu, 2x2x2 Gauss points, dofs 3 node + c,
 * the face x = 0 clamped by identity rows / columns as FEMSolver.cpp:136-161 does; 3 M^3 rows (M = 100: 3e6 DOF,
 * 2.4e8 nonzeros), generated on the device, then factorized with the handle's current parameters. */
int psolve_hip_generate_elasticity_q1(psolve_hip_t h, int M, double E, double nu);
/* The same stiffness matrix with its NODES renumbered pseudo-randomly (mode 1: all nodes, 2: inside windows of `window`
 * nodes, 0: the grid numbering; psolve_hip_permutation(M^3, ...) is the node renumbering); the three displacements of a
 * node stay together, so the 3 x 3 blocks stay blocks: what an unstructured mesh's numbering does to configs[2]. */
int psolve_hip_generate_elasticity_q1_permuted(psolve_hip_t h, int M, double E, double nu, int mode, int64_t window,
                                               uint64_t seed);
/* The same 7-point Poisson system under a symmetric pseudo-random renumbering, B = Pi A Pi^T with sorted columns:
 * mode 1 permutes all rows (every gather its own cache line: the worst case), mode 2 shuffles the rows inside
 * consecutive windows of `window` rows (the locality of a mesh generator's numbering).  No column-offset pattern
 * repeats, so the products run on the plain CSR stream: the unstructured leg of bench.py (SURVEY.md 8(d) "report
 * index compression separately").  psolve_hip_permutation (host only, no GPU) returns the renumbering itself:
 * new_index[i] = row of B that original row i became. */
int psolve_hip_generate_poisson7_permuted(psolve_hip_t h, int nx, int ny, int nz, int mode, int64_t window,
                                          uint64_t seed);
int psolve_hip_permutation(int64_t n, int mode, int64_t window, uint64_t seed, int32_t *new_index);
/* d_b = A * x_star with x_star[r] = U(-1,1) from SplitMix64(seed + global row r); d_xstar (local
 * rows, may be NULL) receives x_star. */
int psolve_hip_generate_rhs(psolve_hip_t h, uint64_t seed, double *d_b, double *d_xstar);

/* Hot-path kernels one by one, on the factorized matrix (parity tests, roofline bench). */
int psolve_hip_spmv_device(psolve_hip_t h, const double *d_x, double *d_y);                  /* y = A x       */
int psolve_hip_spmv_dot_device(psolve_hip_t h, const double *d_x, double *d_y, double *xy);  /* + x.y (host)  */
int psolve_hip_dot_device(psolve_hip_t h, int64_t n, const double *d_a, const double *d_b, double *out_host);
int psolve_hip_axpby_device(psolve_hip_t h, int64_t n, double a, const double *d_x, double b, double *d_y);
int psolve_hip_precond_apply_device(psolve_hip_t h, const double *d_r, double *d_z);         /* z = M^-1 r    */
/* Time `reps` back-to-back SpMV launches with HIP events on the handle's stream; *ms_avg = mean. */
int psolve_hip_time_spmv(psolve_hip_t h, const double *d_x, double *d_y, int reps, double *ms_avg);
int psolve_hip_time_vecops(psolve_hip_t h, int reps, double *ms_update_avg, double *ms_direction_avg);
/* What this box's memory system does (bench.py's "box.probe"; no reference counterpart): out[0..2] = latency in ns of a
 * dependent load with a working set of 1 MiB (L2) / 64 MiB (Infinity Cache) / 1 GiB (HBM); out[3..4] = independent 8-byte
 * gathers per second (G/s) from a 2 MiB / 64 MiB vector; out[5] = ticks of the shader-cycle counter (s_memtime) per
 * microsecond while every CU runs an FMA loop, out[6] = length of that window in us.  n_out >= 7.  Takes ~0.1 s and 1 GiB
 * of device memory. */
int psolve_hip_box_probe(psolve_hip_t h, double *out, int n_out);

/* host <-> device helpers so a caller needs no HIP runtime of its own */
int psolve_hip_malloc(psolve_hip_t h, void **d_ptr, size_t bytes);
int psolve_hip_free(psolve_hip_t h, void *d_ptr);
int psolve_hip_memcpy_h2d(psolve_hip_t h, void *d_dst, const void *src, size_t bytes);
int psolve_hip_memcpy_d2h(psolve_hip_t h, void *dst, const void *d_src, size_t bytes);
int psolve_hip_matrix_shape(psolve_hip_t h, int64_t *n_local, int64_t *nnz_local, int64_t *n_halo);
/* The factorized matrix of a single-device handle, copied back to host arrays (rowptr[n + 1], col[nnz], val[nnz]; any
 * of them may be NULL): how bench.py brings a device-generated system to the host to time the HOST contract on it.  With
 * "reorder" active these are the arrays of the renumbered system (psolve_hip_reorder_perm). */
int psolve_hip_matrix_copy(psolve_hip_t h, int32_t *rowptr, int32_t *col, double *val);
/* AMG hierarchy introspection (precond == amg, after factorize): rows / nnz of level `level` and the
 * spectral-radius estimate rho(D^-1 A) its Chebyshev smoother uses.  get_info().amg_levels = count. */
int psolve_hip_amg_level_info(psolve_hip_t h, int level, int64_t *rows, int64_t *nnz, double *rho);
/* The matrices of the device-resident hierarchy, copied back for inspection: `what` 0 = A_l, 1 = P_l,
 * 2 = R_l, 3 = A_l P_l (the intermediate of the Galerkin product, kept for the numeric refresh; device setup only) -- P, R,
 * A P absent on the coarsest level -> PSOLVE_HIP_EINVAL; out = {rows, cols, nnz}. */
int psolve_hip_amg_level_matrix_shape(psolve_hip_t h, int level, int what, int64_t out[3]);
/* HIP-event times (us per launch, mean of `reps`) of the cycle's operations on level `level`, launched on the hierarchy's
 * own operators (bench.py's per-level block; single-device hierarchies): out_us[0] one Chebyshev step (product + fused
 * update), [1] residual, [2] restriction to the next level, [3] prolongation from it ([2], [3]: 0 on the coarsest level),
 * [4] the first Chebyshev step from x = 0 (no product). */
int psolve_hip_amg_time_level_ops(psolve_hip_t h, int level, int reps, double out_us[5]);
int psolve_hip_amg_level_matrix_copy(psolve_hip_t h, int level, int what, int32_t *rowptr, int32_t *col,
                                     double *val);

/* "amg.renumber": levels >= 1 may be renumbered for locality after the setup (the same hierarchy under a symmetric
 * permutation per level).  perm[i] (length = rows of the level) = row of the level's operator that row i of the setup's
 * own numbering -- AMGCL's: the order in which the aggregation sweep creates the aggregates -- became; the identity
 * (and *renumbered = 0) where the level kept its numbering.  The matrices psolve_hip_amg_level_matrix_copy returns are
 * in the renumbered ordering: A_l = Pi_l A Pi_l^T, P_l = Pi_l P Pi_{l+1}^T. */
int psolve_hip_amg_level_perm(psolve_hip_t h, int level, int32_t *perm, int *renumbered);

/* "reorder": new_of_old[i] (length n) = row of the factorized system that row i of the caller's numbering became;
 * *reordered = 0 (and new_of_old untouched) where the system kept the caller's numbering.  The factorized operator is
 * Pi A Pi^T with sorted columns; level 0 of psolve_hip_amg_level_matrix_copy is in that numbering. */
int psolve_hip_reorder_perm(psolve_hip_t h, int32_t *new_of_old, int *reordered);

/* Host-only half of factorize(precond = ic): Eigen::IncompleteCholesky<double, Lower, NaturalOrdering<int>> -- scaled,
 * shifted, left-looking incomplete Cholesky that keeps as many entries per column as the matrix column has (the
 * preconditioner behind the name "Eigen::IncompleteCholesky", /root/reference/src/polysolve/linear/Solver.cpp:179-183,
 * without the reference's AMD ordering).  Needs no GPU; exported so that the factor can be compared with the CPU
 * oracle's.  Input: CSC (= CSR of a symmetric matrix) arrays with sorted inner indices; output: L by columns (diagonal
 * first; colptr[n + 1], rowidx / vals of as many entries as the input has with row >= column) and the scaling S;
 * M^-1 = S L^-T L^-1 S. */
int psolve_hip_ic_host_factorize(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                                 double initial_shift, int32_t *colptr, int32_t *rowidx, double *vals, double *scale,
                                 double *shift, int *attempts);

/* Host-only half of factorize(precond = amg): the smoothed-aggregation hierarchy (aggregation,
 * smoothed prolongation, Galerkin products) for the coarsening parameters of AMGCL.cpp:32-65.  Needs
 * no GPU; exported so that the hierarchy can be compared with the CPU oracle's level by level.
 * `what`: 0 = A_l, 1 = P_l, 2 = R_l (P, R absent on the coarsest level -> PSOLVE_HIP_EINVAL). */
typedef struct psolve_hip_amg_host *psolve_hip_amg_host_t;
int psolve_hip_amg_host_build(psolve_hip_amg_host_t *out, int64_t n, int64_t nnz, const int32_t *rowptr,
                              const int32_t *col, const double *val, int max_levels, int coarse_enough,
                              double eps_strong, double sa_relax, int estimate_spectral_radius, int block_size,
                              int *n_levels);
/* round 5: the same with "amg.aggregation" (0 amgcl's sweep, 1 parallel, 2 compact), "amg.coarsening" (0 smoothed_aggregation,
 * 1 aggregation) and "amg.over_interp" */
int psolve_hip_amg_host_build2(psolve_hip_amg_host_t *out, int64_t n, int64_t nnz, const int32_t *rowptr,
                               const int32_t *col, const double *val, int max_levels, int coarse_enough,
                               double eps_strong, double sa_relax, int estimate_spectral_radius, int block_size,
                               int aggregation, int coarsening, double over_interp, int *n_levels);
int psolve_hip_amg_host_level_shape(psolve_hip_amg_host_t H, int level, int what, int64_t out[3], double *omega);
int psolve_hip_amg_host_level_copy(psolve_hip_amg_host_t H, int level, int what, int32_t *rowptr, int32_t *col,
                                   double *val);
void psolve_hip_amg_host_free(psolve_hip_amg_host_t H);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: 1-D row partition, one handle (one process) per GPU, RCCL over xGMI.  No counterpart
 * in the reference (no collective call sites: SURVEY.md section 2); new design, SURVEY.md 8(e).
 *   - psolve_hip_comm_unique_id: rank 0 creates the RCCL id; the host side broadcasts the 128
 *     bytes by any means (bench.py uses torch.distributed).
 *   - psolve_hip_comm_init: ncclCommInitRank on the handle's device.
 *   - psolve_hip_set_partition: this handle owns global rows [row_begin, row_end).
 * After that factorize_device/solve_device work on the shard: CG dot products are
 * ncclAllReduce'd, halo x entries travel by grouped ncclSend/ncclRecv between neighbours.
 * ------------------------------------------------------------------------------------------- */
#define PSOLVE_HIP_UNIQUE_ID_BYTES 128
int psolve_hip_comm_unique_id(char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path);
int psolve_hip_comm_init(psolve_hip_t h, int rank, int world, const char id[PSOLVE_HIP_UNIQUE_ID_BYTES],
                         const char *rccl_path);
int psolve_hip_set_partition(psolve_hip_t h, int64_t n_global, int64_t row_begin, int64_t row_end);

/* In-process loopback communicator: `world` handles of ONE process (one thread each; they may all sit
 * on the same GPU) exchange through host-synchronised device copies.  RCCL refuses two ranks on one
 * device, so this is how the distributed path (halo plan, column remap, pack/exchange, all-reduced CG
 * scalars) is exercised on real kernels on a box with fewer GPUs than ranks.  Not a performance path. */
typedef struct psolve_hip_local_group *psolve_hip_local_group_t;
int psolve_hip_local_group_create(psolve_hip_local_group_t *out, int world);
void psolve_hip_local_group_destroy(psolve_hip_local_group_t g);
int psolve_hip_comm_init_local(psolve_hip_t h, psolve_hip_local_group_t g, int rank);

/* Host-only: the 64-bit hashes (out[0]: outer, out[1]: inner) by which factorize(host arrays) recognises a pattern it
 * still holds on the device and then uploads the values only (8 of the 12 bytes per stored entry; get_param
 * "stats.pattern_uploads" / "stats.h2d_bytes").  Computed by `threads` host threads (0: half the hardware threads, at
 * most 16) while the values travel; the result does not depend on the number of threads.  No GPU needed. */
int psolve_hip_host_pattern_hash(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int threads,
                                 uint64_t out[2]);

/* Host-only: the ordering precond = "ic" factors in by default ("ic.ordering" 1): Eigen::AMDOrdering<int> restated
 * (amd_order.cpp; Eigen/src/OrderingMethods/Amd.h = CSparse's cs_amd) -- order[k] = the k-th pivot of the approximate
 * minimum degree ordering of the symmetric pattern (outer, inner: both triangles and the diagonal).  No GPU needed. */
int psolve_hip_amd_order(int64_t n, const int32_t *outer, const int32_t *inner, int32_t *order);

/* Host-only: the row partition a multi-device handle's factorize uses -- `world` contiguous ranges with about
 * nnz / world stored entries each, cut at multiples of block_size; row_offsets[world + 1].  No GPU needed. */
int psolve_hip_partition_rows(int64_t n, const int32_t *outer, int world, int block_size, int64_t *row_offsets);

/* Host-only halo planning (no GPU needed; also what the gloo CPU tests drive).  From the global
 * column ids of a shard (any order, duplicates allowed) compute the sorted unique list of
 * off-shard columns and, per owning rank, how many of them it owns.  row_offsets[world+1] is the
 * partition.  halo_out must hold n_cols entries at most; returns the halo count in *n_halo. */
int psolve_hip_plan_halo(int rank, int world, const int64_t *row_offsets, int64_t n_cols, const int32_t *cols,
                         int32_t *halo_out, int64_t *n_halo, int64_t *recv_counts /* [world] */);

#ifdef __cplusplus
}
#endif
#endif /* PSOLVE_HIP_H */
