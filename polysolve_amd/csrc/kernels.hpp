// kernels.hpp -- launchers of the hand-written gfx950 kernels of the PCG hot path.
//
// Reference analogues (kernel table): cusparseSpMV (MASSolver.cu:271-290),
// inner_product_kernel (mas_utils/InnerProduct.cu:16-45), axpby / scalar_division device lambdas
// (MASSolver.cu:46-81).  Nothing here is a translation of those: the CSR SpMV is an LDS-staged
// row-block stream with XCD-contiguous row ranges, reductions are deterministic two-level partial
// sums (no atomics), and CG's scalars never leave the device.
#pragma once
#include <vector>
#include "common.hpp"
#include <hip/hip_runtime.h>

#include <cstdint>

namespace psolve {

constexpr int kBlock = 256;        // threads per workgroup = rows per SpMV row-block (4 waves)
constexpr int kMaxPartials = 4096; // upper bound on persistent-grid size (partials per reduction)

// Device-resident CG scalars.  Every field that several workgroups read while one writes is
// double-buffered on iteration parity so that no kernel reads a word another workgroup of the SAME
// launch writes (visibility then needs nothing beyond the kernel boundary).
struct PcgState {
    double rz[2];      // r.z ("absNew" in Eigen's conjugate_gradient), ping-pong on parity
    int done[2];       // convergence latch, ping-pong on parity
    double rhs_norm2;  // ||b||^2
    double threshold;  // max(rel^2 ||b||^2, abs^2, DBL_MIN)
    double abs2;       // abs_tol^2 (to tell which tolerance fired)
    double rn2;        // ||r||^2 of the last completed pass (recurrence residual)
    double rn2_init;   // ||b - A x0||^2
    int passes;        // completed passes through the loop (AMGCL's iteration count)
    int status;        // PSOLVE_HIP_REACH_*
    int zero_rhs;      // ||b|| == 0: Eigen returns x = 0
    int pad;
    double alpha[2];   // single-reduction loop: the previous step length, ping-pong on parity
};

// 3x3-block view of the same matrix (block_size 3): block rows / block column ids / 9 values per
// block, row-major, zero-filled.  76 B per block of 9 entries instead of 108 B in CSR.
// Round 5, block-row kinds of a 3x3-block operator whose block rows repeat their block-column offsets AND their values bit
// for bit (constant-coefficient elasticity on a structured mesh: 28 kinds at any size): a 16-bit kind per block row; a
// kind is a list of (offset, block id), the distinct 3x3 blocks (a few hundred) are kept once.  The product reads both
// tables from LDS and streams no matrix: 2 + 24 + 24 bytes per node (kinds, x, y) instead of 76 per BLOCK.  Row sums in
// column order (the scalar CSR loop's order).  Built from the values of every factorize (pattern.hip: Bsr3Kinds).
struct Bsr3KindDev {
    const unsigned short *kind = nullptr; // [nb]
    const int *koff = nullptr;            // [nk * kml] block column - block row; padded with 0
    const unsigned short *kblk = nullptr; // [nk * kml] id of the 3x3 block; padded with the id of an all-zero block
    const double *blocks = nullptr;       // [nblk * 9] row-major 3x3 blocks
    const int *krep = nullptr;            // [nk] a block row of each kind (its smallest): what is a function of the block row's
                                          // contents alone need be computed for these nk rows only
    int nk = 0, kml = 0, nblk = 0;
};
constexpr int kBsrKindLdsBytes = 40 * 1024;

struct Bsr3Dev {
    const Bsr3KindDev *kinds = nullptr; // when set, the products run from the block-row kinds (no matrix stream)
    int nb = 0;
    int64_t nnzb = 0;
    const int *rowptr = nullptr;
    const int *col = nullptr;
    const double *val = nullptr;
    const float *val32 = nullptr; // when set, the products stream these single-precision copies (40 B per block)
    int brows_per_group = 8; // block rows per workgroup step (<= 254 blocks on average)
};

// SELL-64-sigma copy of a CSR operator whose rows are wide (coarse AMG levels, Q1 elasticity as CSR): slices of
// 64 rows -- one row per lane of a wave -- stored column-major and padded to the slice's longest row, rows sorted
// by length inside windows of kSellWindow rows so that the padding stays at a few per cent.  A wave streams its
// slice with whole-line loads straight into registers and every lane sums ITS row in column order (the scalar
// loop's order).  sell.hpp builds it on the device.
constexpr int kSellWindow = 256; // = the rows of a workgroup step: its four waves share the window's x lines in L1
struct SellDev {
    int nslices = 0;
    const int *slice_ptr = nullptr; // [nslices + 1] first entry of a slice (multiples of 64)
    const int *col = nullptr;       // entry (slice s, position k, lane l) at slice_ptr[s] + 64 k + l
    const double *val = nullptr;
    const int2 *slot = nullptr;     // [64 nslices] (row, stored entries) of a lane; row -1: padding lane
};

// Pattern dictionary of a CSR operator whose rows repeat a few column-offset patterns (col - row): stencils on
// structured grids, FEM on (semi-)structured meshes.  The product then needs no column stream at all -- a row
// carries a 16-bit pattern id, the offsets of the few patterns sit in the caches -- and moves 8 nnz + 22 n bytes
// instead of 12 nnz + 20 n (7-point 256^3: 1.31 GB instead of 1.74 GB).  Same columns in the same order, so the
// same sums bit for bit.  pattern.hip builds it on the device and gives up (no dictionary) beyond kPatMaxPatterns
// distinct patterns or kPatMaxLen entries per row: unstructured meshes keep the plain stream.
constexpr int kPatMaxPatterns = 4096;
constexpr int kPatMaxLen = 32;
constexpr int kPatMaxDict = 2048; // npat * ml: the dictionary is copied into LDS by every workgroup (8 KiB)
struct PatDev {
    const unsigned short *id = nullptr; // [n] pattern of a row
    const int *off = nullptr;           // [npat * ml] col - row of the entries of a pattern, in row order
    int ml = 0;                         // stride of `off` (longest row)
    int npat = 0;
    // Round 5, row kinds: rows that repeat a pattern AND its values bit for bit (a constant-coefficient stencil, FEM with one
    // material on a structured mesh) share a "kind"; the product of such an operator streams no matrix at all -- a row is a
    // 16-bit kind id, the few kinds' offsets and values sit in LDS: 2 n + the vectors instead of 8 nnz + 22 n bytes.  The same
    // values times the same entries of x in the same order: the same sums bit for bit.  Built per factorize (the values
    // change under a kept pattern); absent (kind == nullptr) when the rows do not repeat, which is the usual FEM matrix.
    const unsigned short *kind = nullptr; // [n]
    const double *kval = nullptr;         // [nkind * kml], a kind's row padded with 0.0
    const int *koff = nullptr;            // [nkind * kml], ... and with offset 0
    const int *klen = nullptr;            // [nkind] entries of a row of this kind
    int nkind = 0;
    int kml = 0;                          // ml rounded up to a multiple of 8 (the kernel takes eight entries at a time)
    // ... and where all kinds together use at most kSlotMax distinct offsets, each kind's in ascending order (a 5- / 7-point
    // stencil with its boundary rows): the SLOT form.  Slot s stands for offset soff[s] (ascending); a kind is a dense
    // vector of kSlotMax coefficients and a presence mask.  Every lane gathers x at ALL slots' offsets whatever its kind --
    // the gathers no longer wait for the kind to arrive -- and adds the products of the slots its kind has, in slot order,
    // which is its row's entry order: the same sums.
    const double *scoef = nullptr;        // [nkind * kSlotMax]
    const unsigned *smask = nullptr;      // [nkind] bit s: the kind has slot s
    int soff[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // padded with 0 (the gather of a padded slot is x[row]; no kind has it)
    int nslot = 0;                        // > 0: the slot form is there
    int sdiag = -1;                       // slot of offset 0, or -1
};
constexpr int kSlotMax = 8;
constexpr int kKindTabMax = 1024; // kinds a per-kind table of the fused vector kernels holds in LDS (= pattern.hip's kKindMax)
constexpr int kKindMaxLdsBytes = 24 * 1024; // nkind * (12 kml + 4): the kinds are copied into LDS by every workgroup

struct CsrDev {
    int n = 0;        // local rows
    int n_ext = 0;    // local rows + halo columns (length of SpMV input vectors)
    int64_t nnz = 0;
    const int *rowptr = nullptr;
    const int *col = nullptr; // LOCAL column ids in [0, n_ext)
    const double *val = nullptr;
    const float *val32 = nullptr; // when set, the CSR products stream these single-precision copies of val
    int rows_per_block = 256; // SpMV row-block height (spmv_rows_per_block(nnz / n))
    const Bsr3Dev *bsr3 = nullptr; // when set, PLAIN / DOT / RESIDUAL products run on the block format
    const SellDev *sell = nullptr; // when set (and no row-block list is given), the products run on the SELL copy
    const PatDev *pat = nullptr;   // when set (one thread per row, no row-block list), the products skip the column stream
    // 16-bit columns (round 3): entry k of a row of row-block rb (col16_R rows each) sits in column
    // rb_base[8 rb + (col16[k] >> 13)] + (col16[k] & 8191) -- eight 8192-column windows per row-block, which is what the
    // row-blocks of an operator in a local numbering touch (a grid: the planes above and below; a breadth-first order: the
    // previous and the next level; a coarse AMG level).  The LDS-DMA kernel then streams 10 instead of 12 bytes per entry.
    const unsigned short *col16 = nullptr;
    const int *rb_base = nullptr;
    // round 4, wide-row operators on spmv_csr_dma: row-blocks of VARIABLE height -- block k covers the rows
    // [rb_start[k], rb_start[k + 1]): at most rows_per_block rows whose entries fit the LDS tile in ONE pass (a fixed height of
    // 64 rows left half of the row-blocks of a 31-entry-per-row operator with a few entries in a second pass)
    const int *rb_start = nullptr;
    int rb_count = 0, rb_R = 0, rb_tile = 0;
    int col16_R = 0;
};


int bsr3_brows_per_group(double avg_blocks_per_brow);

// The "lab.*" knobs of ONE handle (A/B runs and the tests that force a code path).  Round 6: members of the handle's Launch
// objects -- every Launch derived from the handle's (fit_launch: the AMG levels', the setup's) carries a copy -- where rounds
// 2-5 kept them in process-wide globals that psolve_hip_set_param on any handle wrote (instances are
// independent, like the reference's MAS handle with its private stream and pool, MASSolver.cu:186-196).
struct LabKnobs {
    int dma_tile_max = 2048;     // "lab.dma_tile_max": the largest LDS tile (entries) of spmv_csr_dma (= kDmaTile)
    int verbose = 0;             // "lab.verbose"
    int stage_kb = 256;          // "lab.stage_kb": host vectors up to this size cross PCIe through the pinned staging buffer (solve_host)
    int alternate = 0;           // "lab.alternate": bit 0 time_spmv alternates the sweep direction of consecutive launches, 1 nt y stores, 3 all-forward cycle sweeps
    int var_row_blocks = 1;      // "lab.var_row_blocks": wide-row operators of the AMG cycle get row-blocks packed to the tile (pack_row_blocks)
    int kind_unroll = 1;         // "lab.kind_unroll": rows per thread of spmv_csr_kind (1 / 2 / 4)
    int kind_probe = 0;          // "lab.kind_probe": measurement only (wrong results) -- 1 no gathers, 2 no store
    int kind_sched = -1;         // "lab.kind_sched": 0 the Launch's schedule (spmv_csr_pat's), 1 contiguous runs per workgroup, -1 the kernel's own
    int kind_slots = 1;          // "lab.kind_slots": 0 keeps spmv_csr_kind where the slot form exists
    int bsr3_kinds = 1;          // "lab.bsr3_kinds": 0 keeps the block stream where block-row kinds exist
    int agg_two_pass_assign = 1; // "lab.agg_two_pass_assign": the membership rule by two one-hop passes (0: round 4's two-hop walk)
    int cheb_split = -1;         // "lab.cheb_split": the block Chebyshev step as residual product + update launch: -1 where the operator is beyond the Infinity Cache (amg.hip: cheb_solve), 0 never, 1 always
    int symbolic_bitmap = 1;     // "lab.symbolic_bitmap": 0 keeps the hash tiers for every row of a symbolic product
};

struct Launch {
    hipStream_t stream = nullptr;
    LabKnobs lab;
    // Round 6: where set, the NEXT product / fused vector kernel launched with this Launch is started by hipExtLaunchKernelGGL
    // with these two events, which then carry the kernel's own begin and end timestamps (what rocprofv3's kernel trace reports)
    // -- hipEventRecord before and after a launch also times the dispatch gap (0.3136 against 0.2925 ms on the bench's product).
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int grid = 2048;      // persistent grid of the vector kernels (multiple of 8, <= kMaxPartials)
    int spmv_grid = 1280; // persistent grid of the SpMV (5 workgroups per CU: what its LDS admits)
    int spmv_xcd_map = 2; // 0 round-robin row-blocks, 1 contiguous eighth per XCD, 2 chunks dealt to XCDs
    int spmv_chunk_rows = 8192; // xcd_map 2: rows per chunk
    int spmv_kernel = -1; // 3: spmv_csr_pat (pattern dictionary, no column stream) where a dictionary exists, 2: SELL copy, 1: spmv_csr_dma (LDS-DMA staged stream, round 2), 0: spmv_csr_pipe (register staged, round 1), -1: dma for the operators streamed non-temporally
    int spmv_nt = -1;     // non-temporal matrix stream + y stores: -1 auto (operators above spmv_nt_bytes), 0 off, 1 on
    int64_t spmv_nt_bytes = 384ll << 20;
    bool vec_nt = false;  // the fused PCG vector kernels stream non-temporally too (set with the operator's verdict)
    int vec_policy = 7;   // which of their streams: bit 0 loads, 1 store of r, 2 store of x, 3 store of p
    int num_cus = 256;
    // Round 5, row kinds: a per-row vector that is constant within every kind of the operator's rows (the inverse diagonal
    // of Jacobi-PCG: rows of one kind have the same diagonal entry, bit for bit) is read as table[kind[row]] by the fused
    // vector kernels -- 2 bytes per row instead of 8, the same values: K2 26 n instead of 32 n, K3 42 n instead of 48 n.
    // kd_for: the vector the table was verified against (the kernels use the table only when handed that very vector).
    const unsigned short *kd_kind = nullptr;
    const double *kd_tab = nullptr, *kd_for = nullptr;
    int kd_n = 0;
    int bsr3_variant = -1; // spmv_bsr3_dma's gathers before the barrier: -1 by epilogue (the fused Chebyshev step only), 0 / 1 forces it off / on (lab)
};

int spmv_rows_per_block(double avg_nnz_per_row);
// host: greedy packing of consecutive rows into row-blocks of at most R rows and `tile_entries` stored entries (rowptr: host
// copy); starts gets count + 1 entries
void pack_row_blocks(int n, const int *rowptr_host, int R, int tile_entries, std::vector<int> &starts);
int spmv_dma_tile(const LabKnobs &lab, int R, double avg_nnz_per_row); // the LDS tile (entries) spmv_csr_dma takes for such an operator
// persistent-grid sizes fitted to a problem of n rows (row-block height R): small systems and coarse
// AMG levels get small grids, so that folding the per-workgroup partial sums stays negligible
// avg_nnz_per_row > 0: also raise the SpMV grid to what the operator's kernel admits per CU (wide rows: smaller LDS tiles)
Launch fit_launch(const Launch &max_cfg, int n, int rows_per_block, double avg_nnz_per_row = 0.0);
Launch fit_setup_launch(const Launch &max_cfg, int n, int64_t nnz, int rows_per_block); // ... for the setup kernels (grid by rows x lanes)

// the 16-bit column copy of an operator (see CsrDev::col16) for row-blocks of A.rows_per_block rows; false (and nothing to
// use) when some row-block touches more than eight 8192-column windows.  Synchronises the stream.
struct Col16 {
    DeviceBuffer<unsigned short> col;
    DeviceBuffer<int> base, flag;
    bool valid = false;
    bool build(const Launch &L, const CsrDev &A);
    void reset()
    {
        col.release();
        base.release();
        valid = false;
    }
    void attach(CsrDev &A) const
    {
        A.col16 = valid ? col.ptr : nullptr;
        A.rb_base = valid ? base.ptr : nullptr;
        A.col16_R = valid ? A.rows_per_block : 0;
    }
};

// SpMV epilogues (row-local work fused behind the row sum)
enum SpmvMode {
    SPMV_PLAIN = 0,    // y = A x
    SPMV_DOT = 1,      // y = A x ; partials[g] = sum_{rows of g} x[r] * y[r]
    SPMV_RESIDUAL = 2, // y = b - A x ; partials[g] = sum y[r]^2
    SPMV_ADD = 3,      // y += A x                                  (prolongation x += P u)
    SPMV_CHEB = 4,     // one Chebyshev step: res = dinv (b - A x); p = alpha res + beta p; y = x + p
    SPMV_POWER = 5     // s = dinv (A x); y = s; partials = sum s^2, partials2 = sum |s x|  (power iteration)
};

struct SpmvExtra {
    const double *dinv = nullptr;
    double *p = nullptr;
    double alpha = 0.0, beta = 0.0;
    double *partials2 = nullptr;
    // optional subset of row-blocks (distributed overlap: interior rows run while the halo travels)
    const int *rb_list = nullptr;
    int n_list = 0;
    int chunk = 1; // row-blocks per XCD chunk (filled by launch_spmv from Launch::spmv_chunk_rows)
    int gather4 = 1; // several threads per row: four gathers of a thread in flight (0: one at a time)
    const double *dinv_blk = nullptr; // SPMV_CHEB on a 3x3-block copy: inverted diagonal blocks (9 per node) instead of dinv
    int reverse = 0; // sweep the row-blocks from the last one (a product that follows one of the same operator finds its tail in the Infinity Cache)
};

// the instantiation the last launch_spmv of this thread chose, recorded while tl_spmv_kernel_record is set (kernels.hip)
extern thread_local char tl_spmv_kernel_name[160];
extern thread_local int tl_spmv_kernel_record;
extern thread_local char tl_vec_kernel_name[2][96];
void launch_spmv(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b, double *y,
                 double *partials, const int *done_flag, const SpmvExtra *extra = nullptr);
// whether launch_spmv would run `mode` on the operator's 3x3-block copy (the fused block epilogues: SPMV_ADD, and
// SPMV_CHEB with SpmvExtra::dinv_blk -- the Chebyshev step with block scaling in one launch)
bool bsr3_serves(const Bsr3Dev &B, SpmvMode mode, const Launch &L, const SpmvExtra &ex);
// Chebyshev step from x = 0 (no SpMV needed): p = alpha dinv b ; y = p
void launch_cheb_first(const Launch &L, int n, double alpha, const double *dinv, const double *b, double *p,
                       double *y);
// y = (float) x
void launch_to_f32(const Launch &L, int64_t n, const double *x, float *y);
// y[i] = a * x[i / bs]   (power-iteration start vector from the raw random stream, constant per block)
void launch_scale_expand(const Launch &L, int n, int bs, double a, const double *x, double *y);
// b0 = s / sqrt(sum(partials))   (power-iteration normalisation)
void launch_scale_by_norm(const Launch &L, int n, const double *partials, int np, const double *s, double *b0);

// partials[g] = sum a[i] * b[i] over g's slice (deterministic)
void launch_dot(const Launch &L, int n, const double *a, const double *b, double *partials);
// out[v] = sum of partials[v*stride .. +np) (one workgroup); host-visible dots, RCCL all-reduce staging
void launch_sum_partials(const Launch &L, const double *partials, int np, int stride, double *out, int nvec);
void launch_axpby(const Launch &L, int n, double a, const double *x, double b, double *y);
void launch_diag_inverse(const Launch &L, const CsrDev &A, double *invdiag, int *bad_count);
void launch_vmul(const Launch &L, int n, const double *d, const double *r, double *z); // z = d .* r (d may be null)
void launch_fill(const Launch &L, int n, double v, double *x);

// ---- block value types (block_size 3: AMGCL_Block<3>) ---------------------------------------------
// dinv_blk[node] = inverse of the b x b diagonal block of node `node` (identity if the block is absent)
void launch_block_diag_inverse(const Launch &L, const CsrDev &A, int bs, double *dinv_blk, int *bad_count);
void launch_block_diag_inverse_bsr(const Launch &L, int nb, int bs, const int *didx, const double *bval, double *dinv_blk,
                                   int *bad_count); // ... from a block copy (zero-filled blocks + diagonal positions)
// chebyshev update with block scaling: res = Dinv t; p = alpha res + beta p; x (+)= p
void launch_block_cheb_update(const Launch &L, int n, int bs, const double *dinv_blk, const double *t, double *p,
                              double *x, double alpha, double beta, bool x_is_zero);
// power-iteration step with block scaling: s = Dinv t (in place); partials = sum s^2; partials2 = sum_blocks |<s, b0>|
void launch_block_power(const Launch &L, int n, int bs, const double *dinv_blk, double *t, const double *b0,
                        double *partials, double *partials2);

// ---- AMG numeric setup on the device (pattern reuse: same sparsity, new values) --------------------
struct CsrMut { // CSR whose values are written by a kernel
    int n = 0;
    const int *rowptr = nullptr;
    const int *col = nullptr;
    double *val = nullptr;
};
// order-independent 64-bit hash of an int array (pattern identity check); *out += hash, out pre-zeroed
void launch_hash_i32(const Launch &L, int64_t n, const int *data, unsigned long long *out);
// *out += hash of the set {i : val[i] != 0} (which stored entries are nonzero); out pre-zeroed
void launch_hash_nonzero(const Launch &L, int64_t n, const double *val, unsigned long long *out);
// partials[g] = max over g's rows of (sum_j |a_ij|) / |a_ii|   (Gershgorin bound of rho(D^-1 A))
void launch_gershgorin(const Launch &L, const CsrDev &A, double *partials);
// P = (I - omega D_f^-1 A_f) P_tent for the aggregate map `id` (P's pattern given, values written);
// dia = diagonal of A for the strength test eps^2 a_ii a_jj < a_ij^2 (nullptr: eps_strong = 0)
void launch_prolongation_values(const Launch &L, const CsrDev &A, const int *id, double omega, const double *dia,
                                double eps_strong, CsrMut P);
// C = A * B for a C whose pattern (sorted columns) is already known
void launch_spgemm_numeric(const Launch &L, CsrMut C, const CsrDev &A, const CsrDev &B, double avg_c_row);

// ---- fused Jacobi/identity PCG steps (Eigen::internal::conjugate_gradient's recurrence) ----------
// init: p = M^-1 r ; partials_rz = r.p       (r and partials_rr come from SPMV_RESIDUAL)
void launch_pcg_init_dir(const Launch &L, int n, const double *invdiag, const double *r, double *p,
                         double *partials_rz);
// one workgroup: fold the three partial arrays into the state (thresholds, done[0], rz[0])
void launch_pcg_init_state(const Launch &L, PcgState *S, const double *part_rr, const double *part_bb,
                           const double *part_rz, int np_rr, int np_bb, int np_rz, double rel_tol, double abs_tol);
// K2: alpha = rz[par] / sum(part_pq) ; r -= alpha q ; partials of r.r and r.(M^-1 r)
void launch_pcg_update_r(const Launch &L, int n, int parity, const PcgState *S, const double *part_pq, int np_pq,
                         const double *invdiag, const double *q, double *r, double *part_rr, double *part_rz);
// K3: x += alpha p (always) ; latch convergence ; else beta = rz_new / rz_old, p = M^-1 r + beta p
void launch_pcg_update_xp(const Launch &L, int n, int parity, PcgState *S, const double *part_pq, int np_pq,
                          const double *part_rr, const double *part_rz, int np_rr, const double *invdiag,
                          const double *r, double *p, double *x, int max_iter);

// ---- generic-preconditioner PCG steps (z comes from an arbitrary M^-1, e.g. the AMG V-cycle) ------
// x += alpha p ; r -= alpha q ; partial r.r
void launch_pcg_update_xr(const Launch &L, int n, int parity, const PcgState *S, const double *part_pq, int np_pq,
                          const double *p, const double *q, double *x, double *r, double *part_rr);
// latch convergence from part_rr (one workgroup)
void launch_pcg_check(const Launch &L, int parity, PcgState *S, const double *part_rr, int np_rr, int max_iter);
// beta = (r.z) / rz_old ; p = z + beta p ; stores rz_new
void launch_pcg_update_p(const Launch &L, int n, int parity, PcgState *S, const double *part_rz, int np_rz,
                         const double *z, double *p);

// ---- single-reduction PCG for shards (Chronopoulos-Gear recurrences; one all-reduce per iteration) ------
// mode 0 regular / 1 first iteration / 2 check only; red3 = reduced (r.u, r.r, w.u) of the current residual
void launch_cg1_update(const Launch &L, int n, int parity, int mode, PcgState *S, const double *red3,
                       const double *invdiag, double *u, const double *w, double *p, double *s, double *x, double *r,
                       double *part_g, double *part_rr);
void launch_cg1_fold(const Launch &L, const double *part_g, const double *part_rr, int np, const double *part_d,
                     int np_d, double *out3);

// ---- synthetic inputs ------------------------------------------------------------------------------
void launch_poisson7_generate(const Launch &L, int nx, int ny, int nz, int z0, int z1, int *rowptr, int *col,
                              double *val);
int64_t poisson7_nnz_before(int nx, int ny, int nz, int64_t row); // closed form, host
void launch_splitmix(const Launch &L, int n, uint64_t seed, int64_t start, double *x);
// x[i] = U(-1,1) from SplitMix64(seed + idx[i])
void launch_splitmix_indexed(const Launch &L, int n, uint64_t seed, const int *idx, double *x);

// ---- distributed helpers -----------------------------------------------------------------------------
void launch_gather(const Launch &L, int n, const int *idx, const double *x, double *out); // out[i] = x[idx[i]]
// count / collect global column ids outside [row0, row1)
void launch_offrange_count(const Launch &L, int64_t nnz, const int *col, int row0, int row1, int *count);
void launch_offrange_collect(const Launch &L, int64_t nnz, const int *col, int row0, int row1, int *out, int *cursor);
// flags[rb] = 1 if any row of row-block rb (R rows each) has a column id >= n_local (i.e. needs the halo)
void launch_classify_row_blocks(const Launch &L, const CsrDev &A, int *flags);
// col := col - row0 if local, else n_local + lower_bound(halo, col)
void launch_remap_cols(const Launch &L, int64_t nnz, int *col, int row0, int row1, int n_local, const int *halo,
                       int n_halo);

// local column ids of a shard back to global ids: out[i] = col[i] + row0 (local) or halo[col[i] - n_local]
void launch_unmap_cols(const Launch &L, int64_t nnz, const int *col, int row0, int n_local, const int *halo, int *out);
void launch_add_offset_i32(const Launch &L, int64_t n, int *v, int offset);

} // namespace psolve
