// amg_symbolic.hpp -- device-side symbolic half of the smoothed-aggregation setup (SURVEY.md 8(f)
// "device-side AMG setup").  Everything amgcl::coarsening::smoothed_aggregation needs besides the
// greedy aggregation sweep itself (which is sequential by definition, amgcl/coarsening/plain_aggregates.hpp,
// and stays on the host working on a compacted strong-connection graph):
//   * strength-of-connection graph  (strong off-diagonals + the stored diagonal, sorted)
//   * pattern of P  = pattern(S) x aggregate map,  R = P^T with the entry map,
//   * patterns of A P and R (A P)  (row-wise hash sets in LDS, sorted by a bitonic network),
// all with sorted columns, i.e. exactly the patterns the host Gustavson product (amg_setup.cpp) builds,
// so that the numeric kernels (kernels.hip: prolongation_values, spgemm_numeric) produce the same
// hierarchy bit for bit.
#pragma once
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace psolve {



struct SymbolicScratch {
    DeviceBuffer<int> cand;          // per-row candidate bound / counts
    DeviceBuffer<unsigned char> tier;
    DeviceBuffer<int> counters;      // small device counters
    DeviceBuffer<int> table;         // global hash tables of the rows too wide for LDS
    DeviceBuffer<long long> bsum;    // scan block sums
    DeviceBuffer<int> tmp;           // transpose staging
    DeviceBuffer<int> cursor;
    PinnedBuffer<long long> host;    // small D2H results
};

// data[0..n) = counts on entry; data[0..n] = exclusive prefix sums on exit (data[n] = total).  Returns the
// total (synchronises the stream); throws PSOLVE_HIP_ERANGE when it does not fit int32.
int64_t device_exclusive_scan(const Launch &L, int *data, int64_t n, SymbolicScratch &S);

// the square diagonal block of a shard: rows of A with the halo columns (>= A.n) dropped; returns its nnz
int64_t device_diagonal_block(const Launch &L, const CsrDev &A, DeviceBuffer<int> &ptr, DeviceBuffer<int> &col,
                              DeviceBuffer<double> &val, SymbolicScratch &S);

// dia[i] = a_ii (0 if not stored)
void launch_extract_diagonal(const Launch &L, const CsrDev &A, double *dia);

// strong connections of plain_aggregates (eps^2 a_ii a_jj < a_ij^2, i != j) plus the stored diagonal,
// columns in row order.  sptr/scol are (re)allocated; returns nnz of the graph.  id0[i] = -1 (undefined) for
// rows with a strong connection, -2 (removed) for the others: the start state of the aggregation sweep.
int64_t device_strength_graph(const Launch &L, const CsrDev &A, double eps_strong, const double *dia,
                              DeviceBuffer<int> &sptr, DeviceBuffer<int> &scol, int *id0, SymbolicScratch &S);

// pattern of C = A * B, sorted columns.  B is CSR (bptr, bcol) or, with bptr == nullptr, a map:
// row c of B is {bcol[c]} when bcol[c] >= 0 and empty otherwise (the tentative prolongation).  With bptr ==
// bcol == nullptr row c of B is {c / div}: scalar columns folded onto block columns (the block graph).
int64_t device_spgemm_symbolic(const Launch &L, int n, const int *aptr, const int *acol, const int *bptr,
                               const int *bcol, int ncols_c, DeviceBuffer<int> &cptr, DeviceBuffer<int> &ccol,
                               SymbolicScratch &S, int div = 1);

// R = P^T (pattern, sorted columns) and r_from_p with R.val[k] = P.val[r_from_p[k]]
void device_transpose_pattern(const Launch &L, int n, int ncols, const int *pptr, const int *pcol, int64_t nnz,
                              DeviceBuffer<int> &rptr, DeviceBuffer<int> &rcol, DeviceBuffer<int> &r_from_p,
                              SymbolicScratch &S);

// ---- numeric products on 3 x 3 blocks (amg_bspgemm.hip) ---------------------------------------------------
// C = A B on block patterns, C's values in the expanded scalar layout (expand_block_csr); amap_transposed != nullptr:
// block p of A is the transpose of block amap[p] of aval; b_expanded: B's values in the expanded layout of ITS pattern
void launch_bspgemm3_numeric(const Launch &L, int nbr, const int *cptr, const int *ccol, double *cval_expanded, const int *aptr,
                             const int *acol, const double *aval, const int *amap_transposed, const int *bptr,
                             const int *bcol, const double *bval, bool b_expanded, double avg_c_blocks = 0.0);

// ---- amgcl's other runtime classes (amg_relax.hip, round 5) -------------------------------------------------
struct BlockGraph;
constexpr int kDirectCoarseMaxRows = 4096; // "amg.direct_coarse": the coarsest operator is inverted densely (n^2 doubles, n launches)
// the diagonal scaling M of a one-step relaxation x <- x + M (rhs - A x): type 1 damped_jacobi (damping / a_ii), 2 spai0
// (a_ii / sum_j a_ij^2), 3 the identity (chebyshev.scale = false).  bad += entries that are not finite
void launch_relax_scaling(const Launch &L, const CsrDev &A, int type, double damping, double *m, int *bad);
// the same on the blocks of G; type 1 expects the inverted diagonal blocks in m on entry and scales them
void launch_block_relax_scaling(const Launch &L, const BlockGraph &G, int type, double damping, double *m, int *bad);
// tentative prolongation of amgcl's aggregation coarsening: row i -> {id[i]} when id[i] >= 0 (one entry per kept row; val,
// when given, gets the b x b identity per entry).  Returns the entry count
int64_t device_tentative_prolongation(const Launch &L, int n_nodes, const int *id, int b, DeviceBuffer<int> &ptr,
                                      DeviceBuffer<int> &col, DeviceBuffer<double> *val, SymbolicScratch &S);
void launch_scale_values(const Launch &L, int64_t n, double s, double *v); // v = s * v
// inv = A^-1, dense n x n row-major, by Gauss-Jordan without pivoting (A SPD, n <= kDirectCoarseMaxRows); synchronises
void device_dense_inverse(const Launch &L, const CsrDev &A, DeviceBuffer<double> &inv, DeviceBuffer<double> &work);
// one smoothing step of the coarsest level run on the identity (n x n iterates, row-major; see amg_relax.hip): xout = xin + p,
// p = alpha M (I - A xin) + beta p; first: xin = 0
void launch_dense_smoother_step(const Launch &L, const CsrDev &A, int bs, const double *m, const double *mblk, const double *xin,
                                double *pm, double *xout, double alpha, double beta, bool first);
void launch_dense_matvec(const Launch &L, int n, const double *ainv, const double *x, double *y, const int *done_flag);

// ---- amgcl's ordered relaxations: gauss_seidel, ilu0 (amg_sweep.hip, round 6) ------------------------------
// the operator of a sweep: b x b blocks, row-major inside a block (b = 1: the CSR arrays themselves), rows sorted by column
struct SweepView {
    int nb = 0, b = 1;
    int64_t nnzb = 0;
    const int *ptr = nullptr, *col = nullptr;
    const double *val = nullptr;
};
// one sweep in row order (mode 0 / 1: gauss_seidel forward / backward with new values from `out`, old ones from `old`; 2 / 3:
// forward / backward substitution with ilu0's factors), out = the swept vector; ctrl: 4 ints of scratch
void launch_sweep(const Launch &L, const SweepView &A, int mode, const double *dinv, const double *in, const double *old, double *out,
                  int *ctrl, const int *done);
// lu = ilu0's factors on A's pattern (strictly lower part: the multipliers, the rest: the eliminated rows), dinv = the inverted
// pivots; synchronises, throws on a missing diagonal / unsorted rows
void device_ilu0_factor(const Launch &L, const SweepView &A, DeviceBuffer<double> &work, DeviceBuffer<double> &lu, double *dinv,
                        int *ctrl);

// ---- locality renumbering of the coarse levels (amg_renumber.hip) -----------------------------------------
// new_of_old[i] = position of node i when the nodes are ordered by (new id of their aggregate, old id): key =
// parent_new[id[i]] (parent_new == nullptr: id[i] itself); nodes with id < 0 go last.  w_*: scratch.
void device_order_by_parent(const Launch &L, int n, const int *id, const int *parent_new, int n_parent,
                            DeviceBuffer<int> &new_of_old, SymbolicScratch &S, DeviceBuffer<int> &w_key,
                            DeviceBuffer<int> &w_iota, DeviceBuffer<int> &w_ptr, DeviceBuffer<int> &w_order,
                            DeviceBuffer<int> &w_map);
// out = the CSR matrix with row i moved to row_new[i] and column c renamed col_new[c] (either may be nullptr: identity),
// columns sorted inside every row; val / oval may be nullptr (pattern only)
void device_permute_csr(const Launch &L, int n, int64_t nnz, const int *ptr, const int *col, const double *val,
                        const int *row_new, const int *col_new, DeviceBuffer<int> &optr, DeviceBuffer<int> &ocol,
                        DeviceBuffer<double> *oval, SymbolicScratch &S, DeviceBuffer<int> *omap = nullptr);
void launch_iota(const Launch &L, int n, int *out);
// out[new_of_old ? new_of_old[i] : i] = id[i] < 0 || !value_map ? id[i] : value_map[id[i]]
void launch_relabel_ids(const Launch &L, int n, const int *id, const int *new_of_old, const int *value_map, int *out);
void launch_permute_f64(const Launch &L, int n, const double *in, const int *new_of_old, double *out); // out[new_of_old[i]] = in[i]

// ---- aggregation (amg_aggregate.hip) -------------------------------------------------------------------
struct AggregateScratch {
    DeviceBuffer<int> ints;
    DeviceBuffer<int> tptr, tcol, tmap; // transposed strength graph (unsymmetric / unsorted patterns only)
};
// The greedy sweep of plain_aggregates on the device, same result as the sequential loop (see the file
// header).  graph = strong connections + diagonal (device_strength_graph), id0 = the sweep's start state.
// Returns the aggregate count and fills id[n]; -1 = more than max_rounds dependency rounds (the caller falls
// back to the host sweep).
// mode 1: dependency rounds (two kernels per round); mode 2: no rounds, every vertex waits for the earlier vertices
// it depends on inside one kernel (max_rounds then bounds the time: 10 us per round); mode 3: "amg.aggregation" = "parallel" --
// NOT the sweep's seeds but the distance-2 maximal independent set by hashed priorities (defined on the graph as given, symmetric or not).
int64_t device_aggregate(const Launch &L, int n, const int *sptr, const int *scol, const int *id0, int *id,
                         int max_rounds, AggregateScratch &W, SymbolicScratch &S, int *rounds_out, int mode = 2);

// ---- block value types (amg_block.hip) -----------------------------------------------------------------
// b x b block view of a scalar CSR operator: sorted block columns, zero-filled row-major blocks
struct BlockGraph {
    int nb = 0, b = 1;
    int64_t nnzb = 0;
    DeviceBuffer<int> rowstart; // rowptr[i * b]: the scalar entries of block row i are one contiguous run
    DeviceBuffer<int> ptr, col;
    DeviceBuffer<double> val;          // nnzb * b * b
    DeviceBuffer<int> didx;            // position of the diagonal block of every block row (-1: absent)
    DeviceBuffer<unsigned char> strong; // strength flag of every block
    mutable DeviceBuffer<double> per_block; // scratch of the two-phase kernels (round 6): one number per block
};
// pattern of the block graph (returns nnzb); then its values + diagonal positions
int64_t device_block_graph(const Launch &L, const CsrDev &A, int b, BlockGraph &G, SymbolicScratch &S);
void device_block_values(const Launch &L, const CsrDev &A, BlockGraph &G);
// strength flags only (flags[nnzb], cnt[nb] scratch): eps^2 tr(D_i D_j) < tr(A_ij A_ij), i != j
void device_block_strong_flags(const Launch &L, const BlockGraph &G, double eps_strong, unsigned char *flags, int *cnt);
// flags into G.strong + the compacted graph (strong blocks + the diagonal) and the sweep's start state
int64_t device_block_strength_graph(const Launch &L, BlockGraph &G, double eps_strong, DeviceBuffer<int> &sptr,
                                    DeviceBuffer<int> &scol, int *id0, SymbolicScratch &S);
int device_block_flag_changes(const Launch &L, int64_t n, const unsigned char *a, const unsigned char *b,
                              SymbolicScratch &S);
// max_i (sum_j ||A_ij||_F) ||D_i^-1||_F   (synchronises)
// rows_list (n_list entries): the bound over these block rows only (3x3 blocks: the representatives of block-row kinds)
double device_block_gershgorin(const Launch &L, const BlockGraph &G, double *partials, const int *rows_list = nullptr, int n_list = 0);
// block values of P = (I - omega D_f^-1 A_f) P_tent on the block pattern (pbptr, pbcol)
void launch_block_prolongation_values(const Launch &L, const BlockGraph &G, const int *id, double omega,
                                      const int *pbptr, const int *pbcol, double *pbval);
// block CSR -> scalar CSR with full blocks; ptr / col may be nullptr (values only, for the numeric refresh)
void launch_expand_block_csr(const Launch &L, int nb, int b, const int *pbptr, const int *pbcol, const double *pbval,
                             int *ptr, int *col, double *val);
// pattern only: the rows as they are, (block) column c -> scalar columns c b .. c b + b - 1
void launch_expand_block_columns(const Launch &L, int n, int b, const int *fptr, const int *fcol, int *ptr, int *col);

} // namespace psolve
