// amg.hpp -- Chebyshev-smoothed aggregation AMG V-cycle preconditioner (device apply).
// Reference configuration it mirrors: /root/reference/src/polysolve/linear/AMGCL.cpp:32-65.
#pragma once
#include <memory>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

struct AmgParams;
class Context;

class AmgHierarchy {
public:
    AmgHierarchy();
    ~AmgHierarchy();
    // A: factorized fine-level matrix on the device (local column ids, single GPU)
    void setup(Context &ctx, const CsrDev &A, const AmgParams &prm);
    // shards ("amg.dist_global"): Aglobal is the WHOLE matrix, gathered on every rank; the hierarchy is the global one,
    // this rank applies level 0 on its rows [row0, row0 + n_loc) (ctx.A: the shard with its halo) and levels >= 1
    // replicated.  With fewer than two levels nothing is split (global_on_shards() stays false).
    void setup_global(Context &ctx, const CsrDev &Aglobal, int row0, int n_loc, const AmgParams &prm);
    bool global_on_shards() const;
    // z = M^-1 r  (x = 0; one cycle -- amgcl::amg::apply).  done_flag (device, optional): when set the
    // products of the cycle return at once (iterations queued behind the converged one)
    void apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag = nullptr);
    int levels() const;
    bool last_setup_reused() const;
    int levels_aggregated_on_device() const; // of the last full setup
    int operators_with_packed_row_blocks() const; // A_l / R_l whose row-blocks are packed to the LDS tile (DevCsr::set_row_blocks)
    void level_shape(int l, int64_t *rows, int64_t *nnz, double *rho) const;
    // what: 0 = A_l, 1 = P_l, 2 = R_l; out = {rows, cols, nnz}; copy = D2H of the three CSR arrays
    void level_matrix_shape(int l, int what, int64_t out[3]) const;
    void level_matrix_copy(hipStream_t s, int l, int what, int *rowptr, int *col, double *val) const;
    // perm[i] = row of level l's operator that row i of the setup's (AMGCL's) numbering became ("amg.renumber");
    // the identity where the level kept its numbering.  Returns whether the level was renumbered.
    bool level_perm_copy(hipStream_t s, int l, int *perm) const;
    // HIP-event times (us per launch, mean of `reps`) of the cycle's operations on level l, on the hierarchy's own operators:
    // out[0] one Chebyshev step (product + fused update), [1] residual, [2] restriction to level l + 1, [3] prolongation
    // from it, [4] the first Chebyshev step from x = 0 (no product); [2] / [3] are 0 on the coarsest level
    void time_level_ops(Context &ctx, int l, int reps, double out_us[5]);

    struct Impl;
    std::unique_ptr<Impl> impl;
};

} // namespace psolve
