// amg_dist.hpp -- smoothed-aggregation AMG whose hierarchy is built and applied on the shards of a row-partitioned
// matrix ("fine levels partitioned the same way (aggregates never cross partitions ...), coarse levels
// below ~50 k rows/GPU are all-gathered and solved redundantly on every GPU").
//
// The reference has no counterpart: its only "distributed" preconditioner runs on one rank
// (/root/reference/src/polysolve/linear/HypreSolver.cpp:96-102), and AMGCL's MPI backend is not what
// /root/reference/src/polysolve/linear/AMGCL.cpp:32-65 instantiates.  What is kept from the reference is the method --
// AMGCL's plain aggregation (run per shard on the strong connections INSIDE the shard), its smoothed prolongation
// P = (I - omega D^-1 A_F) P_tent with the global Gershgorin omega, R = P^T, the Galerkin product R A P, the Chebyshev
// smoother and the cycle (amg.hpp) -- and what is new is where the numbers live:
//   * level l: rows [offsets_l[rank], offsets_l[rank + 1]) of A_l on this device, columns = local rows + halo;
//   * aggregates confined to the shard, numbered rank after rank (an all-gather of the counts);
//   * P_l rows local; its halo rows fetched from their owners (variable-length row exchange) for the local product
//     A_l P_l; R_l = the transpose of [P_l local ; P_l halo rows] restricted to the coarse nodes this rank owns;
//     A_{l+1} = R_l (A_l P_l) with the halo rows of A_l P_l fetched the same way -- no rank ever holds more than its
//     rows plus one ring of halo rows of any operator;
//   * a level whose global row count falls under "amg.dist_replicate_rows" x ranks is gathered to every rank and the
//     rest of the hierarchy is the single-device one (amg.hpp), replicated;
//   * the cycle exchanges halos (neighbour to neighbour) before each product; the only collectives are the all-gather
//     of the first replicated level's right-hand side and the dot products of PCG itself.
#pragma once
#include <memory>
#include <vector>

#include "amg.hpp"
#include "amg_symbolic.hpp"
#include "common.hpp"
#include "dist.hpp"
#include "kernels.hpp"

namespace psolve {

struct AmgParams;
class Context;

// halo of a row-partitioned column space: which off-rank entries this rank reads, which of its own it sends
struct HaloLink {
    HaloPlan plan;               // plan.row_offsets: the partition of the column space
    DeviceBuffer<int> halo_dev;  // sorted global ids of the halo columns
    DeviceBuffer<int> send_idx;  // local ids of the entries this rank sends, grouped by destination rank
    DeviceBuffer<double> send_buf;
    DeviceBuffer<int> send_buf_i;
    int n_local = 0;
    int n_halo() const { return (int)plan.halo.size(); }
};

class DistAmg {
public:
    DistAmg();
    ~DistAmg();
    // ctx: the factorized shard (ctx.A with its halo, the communicator).  Collective: every rank calls it.
    void setup(Context &ctx, const AmgParams &prm);
    // z = M^-1 r on the shard's rows (x = 0, one cycle)
    void apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag);
    int levels() const;             // distributed levels + levels of the replicated tail
    int distributed_levels() const; // levels whose rows are partitioned
    bool last_setup_reused() const; // the last setup() kept the patterns and recomputed the numbers
    void level_shape(int l, int64_t *rows_global, int64_t *rows_local, int64_t *nnz_local, double *rho) const;

    struct Impl;
    std::unique_ptr<Impl> impl;
};

} // namespace psolve
