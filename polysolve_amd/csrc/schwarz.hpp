// schwarz.hpp -- multilevel additive Schwarz preconditioner on wave64-sized dense blocks.
//
// the reference's in-tree GPU preconditioner is MAS ("multilevel additive Schwarz",
// /root/reference/src/polysolve/linear/mas_utils/MASPreconditioner.cu:58-457): nodes grouped in warp-sized
// (BANK_SIZE = 32) domains, coarser levels obtained by merging nodes inside a domain with __ballot_sync masks,
// a dense matrix per domain and level assembled from the entries that fall inside it, inverted once, and
// z = sum over levels of P_l B_l^-1 P_l^T r applied by a packed symmetric product (:661-664 "vram bandwidth
// bound").  This is that method re-thought for CDNA4's 64-wide wavefronts, not a translation of it:
//   * a domain is 64 consecutive unknowns (one wavefront, one 64 x 64 fp64 block = 32 KiB);
//   * level l+1 has one unknown per level-l domain (piecewise-constant transfer: index >> 6), so a level is
//     reached from the fine index by shifts -- no connectivity masks, no ballots;
//   * B_l = the 64 x 64 diagonal blocks of P_l^T A P_l, summed DETERMINISTICALLY (one wave per coarse
//     unknown walks its fine rows in order; lane J keeps the sum for column J) -- no atomics;
//   * each block is inverted in LDS by one wavefront (Gauss-Jordan, SPD: no pivoting);
//   * apply: r restricted by wave sums, coarse levels first, then one streaming pass over the level-0 blocks.
#pragma once
#include <memory>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

class Context;

class SchwarzPrecond {
public:
    static constexpr int kDomain = 64;
    // levels: 1..4 (level l has ceil(n / 64^l) unknowns; coarser levels than the matrix has rows are dropped)
    // block_size > 1 (unknowns interleaved per node): coarse unknowns are per component (node group, c)
    void setup(Context &ctx, const CsrDev &A, int levels, int block_size);
    // z = sum_l P_l B_l^-1 P_l^T r; done_flag (device, optional): set -> the kernels return at once
    void apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag = nullptr);
    int levels() const { return (int)lv_.size(); }
    int rows() const { return n_; }

private:
    struct Level {
        int n = 0, nblk = 0;
        DeviceBuffer<double> inv; // nblk x 64 x 64, row-major, symmetric
        DeviceBuffer<double> r, z; // restricted residual / correction of this level (levels > 0)
    };
    std::vector<std::unique_ptr<Level>> lv_;
    int n_ = 0, bs_ = 1;
};

} // namespace psolve
