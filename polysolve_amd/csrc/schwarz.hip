// schwarz.hip -- see schwarz.hpp.
#include "schwarz.hpp"

#include "solver.hpp"

namespace psolve {

namespace {

constexpr int D = SchwarzPrecond::kDomain; // 64

__device__ __forceinline__ double wave_sum64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(v));
        int hi = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// Index maps.  With block_size bs > 1 (vector problems: bs unknowns per node, interleaved) the coarse levels keep the
// components apart, as MAS does with its 3 x 3 node blocks: the level-l unknown of fine unknown i = node * bs + c is
// (node >> 6 l) * bs + c -- a constant per component over 64^l nodes, not a constant over mixed components.
__device__ __forceinline__ int coarse_of(int i, int shift, int bs)
{
    if (bs == 1) return i >> shift;
    const int node = i / bs;
    return (node >> shift) * bs + (i - node * bs);
}

// B_l: one wave per level-l unknown I = (g, c).  It walks its fine rows -- nodes g * 64^l .. (g+1) * 64^l - 1,
// component c -- in order, 64 stored entries at a time (coalesced), and lane J accumulates the entries whose
// level-l column lies in I's domain at position J -- a fixed order of additions, no atomics.  Output row I of
// block I >> 6.
__global__ __launch_bounds__(256) void schwarz_assemble_kernel(int n, const int *__restrict__ rowptr,
                                                                const int *__restrict__ col,
                                                                const double *__restrict__ val, int shift, int bs,
                                                                int n_l, double *__restrict__ B)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const int nnodes = n / bs;
    for (int I = wave; I < n_l; I += nwaves) {
        const int g = I / bs, c = I - g * bs;
        const int64_t node0 = (int64_t)g << shift;
        const int64_t node1 = min((int64_t)nnodes, ((int64_t)g + 1) << shift);
        const int dom = I >> 6;
        double acc = 0.0;
        // bs == 1: the rows are consecutive, their entries one contiguous run; bs > 1: row by row (stride bs)
        const int64_t nruns = bs == 1 ? 1 : node1 - node0;
        for (int64_t run = 0; run < nruns; ++run) {
            const int64_t r0 = bs == 1 ? node0 : (node0 + run) * bs + c;
            const int64_t r1 = bs == 1 ? node1 : r0 + 1;
            const int k0 = __builtin_amdgcn_readfirstlane(rowptr[r0]), k1 = __builtin_amdgcn_readfirstlane(rowptr[r1]);
            for (int k = k0; k < k1; k += 64) {
                const int kk = k + lane;
                int cc = -1;
                double v = 0.0;
                if (kk < k1) {
                    cc = col[kk];
                    v = val[kk];
                }
                // level-l column of my entry, or -1 when it is a halo column / outside I's domain
                int J = -1;
                if (cc >= 0 && cc < n) {
                    const int cl = coarse_of(cc, shift, bs);
                    if ((cl >> 6) == dom) J = cl & 63;
                }
                const int m = min(64, k1 - k);
                for (int e = 0; e < m; ++e) { // entries in storage order; lane J takes the ones of its column
                    const int Je = __builtin_amdgcn_readlane(J, e);
                    if (Je < 0) continue;
                    const int vlo = __builtin_amdgcn_readlane(__double2loint(v), e);
                    const int vhi = __builtin_amdgcn_readlane(__double2hiint(v), e);
                    if (lane == Je) acc += __hiloint2double(vhi, vlo);
                }
            }
        }
        B[(int64_t)dom * (D * D) + (int64_t)(I & 63) * D + lane] = acc;
    }
}

// in-place inverse of every 64 x 64 block (one wave per block, Gauss-Jordan in LDS, no pivoting: SPD).
// Rows of unknowns beyond n_l (the last block's padding) become identity rows.  A pivot that is zero or negligible
// against the block's largest diagonal entry (a floating sub-domain: the block is only semi-definite; an empty row)
// takes its unknown OUT of the domain solve: row and column k are replaced by the identity's before the step, so that
// what is inverted is the (still SPD) block without that unknown -- not a row scaled by 1 with its couplings kept,
// which made M^-1 indefinite without a word (round-2 advice).
__global__ __launch_bounds__(64) void schwarz_invert_kernel(int nblk, int n_l, double *__restrict__ B, int *bad)
{
    __shared__ double M[D][D + 1];
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        double *Bb = B + (int64_t)b * (D * D);
        for (int i = 0; i < D; ++i) {
            const bool pad = b * D + i >= n_l || b * D + lane >= n_l;
            M[i][lane] = pad ? (i == lane ? 1.0 : 0.0) : Bb[i * D + lane];
        }
        __syncthreads();
        double dmax = fabs(M[lane][lane]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int lo = __builtin_amdgcn_ds_bpermute((lane ^ off) << 2, __double2loint(dmax));
            const int hi = __builtin_amdgcn_ds_bpermute((lane ^ off) << 2, __double2hiint(dmax));
            dmax = fmax(dmax, __hiloint2double(hi, lo));
        }
        const double tiny = 64.0 * 2.220446049250313e-16 * dmax;
        for (int k = 0; k < D; ++k) {
            double piv = M[k][k];
            if (!(fabs(piv) > tiny) || !isfinite(piv)) { // (wave-uniform: every lane reads the same M[k][k])
                if (lane == 0 && isfinite(piv) == false) atomicAdd(bad, 1);
                __syncthreads();
                M[k][lane] = (lane == k) ? 1.0 : 0.0;
                M[lane][k] = (lane == k) ? 1.0 : 0.0;
                __syncthreads();
                piv = 1.0;
            }
            const double ip = 1.0 / piv;
            __syncthreads();
            // row k: scaled; its own column holds the inverse pivot
            const double mk = (lane == k) ? ip : M[k][lane] * ip;
            __syncthreads();
            M[k][lane] = mk;
            __syncthreads();
            for (int i = 0; i < D; ++i) {
                if (i == k) continue;
                const double f = M[i][k];
                __syncthreads();
                M[i][lane] = (lane == k) ? -f * ip : M[i][lane] - f * mk;
                __syncthreads();
            }
        }
        for (int i = 0; i < D; ++i) Bb[i * D + lane] = M[i][lane];
        __syncthreads();
    }
}

// r_c[(g, c)] = sum over the 64 children (64 g + k, c) of r_f  (one wave per coarse unknown; fixed butterfly order)
__global__ __launch_bounds__(256) void schwarz_restrict_kernel(int n_f, const double *__restrict__ rf, int n_c,
                                                                double *__restrict__ rc, int bs, const int *done)
{
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int I = wave; I < n_c; I += nwaves) {
        const int g = I / bs, c = I - g * bs;
        const int64_t i = ((int64_t)g * D + lane) * bs + c;
        const double s = wave_sum64(i < n_f ? rf[i] : 0.0);
        if (lane == 0) rc[I] = s;
    }
}

// z[64 b + i] = sum_j Binv_b[j][i] r[64 b + j]  (+ z_coarse[b]: the next level's correction, injected);
// one wave per block, the block streamed row by row (512 B coalesced per row)
__global__ __launch_bounds__(256) void schwarz_block_apply_kernel(int n_l, int nblk, const double *__restrict__ Binv,
                                                                   const double *__restrict__ r,
                                                                   const double *__restrict__ zc,
                                                                   double *__restrict__ z, int bs, const int *done)
{
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int b = wave; b < nblk; b += nwaves) {
        const int64_t i = (int64_t)b * D + lane;
        const double ri = i < n_l ? r[i] : 0.0;
        const double *Bb = Binv + (int64_t)b * (D * D);
        double acc = 0.0;
#pragma unroll 8
        for (int j = 0; j < D; ++j) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(ri), j);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(ri), j);
            acc += __builtin_nontemporal_load(Bb + j * D + lane) * __hiloint2double(hi, lo);
        }
        if (zc && i < n_l) acc += zc[coarse_of((int)i, 6, bs)];
        if (i < n_l) z[i] = acc;
    }
}

} // namespace

void SchwarzPrecond::setup(Context &ctx, const CsrDev &A, int levels, int block_size)
{
    bs_ = (block_size > 1 && A.n % block_size == 0) ? block_size : 1;
    PS_REQUIRE(levels >= 1 && levels <= 4, PSOLVE_HIP_EINVAL, "schwarz.levels must be 1..4");
    hipStream_t s = ctx.stream;
    const Launch L = ctx.launch_config();
    n_ = A.n;
    lv_.clear();
    DeviceBuffer<int> bad;
    bad.ensure(1);
    PS_HIP_CHECK(hipMemsetAsync(bad.ptr, 0, sizeof(int), s));
    int64_t n_l = A.n;
    for (int l = 0; l < levels; ++l) {
        std::unique_ptr<Level> lv(new Level());
        lv->n = (int)n_l;
        lv->nblk = (int)((n_l + D - 1) / D);
        lv->inv.ensure((size_t)lv->nblk * D * D);
        if (l > 0) {
            lv->r.ensure((size_t)lv->n + 1);
            lv->z.ensure((size_t)lv->n + 1);
        }
        // blocks of the last domain are only partly written by the assembly: clear first
        PS_HIP_CHECK(hipMemsetAsync(lv->inv.ptr + (size_t)(lv->nblk - 1) * D * D, 0, (size_t)D * D * sizeof(double), s));
        const int grid = (int)std::min<int64_t>(L.grid, std::max<int64_t>(1, (n_l + 3) / 4));
        hipLaunchKernelGGL(schwarz_assemble_kernel, dim3(grid), dim3(256), 0, s, A.n, A.rowptr, A.col, A.val, 6 * l, bs_, lv->n,
                           lv->inv.ptr);
        PS_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(schwarz_invert_kernel, dim3(std::min(lv->nblk, 256 * 16)), dim3(64), 0, s, lv->nblk, lv->n,
                           lv->inv.ptr, bad.ptr);
        PS_HIP_CHECK(hipGetLastError());
        lv_.push_back(std::move(lv));
        if (n_l <= D) break; // one domain covers the level: nothing coarser to add
        n_l = ((n_l / bs_ + D - 1) / D) * bs_; // one unknown per component and group of 64 nodes
    }
    int nbad = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&nbad, bad.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    PS_REQUIRE(nbad == 0, PSOLVE_HIP_ENUMERIC, "schwarz: non-finite pivot in a domain matrix");
}

void SchwarzPrecond::apply(Context &ctx, const double *d_r, double *d_z, const int *done)
{
    hipStream_t s = ctx.stream;
    const Launch L = ctx.launch_config();
    const int nl = (int)lv_.size();
    auto grid_for = [&](int64_t waves) { return (int)std::min<int64_t>(L.grid, std::max<int64_t>(1, (waves + 3) / 4)); };
    // restriction chain: r_1 = sums of r, r_2 = sums of r_1, ...
    for (int l = 1; l < nl; ++l) {
        const double *rf = l == 1 ? d_r : lv_[(size_t)l - 1]->r.ptr;
        hipLaunchKernelGGL(schwarz_restrict_kernel, dim3(grid_for(lv_[(size_t)l]->n)), dim3(256), 0, s, lv_[(size_t)l - 1]->n, rf,
                           lv_[(size_t)l]->n, lv_[(size_t)l]->r.ptr, bs_, done);
    }
    // coarsest first; every level adds the injected correction of the level above it
    for (int l = nl - 1; l >= 0; --l) {
        Level &lv = *lv_[(size_t)l];
        const double *r = l == 0 ? d_r : lv.r.ptr;
        double *z = l == 0 ? d_z : lv.z.ptr;
        const double *zc = l + 1 < nl ? lv_[(size_t)l + 1]->z.ptr : nullptr;
        hipLaunchKernelGGL(schwarz_block_apply_kernel, dim3(grid_for(lv.nblk)), dim3(256), 0, s, lv.n, lv.nblk, lv.inv.ptr, r, zc,
                           z, bs_, done);
    }
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
