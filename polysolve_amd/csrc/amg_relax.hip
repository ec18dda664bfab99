// amg_relax.hip -- round 5: the other classes amgcl's runtime wrappers build when the reference forwards its free strings
// (/root/reference/linear-solver-spec.json:393-397 relax `type`, :423-427 coarsening `type`; AMGCL.cpp:67-92, 178-181), on the
// kernels this backend already has.
//
//   * relaxation damped_jacobi (amgcl/relaxation/damped_jacobi.hpp) and spai0 (amgcl/relaxation/spai0.hpp): both are one
//     step x <- x + M (rhs - A x) with a (block) diagonal M -- the fused Chebyshev step of the product kernels with
//     alpha = 1, beta = 0 and M in the place of the inverted diagonal (kernels.hip: SPMV_CHEB; spmv_bsr3_dma's block
//     epilogue).  Here: the kernels that compute M.
//         damped_jacobi: M_i = damping * inverse(a_ii)          (backend::vmul(damping, dia, tmp, 1, x): the scaling first)
//         spai0:         M_i = inverse(sum_j |a_ij|^2) * a_ii   (block value types: Frobenius norms, the diagonal block)
//   * chebyshev with scale = false: M = identity (1.0 * r is r, bit for bit), the radius by power iterations on A itself.
//   * coarsening aggregation (amgcl/coarsening/aggregation.hpp): P = the tentative prolongation (one entry 1 per row, the
//     identity block for block value types), Galerkin operator scaled by 1 / over_interp.
//   * direct_coarse = true (amgcl/amg.hpp: the coarsest level gets a direct solver; builtin backend: skyline_lu): the
//     coarsest operator (SPD, at most a few thousand rows) is inverted densely ON the device -- Gauss-Jordan without
//     pivoting, one launch per column, ping-pong between two n x n buffers -- and applying the solver is one dense
//     matrix-vector product.  Same solution as the oracle's dense Cholesky solve up to rounding.
#include "amg_symbolic.hpp"

#include <algorithm>
#include <cmath>

namespace psolve {

namespace {

// scalar rows: type 1 damped_jacobi, 2 spai0, 3 identity (chebyshev.scale = false)
__global__ __launch_bounds__(kBlock) void relax_scaling_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                const double *__restrict__ val, int type, double damping,
                                                                double *__restrict__ m, int *__restrict__ bad)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (type == 3) {
            m[i] = 1.0;
            continue;
        }
        double num = 0.0, den = 0.0;
        bool has = false;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const double v = val[j], nv = fabs(v);
            den += nv * nv;
            if (col[j] == i) {
                num += v;
                has = true;
            }
        }
        double out;
        if (type == 1) out = has ? damping * (1.0 / num) : 0.0;
        else out = (1.0 / den) * num;
        if (!isfinite(out)) atomicAdd(bad, 1);
        m[i] = out;
    }
}

// block rows of a BlockGraph (values b*b per block, row-major): type 1: M = damping * dinv (dinv = inverted diagonal blocks,
// already computed), 2: M = (1 / sum_j ||A_ij||_F^2) D_i, 3: identity blocks
template <int B>
__global__ __launch_bounds__(kBlock) void block_relax_scaling_kernel(int nb, const int *__restrict__ bptr,
                                                                      const double *__restrict__ bval,
                                                                      const int *__restrict__ didx, int type, double damping,
                                                                      double *__restrict__ m, int *__restrict__ bad)
{
    constexpr int BB = B * B;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double *M = m + (size_t)i * BB;
        if (type == 3) {
#pragma unroll
            for (int k = 0; k < BB; ++k) M[k] = (k % (B + 1) == 0) ? 1.0 : 0.0;
        } else if (type == 1) {
#pragma unroll
            for (int k = 0; k < BB; ++k) M[k] = damping * M[k];
        } else {
            double den = 0.0;
            for (int j = bptr[i]; j < bptr[i + 1]; ++j) {
                const double *v = bval + (size_t)j * BB;
                double s2 = 0.0;
#pragma unroll
                for (int k = 0; k < BB; ++k) s2 += v[k] * v[k];
                const double nv = sqrt(s2);
                den += nv * nv;
            }
            const double inv = 1.0 / den;
            const int d = didx[i];
            if (!isfinite(inv)) atomicAdd(bad, 1);
#pragma unroll
            for (int k = 0; k < BB; ++k) M[k] = d >= 0 ? inv * bval[(size_t)d * BB + k] : 0.0;
        }
    }
}

// ---- tentative prolongation -------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void tentative_count_kernel(int n, const int *__restrict__ id, int *__restrict__ cnt)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) cnt[i] = id[i] >= 0 ? 1 : 0;
}

// ptr = the scanned counts; one entry (id[i], one) per kept row; block pattern: valb gets the identity block
template <int B>
__global__ __launch_bounds__(kBlock) void tentative_fill_kernel(int n, const int *__restrict__ id, const int *__restrict__ ptr,
                                                                 int *__restrict__ col, double *__restrict__ val)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (id[i] < 0) continue;
        const int p = ptr[i];
        col[p] = id[i];
        if (val) {
#pragma unroll
            for (int k = 0; k < B * B; ++k) val[(size_t)p * B * B + k] = (k % (B + 1) == 0) ? 1.0 : 0.0;
        }
    }
}

__global__ __launch_bounds__(kBlock) void scale_values_kernel(int64_t n, double s, double *__restrict__ v)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) v[i] = s * v[i];
}

// ---- dense inverse of the coarsest operator ---------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void dense_from_csr_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                 const double *__restrict__ val, double *__restrict__ a)
{
    // (a zeroed by the caller) one wave per row; duplicates add up
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nw = (gridDim.x * kBlock) >> 6;
    for (int i = wave; i < n; i += nw)
        for (int j = rowptr[i] + lane; j < rowptr[i + 1]; j += 64) atomicAdd(&a[(size_t)i * n + col[j]], val[j]);
}

// one Gauss-Jordan step on column k, out of place: dst = the matrix after eliminating column k of src.
//   p = src[k][k];  dst[k][k] = 1 / p;  dst[k][j] = src[k][j] / p;  dst[i][k] = -src[i][k] / p;
//   dst[i][j] = src[i][j] - src[i][k] * src[k][j] / p            (i, j != k)
// After n steps dst holds the inverse (no pivoting: the operator is SPD).  flag: a pivot that is not positive and finite.
__global__ __launch_bounds__(kBlock) void gauss_jordan_step_kernel(int n, int k, const double *__restrict__ src,
                                                                    double *__restrict__ dst, int *__restrict__ flag)
{
    const double p = src[(size_t)k * n + k];
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(p > 0.0 && isfinite(p))) atomicAdd(flag, 1);
    const double ip = 1.0 / p;
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
        double out;
        if (i == k) out = j == k ? ip : src[e] * ip;
        else if (j == k) out = -src[e] * ip;
        else out = src[e] - src[(size_t)i * n + k] * (src[(size_t)k * n + j] * ip);
        dst[e] = out;
    }
}


// ---- the same inverse in BLOCKS of kGjB columns (round 6) -------------------------------------------------------------
// n single-column steps are n launches that each stream the whole matrix (n = 1641, the coarsest level of configs[2] under
// "amg.aggregation" compact: ~20 ms).  A block step k eliminates kGjB columns K = [k0, k0 + bk) at once, in place:
//   Pinv = A[K, K]^-1                      (gj_pivot_kernel: the single-column steps above on a tile in LDS, one workgroup)
//   R = Pinv A[K, :],  C = A[:, K]          (gj_panels_kernel: the new row panel and a copy of the old column panel)
//   A[i, j] -= C[i, :] R[:, j]              (gj_update_kernel, i, j not in K: a rank-bk update in 64 x 64 tiles)
//   A[K, j] = R[:, j];  A[i, K] = -C[i, :] Pinv;  A[K, K] = Pinv
// -- the block form of the same formulas: n / 32 x 3 launches, 2 n^3 flops.  No pivoting (SPD); flag as above.
constexpr int kGjB = 32;

__global__ __launch_bounds__(kBlock) void gj_pivot_kernel(int n, int k0, int bk, const double *__restrict__ A,
                                                           double *__restrict__ pinv, int *__restrict__ flag)
{
    __shared__ double t[2][kGjB][kGjB + 1];
    for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) {
        const int i = e / kGjB, j = e - i * kGjB;
        t[0][i][j] = (i < bk && j < bk) ? A[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < bk; ++s) {
        const double p = t[cur][s][s];
        if (threadIdx.x == 0 && !(p > 0.0 && isfinite(p))) atomicAdd(flag, 1);
        const double ip = 1.0 / p;
        for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) {
            const int i = e / kGjB, j = e - i * kGjB;
            double out;
            if (i == s) out = j == s ? ip : t[cur][i][j] * ip;
            else if (j == s) out = -t[cur][i][j] * ip;
            else out = t[cur][i][j] - t[cur][i][s] * (t[cur][s][j] * ip);
            t[cur ^ 1][i][j] = out;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) {
        const int i = e / kGjB, j = e - i * kGjB;
        pinv[e] = (i < bk && j < bk) ? t[cur][i][j] : 0.0;
    }
}

// R[t][j] = sum_u Pinv[t][u] A[k0 + u][j]  (bk x n, row-major with stride n);  C[i][t] = A[i][k0 + t]  (n x kGjB)
__global__ __launch_bounds__(kBlock) void gj_panels_kernel(int n, int k0, int bk, const double *__restrict__ A,
                                                            const double *__restrict__ pinv, double *__restrict__ R,
                                                            double *__restrict__ C)
{
    __shared__ double ps[kGjB * kGjB];
    for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) ps[e] = pinv[e];
    __syncthreads();
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        double a[kGjB];
#pragma unroll
        for (int u = 0; u < kGjB; ++u) a[u] = u < bk ? A[(size_t)(k0 + u) * n + j] : 0.0;
        for (int tt = 0; tt < bk; ++tt) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < kGjB; ++u) acc += ps[tt * kGjB + u] * a[u];
            R[(size_t)tt * n + j] = acc;
        }
    }
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < (int64_t)n * kGjB; e += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(e / kGjB), tt = (int)(e - (int64_t)i * kGjB);
        C[e] = tt < bk ? A[(size_t)i * n + k0 + tt] : 0.0;
    }
}

// one 64 x 64 tile of the matrix per workgroup, 4 x 4 entries per thread
__global__ __launch_bounds__(kBlock) void gj_update_kernel(int n, int k0, int bk, double *__restrict__ A,
                                                            const double *__restrict__ pinv, const double *__restrict__ R,
                                                            const double *__restrict__ C)
{
    __shared__ double cs[64][kGjB + 1];
    __shared__ double rs[kGjB][64 + 1];
    __shared__ double ps[kGjB * kGjB];
    const int tiles = (n + 63) / 64;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    for (int e = threadIdx.x; e < kGjB * kGjB; e += kBlock) ps[e] = pinv[e];
    for (int tile = blockIdx.x; tile < tiles * tiles; tile += gridDim.x) {
        const int i0 = (tile / tiles) * 64, j0 = (tile % tiles) * 64;
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * kGjB; e += kBlock) {
            const int r = e / kGjB, tt = e - r * kGjB;
            cs[r][tt] = (i0 + r < n) ? C[(size_t)(i0 + r) * kGjB + tt] : 0.0;
        }
        for (int e = threadIdx.x; e < kGjB * 64; e += kBlock) {
            const int tt = e / 64, c = e - tt * 64;
            rs[tt][c] = (tt < bk && j0 + c < n) ? R[(size_t)tt * n + j0 + c] : 0.0;
        }
        __syncthreads();
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
        for (int tt = 0; tt < kGjB; ++tt) {
            double cv[4], rv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) cv[a] = cs[ty * 4 + a][tt];
#pragma unroll
            for (int b = 0; b < 4; ++b) rv[b] = rs[tt][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += cv[a] * rv[b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = i0 + ty * 4 + a;
            if (i >= n) continue;
            const bool ik = i >= k0 && i < k0 + bk;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = j0 + tx * 4 + b;
                if (j >= n) continue;
                const bool jk = j >= k0 && j < k0 + bk;
                double out;
                if (ik && jk) out = ps[(i - k0) * kGjB + (j - k0)];
                else if (ik) out = rs[i - k0][tx * 4 + b];
                else if (jk) {
                    double d = 0.0;
                    for (int tt = 0; tt < bk; ++tt) d += cs[ty * 4 + a][tt] * ps[tt * kGjB + (j - k0)];
                    out = -d;
                } else out = A[(size_t)i * n + j] - acc[a][b];
                A[(size_t)i * n + j] = out;
            }
        }
    }
}

// y = Ainv x, one wave per row (n <= a few thousand: a row is a few KB, the matrix tens of MB at most)
__global__ __launch_bounds__(kBlock) void dense_matvec_kernel(int n, const double *__restrict__ a, const double *__restrict__ x,
                                                               double *__restrict__ y, const int *__restrict__ done_flag)
{
    if (done_flag && *done_flag) return;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nw = (gridDim.x * kBlock) >> 6;
    for (int i = wave; i < n; i += nw) {
        const double *row = a + (size_t)i * n;
        double s = 0.0;
        for (int j = lane; j < n; j += 64) s += row[j] * x[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) y[i] = s;
    }
}

// ---- the coarsest level's smoother as ONE dense operator ("amg.coarse_dense") --------------------------------
// A coarsest level that is relaxed, not solved (direct_coarse = false, the reference's configuration, AMGCL.cpp:46), is
// visited with x = 0 and gets npre + npost smoother applications: a FIXED linear map rhs -> x.  For a level of a few hundred
// rows that map is a small dense matrix B = the same recurrence run on the identity instead of on one right-hand side:
// element (i, j) of the iterate is the i-th entry of the vector iterate for rhs = e_j.  One launch per smoothing step at setup
// (n^2 threads); a visit in the cycle is then one dense matrix-vector product instead of (npre + npost) x degree launches of
// a few microseconds each (reference configuration at 216^3: 248 launches = 5.5 % of an iteration).
//   first: X_in = 0 (no product); F = identity.  BS = 1: m = the diagonal scaling; BS > 1: mblk = the block scaling.
template <int BS>
__global__ __launch_bounds__(kBlock) void dense_smoother_step_kernel(int n, const int *__restrict__ rowptr,
                                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                                      const double *__restrict__ m, const double *__restrict__ mblk,
                                                                      const double *__restrict__ xin, double *__restrict__ pm,
                                                                      double *__restrict__ xout, double alpha, double beta,
                                                                      int first)
{
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
        double res;
        if (BS == 1) {
            double r = (i == j) ? 1.0 : 0.0;
            if (!first) {
                double acc = 0.0;
                for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) acc += val[k] * xin[(size_t)col[k] * n + j];
                r = r - acc;
            }
            res = m[i] * r;
        } else {
            const int node = i / BS, rr = i - node * BS;
            res = 0.0;
#pragma unroll
            for (int c = 0; c < BS; ++c) {
                const int row = node * BS + c;
                double r = (row == j) ? 1.0 : 0.0;
                if (!first) {
                    double acc = 0.0;
                    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) acc += val[k] * xin[(size_t)col[k] * n + j];
                    r = r - acc;
                }
                res += mblk[(size_t)node * BS * BS + rr * BS + c] * r;
            }
        }
        const double pn = (beta != 0.0) ? alpha * res + beta * pm[e] : alpha * res;
        pm[e] = pn;
        xout[e] = first ? pn : xin[e] + pn;
    }
}

} // namespace

void launch_dense_smoother_step(const Launch &L, const CsrDev &A, int bs, const double *m, const double *mblk, const double *xin,
                                double *pm, double *xout, double alpha, double beta, bool first)
{
    const int64_t total = (int64_t)A.n * A.n;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)L.num_cus * 8, (total + kBlock - 1) / kBlock));
    dim3 g(grid), blk(kBlock);
    if (bs == 3)
        hipLaunchKernelGGL(dense_smoother_step_kernel<3>, g, blk, 0, L.stream, A.n, A.rowptr, A.col, A.val, m, mblk, xin, pm, xout,
                           alpha, beta, first ? 1 : 0);
    else if (bs == 2)
        hipLaunchKernelGGL(dense_smoother_step_kernel<2>, g, blk, 0, L.stream, A.n, A.rowptr, A.col, A.val, m, mblk, xin, pm, xout,
                           alpha, beta, first ? 1 : 0);
    else
        hipLaunchKernelGGL(dense_smoother_step_kernel<1>, g, blk, 0, L.stream, A.n, A.rowptr, A.col, A.val, m, mblk, xin, pm, xout,
                           alpha, beta, first ? 1 : 0);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_relax_scaling(const Launch &L, const CsrDev &A, int type, double damping, double *m, int *bad)
{
    hipLaunchKernelGGL(relax_scaling_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val, type, damping, m,
                       bad);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_block_relax_scaling(const Launch &L, const BlockGraph &G, int type, double damping, double *m, int *bad)
{
    PS_REQUIRE(G.b == 2 || G.b == 3, PSOLVE_HIP_EINVAL, "block relaxation: block_size must be 2 or 3");
    if (G.b == 3)
        hipLaunchKernelGGL(block_relax_scaling_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.ptr.ptr, G.val.ptr,
                           G.didx.ptr, type, damping, m, bad);
    else
        hipLaunchKernelGGL(block_relax_scaling_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.ptr.ptr, G.val.ptr,
                           G.didx.ptr, type, damping, m, bad);
    PS_HIP_CHECK(hipGetLastError());
}

int64_t device_tentative_prolongation(const Launch &L, int n_nodes, const int *id, int b, DeviceBuffer<int> &ptr,
                                      DeviceBuffer<int> &col, DeviceBuffer<double> *val, SymbolicScratch &S)
{
    ptr.ensure((size_t)n_nodes + 2);
    hipLaunchKernelGGL(tentative_count_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n_nodes, id, ptr.ptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t nnz = device_exclusive_scan(L, ptr.ptr, n_nodes, S);
    col.ensure((size_t)nnz + 4);
    if (val) val->ensure((size_t)nnz * b * b + 4);
    double *v = val ? val->ptr : nullptr;
    if (b == 3)
        hipLaunchKernelGGL(tentative_fill_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, n_nodes, id, ptr.ptr, col.ptr, v);
    else if (b == 2)
        hipLaunchKernelGGL(tentative_fill_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, n_nodes, id, ptr.ptr, col.ptr, v);
    else
        hipLaunchKernelGGL(tentative_fill_kernel<1>, dim3(L.grid), dim3(kBlock), 0, L.stream, n_nodes, id, ptr.ptr, col.ptr, v);
    PS_HIP_CHECK(hipGetLastError());
    return nnz;
}

void launch_scale_values(const Launch &L, int64_t n, double s, double *v)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(scale_values_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, s, v);
    PS_HIP_CHECK(hipGetLastError());
}

void device_dense_inverse(const Launch &L, const CsrDev &A, DeviceBuffer<double> &inv, DeviceBuffer<double> &work)
{
    const int n = A.n;
    PS_REQUIRE(n > 0 && n <= kDirectCoarseMaxRows, PSOLVE_HIP_EINVAL,
               "amg.direct_coarse: the coarsest level has " + std::to_string(n) + " rows (at most " +
                   std::to_string(kDirectCoarseMaxRows) + " are inverted densely); lower amg.coarse_enough or raise amg.max_levels");
    const size_t nn = (size_t)n * n;
    inv.ensure(nn + 2);
    work.ensure(nn + 2);
    DeviceBuffer<int> flag;
    flag.ensure(2);
    PS_HIP_CHECK(hipMemsetAsync(flag.ptr, 0, 2 * sizeof(int), L.stream));
    if (n > 4 * kGjB) { // (work: n^2 doubles hold the panels, 64 n + 1024)
        // block steps, in place in `inv`; `work` holds the two panels (kGjB x n and n x kGjB) and the inverted pivot block
        double *M = inv.ptr, *R = work.ptr, *C = R + (size_t)kGjB * n, *pinv = C + (size_t)n * kGjB;
        PS_HIP_CHECK(hipMemsetAsync(M, 0, nn * sizeof(double), L.stream));
        hipLaunchKernelGGL(dense_from_csr_kernel, dim3(std::max(1, std::min(L.grid, (n + 3) / 4))), dim3(kBlock), 0, L.stream, n, A.rowptr,
                           A.col, A.val, M);
        const int tiles = (n + 63) / 64;
        const int gu = std::max(1, std::min(tiles * tiles, L.num_cus * 8)), gp = std::max(1, std::min(L.num_cus * 2, (n + kBlock - 1) / kBlock));
        for (int k0 = 0; k0 < n; k0 += kGjB) {
            const int bk = std::min(kGjB, n - k0);
            hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(kBlock), 0, L.stream, n, k0, bk, M, pinv, flag.ptr);
            hipLaunchKernelGGL(gj_panels_kernel, dim3(gp), dim3(kBlock), 0, L.stream, n, k0, bk, M, pinv, R, C);
            hipLaunchKernelGGL(gj_update_kernel, dim3(gu), dim3(kBlock), 0, L.stream, n, k0, bk, M, pinv, R, C);
        }
    } else {
    // n steps ping-pong; start in the buffer from which the last step lands in `inv`
    double *src = (n & 1) ? work.ptr : inv.ptr, *dst = (n & 1) ? inv.ptr : work.ptr;
    PS_HIP_CHECK(hipMemsetAsync(src, 0, nn * sizeof(double), L.stream));
    hipLaunchKernelGGL(dense_from_csr_kernel, dim3(std::max(1, std::min(L.grid, (n + 3) / 4))), dim3(kBlock), 0, L.stream, n, A.rowptr,
                       A.col, A.val, src);
    PS_HIP_CHECK(hipGetLastError());
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)L.num_cus * 8, (nn + kBlock - 1) / kBlock));
    for (int k = 0; k < n; ++k) {
        hipLaunchKernelGGL(gauss_jordan_step_kernel, dim3(grid), dim3(kBlock), 0, L.stream, n, k, src, dst, flag.ptr);
        std::swap(src, dst);
    }
    }
    PS_HIP_CHECK(hipGetLastError());
    int bad = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&bad, flag.ptr, sizeof(int), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    PS_REQUIRE(bad == 0, PSOLVE_HIP_ENUMERIC, "amg.direct_coarse: the coarsest operator is not positive definite (a pivot of its dense factorization is not positive)");
}

void launch_dense_matvec(const Launch &L, int n, const double *ainv, const double *x, double *y, const int *done_flag)
{
    const int grid = std::max(1, std::min(L.num_cus * 4, (n + 3) / 4));
    hipLaunchKernelGGL(dense_matvec_kernel, dim3(grid), dim3(kBlock), 0, L.stream, n, ainv, x, y, done_flag);
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
