// amg_block.hip -- device-side coarsening with block value types (polysolve's AMGCL_Block<3>,
// /root/reference/src/polysolve/linear/AMGCL.cpp:243-302; host restatement: amg_setup.cpp block section).
//
// The block graph of the level operator (b x b blocks, zero-filled), the strength test on blocks
// (eps^2 tr(D_i D_j) < tr(A_ij A_ij)), the block Gershgorin bound and the block-smoothed prolongation
// P = (I - omega D_f^-1 A_f) P_tent are computed here; P is then expanded to scalar CSR with full blocks,
// and R = P^T, A P, R (A P) are the scalar products of amg_symbolic.hip / kernels.hip -- exactly the
// split the host construction uses, so both give the same hierarchy.
#include "amg_symbolic.hpp"

namespace psolve {

namespace {

constexpr int kMaxB = 4;

// Gauss-Jordan with partial pivoting, the same operation order as invert_block() in amg_setup.cpp
__device__ void invert_block_dev(int b, const double *X, double *Y)
{
    double a[kMaxB * kMaxB], inv[kMaxB * kMaxB];
    for (int i = 0; i < b * b; ++i) {
        a[i] = X[i];
        inv[i] = 0.0;
    }
    for (int i = 0; i < b; ++i) inv[i * b + i] = 1.0;
    for (int c = 0; c < b; ++c) {
        int piv = c;
        for (int r = c + 1; r < b; ++r)
            if (fabs(a[r * b + c]) > fabs(a[piv * b + c])) piv = r;
        if (piv != c)
            for (int k = 0; k < b; ++k) {
                double t = a[c * b + k];
                a[c * b + k] = a[piv * b + k];
                a[piv * b + k] = t;
                t = inv[c * b + k];
                inv[c * b + k] = inv[piv * b + k];
                inv[piv * b + k] = t;
            }
        const double d = 1.0 / a[c * b + c];
        for (int k = 0; k < b; ++k) {
            a[c * b + k] *= d;
            inv[c * b + k] *= d;
        }
        for (int r = 0; r < b; ++r) {
            if (r == c) continue;
            const double f = a[r * b + c];
            if (f == 0.0) continue;
            for (int k = 0; k < b; ++k) {
                a[r * b + k] -= f * a[c * b + k];
                inv[r * b + k] -= f * inv[c * b + k];
            }
        }
    }
    for (int i = 0; i < b * b; ++i) Y[i] = inv[i];
}

// the same elimination with the size known at compile time: every index is static, the two 3 x 3 work arrays live in
// registers (the generic version indexes them dynamically: 144 bytes of scratch per lane in every kernel that inlines it)
template <int B> __device__ __forceinline__ void invert_block_static(const double *X, double *Y)
{
    double a[B * B], inv[B * B];
#pragma unroll
    for (int i = 0; i < B * B; ++i) {
        a[i] = X[i];
        inv[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < B; ++i) inv[i * B + i] = 1.0;
#pragma unroll
    for (int c = 0; c < B; ++c) {
        // the pivot row: the first of the largest magnitudes, as the generic search picks it; found without indexing a[] by it
        int piv = c;
        double best = fabs(a[c * B + c]);
#pragma unroll
        for (int r = c + 1; r < B; ++r) {
            const double v = fabs(a[r * B + c]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
#pragma unroll
        for (int r = c + 1; r < B; ++r)
            if (r == piv) {
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    double t = a[c * B + k];
                    a[c * B + k] = a[r * B + k];
                    a[r * B + k] = t;
                    t = inv[c * B + k];
                    inv[c * B + k] = inv[r * B + k];
                    inv[r * B + k] = t;
                }
            }
        const double d = 1.0 / a[c * B + c];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            a[c * B + k] *= d;
            inv[c * B + k] *= d;
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            if (r == c) continue;
            const double f = a[r * B + c];
            if (f == 0.0) continue;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                a[r * B + k] -= f * a[c * B + k];
                inv[r * B + k] -= f * inv[c * B + k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < B * B; ++i) Y[i] = inv[i];
}

__device__ __forceinline__ double trace_of_product(int b, const double *X, const double *Y)
{
    // trace(X Y) with the summation order of blk_mul + blk_trace
    double t = 0.0;
    for (int i = 0; i < b; ++i) {
        double s = 0.0;
        for (int k = 0; k < b; ++k) s += X[i * b + k] * Y[k * b + i];
        t += s;
    }
    return t;
}

__device__ __forceinline__ double fro_norm(int bb, const double *X)
{
    double s = 0.0;
    for (int i = 0; i < bb; ++i) s += X[i] * X[i];
    return sqrt(s);
}

__global__ __launch_bounds__(kBlock) void strided_ptr_kernel(int nb, int b, const int *__restrict__ rowptr,
                                                              int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i <= nb; i += gridDim.x * kBlock) out[i] = rowptr[i * b];
}

// bval (zero-filled) += the scalar entries; didx[i] = position of the diagonal block of block row i (or -1).
// 32 lanes share a block row (a 3 x 3 elasticity row has 243 scalar entries, a coarse one a thousand: one thread
// per block row left the coarse levels with a few thousand busy lanes); every scalar entry has its own slot, so the
// lanes never meet.
// (B > 0: the block size as a compile-time constant -- the divisions by it per entry are what the kernel's time goes into)
#define PS_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

constexpr int kValCap = 32; // blocks of a full node staged through LDS by block_values_kernel (288 doubles per lane group)
template <int B>
__global__ __launch_bounds__(kBlock) void block_values_kernel(int nb, int b_rt, const int *__restrict__ rowptr,
                                                               const int *__restrict__ col,
                                                               const double *__restrict__ val,
                                                               const int *__restrict__ bptr,
                                                               const int *__restrict__ bcol, double *__restrict__ bval,
                                                               int *__restrict__ didx)
{
    const int b = B > 0 ? B : b_rt;
    constexpr int G = 32;
    const int bb = b * b, lane = threadIdx.x % G;
    const int groups = gridDim.x * kBlock / G;
    __shared__ double vstage[kBlock / G][kValCap * (B > 0 ? B * B : 1)];
    for (int ib = (blockIdx.x * kBlock + threadIdx.x) / G; ib < nb; ib += groups) {
        const int beg = bptr[ib], end = bptr[ib + 1];
        const int j0 = rowptr[ib * b], j1 = rowptr[ib * b + b]; // the b scalar rows are one contiguous run
        int d = -1;
        // a node whose b scalar rows store their blocks in full (the usual case: (end - beg) b entries each, in block
        // order): entry p of a row belongs to block p / b, no search (round 4; configs[2], level 0: 3.25 ms with the
        // bisection per entry)
        bool full = true;
        for (int r = 0; r < b; ++r) full = full && (rowptr[ib * b + r + 1] - rowptr[ib * b + r] == (end - beg) * b);
        if (B > 0 && full && end - beg <= kValCap) {
            // (round 6) a full node of up to kValCap blocks: its b scalar rows are one contiguous run of the CSR values and its
            // blocks one contiguous run of the block values -- the same numbers in another order.  Read the run, put every
            // value where it belongs in LDS, write the run: both sides of the copy in whole lines (the direct store wrote 24
            // bytes here, 24 bytes there: level 0 of configs[2] 1.8 ms for 3.8 GB)
            const int nblk = end - beg, rowlen = nblk * b;
            double *stage = vstage[(threadIdx.x / G)];
            bool ok = true;
#pragma unroll
            for (int r = 0; r < (B > 0 ? B : 1); ++r) {
                const int jr = rowptr[ib * b + r];
                for (int p = lane; p < rowlen; p += G) {
                    const int cj = col[jr + p], blk = p / b, cc = p - blk * b;
                    ok = ok && cj == bcol[beg + blk] * b + cc;
                    stage[blk * bb + r * b + cc] = val[jr + p];
                    if (cj / b == ib) d = beg + blk;
                }
            }
            // (all lanes of the group must agree: a node whose entries are not where a full node's are takes the search below)
            const unsigned long long bad = __ballot(!ok);
            const unsigned gbad = (unsigned)(bad >> (((threadIdx.x / G) & 1) ? 32 : 0));
            PS_WAVE_SYNC();
            if (gbad == 0) {
                double *dst = bval + (size_t)beg * bb;
                for (int t = lane; t < nblk * bb; t += G) dst[t] = stage[t];
#pragma unroll
                for (int off = G >> 1; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
                if (lane == 0) didx[ib] = d;
                PS_WAVE_SYNC();
                continue;
            }
            d = -1;
            PS_WAVE_SYNC();
        }
        { // (round 6) the blocks of THIS row start from zero here -- there is no memset of all block values any more: the rows
          // staged above, nearly all of them, write every entry of their blocks
            double *dst = bval + (size_t)beg * bb;
            for (int t = lane; t < (end - beg) * bb; t += G) dst[t] = 0.0;
            __threadfence_block();
            PS_WAVE_SYNC();
        }
        for (int j = j0 + lane; j < j1; j += G) {
            int r = 0;
            while (r + 1 < b && j >= rowptr[ib * b + r + 1]) ++r;
            const int cj = col[j];
            const int cb = cj / b, cc = cj % b;
            int lo;
            const int guess = beg + (j - rowptr[ib * b + r]) / b;
            if (full && bcol[guess] == cb) {
                // (a full node: (block, r, cc) is met exactly once -- a plain store, the zeroed block is not read back:
                // 2 GB less at configs[2]'s level 0)
                bval[(size_t)guess * bb + r * b + cc] = val[j];
                if (cb == ib) d = guess;
                continue;
            } else {
                lo = beg;
                int hi = end;
                while (lo < hi) {
                    const int mid = lo + ((hi - lo) >> 1);
                    if (bcol[mid] < cb) lo = mid + 1; else hi = mid;
                }
            }
            bval[(size_t)lo * bb + r * b + cc] += val[j]; // (an atomic add without a return value instead: 2.5 -> 4.1 ms)
            if (cb == ib) d = lo;
        }
        // any lane that met the diagonal block knows its position
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
        if (lane == 0) didx[ib] = d;
    }
}

__device__ __forceinline__ bool block_is_strong(int b, int i, int c, const double *v, const double *di,
                                                const double *dc, double eps2)
{
    if (c == i) return false;
    const double lhs = (di && dc) ? eps2 * trace_of_product(b, di, dc) : eps2 * 0.0;
    return lhs < trace_of_product(b, v, v);
}

// pass 1 (FILL = false): strong flags + per-row counts of the compacted graph (strong + diagonal);
// pass 2: the graph's columns and the start state of the aggregation sweep
template <bool FILL>
__global__ __launch_bounds__(kBlock) void block_strength_kernel(int nb, int b, const int *__restrict__ bptr,
                                                                 const int *__restrict__ bcol,
                                                                 const double *__restrict__ bval,
                                                                 const int *__restrict__ didx, double eps2,
                                                                 unsigned char *__restrict__ strong,
                                                                 int *__restrict__ sptr, int *__restrict__ scol,
                                                                 int *__restrict__ id0)
{
    const int bb = b * b;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        int w = FILL ? sptr[i] : 0;
        bool any = false;
        const double *di = didx[i] >= 0 ? bval + (size_t)didx[i] * bb : nullptr;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j) {
            const int c = bcol[j];
            bool s;
            if (FILL) {
                s = strong[j] != 0;
            } else {
                const double *dc = didx[c] >= 0 ? bval + (size_t)didx[c] * bb : nullptr;
                s = block_is_strong(b, i, c, bval + (size_t)j * bb, di, dc, eps2);
                strong[j] = s ? 1 : 0;
            }
            if (s || c == i) {
                if (FILL) scol[w] = c;
                ++w;
                any = any || s;
            }
        }
        if (!FILL) sptr[i] = w;
        else id0[i] = any ? -1 : -2;
    }
}

__global__ __launch_bounds__(kBlock) void block_gershgorin_kernel(int nb, int b, const int *__restrict__ bptr,
                                                                   const double *__restrict__ bval,
                                                                   const int *__restrict__ didx,
                                                                   double *__restrict__ partials)
{
    __shared__ double red[kBlock];
    const int bb = b * b;
    double m = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double s = 0.0;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j) s += fro_norm(bb, bval + (size_t)j * bb);
        double dia[kMaxB * kMaxB], inv[kMaxB * kMaxB];
        for (int k = 0; k < bb; ++k) dia[k] = (k % (b + 1) == 0) ? 1.0 : 0.0;
        if (didx[i] >= 0)
            for (int k = 0; k < bb; ++k) dia[k] = bval[(size_t)didx[i] * bb + k];
        invert_block_dev(b, dia, inv);
        s *= fro_norm(bb, inv);
        m = fmax(m, s);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// block values of P on the block pattern (pbptr, pbcol): per block row the filtered diagonal (diagonal +
// weak blocks) is inverted, every strong / diagonal block adds its contribution to the aggregate of its
// column, in row order (= the stable sort + merge of block_smoothed_prolongation)
__global__ __launch_bounds__(kBlock) void block_prolongation_values_kernel(
    int nb, int b, const int *__restrict__ bptr, const int *__restrict__ bcol, const double *__restrict__ bval,
    const unsigned char *__restrict__ strong, const int *__restrict__ id, double omega,
    const int *__restrict__ pbptr, const int *__restrict__ pbcol, double *__restrict__ pbval)
{
    const int bb = b * b;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        const int pb = pbptr[i], pe = pbptr[i + 1];
        for (int k = pb * bb; k < pe * bb; ++k) pbval[k] = 0.0;
        double dia[kMaxB * kMaxB], dinv[kMaxB * kMaxB];
        for (int k = 0; k < bb; ++k) dia[k] = 0.0;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j)
            if (bcol[j] == i || !strong[j])
                for (int k = 0; k < bb; ++k) dia[k] += bval[(size_t)j * bb + k];
        invert_block_dev(b, dia, dinv);
        for (int k = 0; k < bb; ++k) dinv[k] *= -omega;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j) {
            const int ca = bcol[j];
            if (ca != i && !strong[j]) continue;
            const int cp = id[ca];
            if (cp < 0) continue;
            double v[kMaxB * kMaxB];
            if (ca == i) {
                for (int k = 0; k < bb; ++k) v[k] = (k % (b + 1) == 0) ? (1.0 - omega) : 0.0;
            } else {
                const double *Y = bval + (size_t)j * bb;
                for (int r = 0; r < b; ++r)
                    for (int c = 0; c < b; ++c) {
                        double s = 0.0;
                        for (int k = 0; k < b; ++k) s += dinv[r * b + k] * Y[k * b + c];
                        v[r * b + c] = s;
                    }
            }
            for (int k = pb; k < pe; ++k)
                if (pbcol[k] == cp) {
                    for (int q = 0; q < bb; ++q) pbval[(size_t)k * bb + q] += v[q];
                    break;
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// b = 3, one WAVE per block row (round 4): the row's blocks (9 nnzb doubles, contiguous) come in by whole-line loads and
// are parked in the wave's slice of LDS; lane j then works on block j.  The one-thread-per-block-row kernels above read
// their row 72 bytes at a time with the lanes of a wave ~1.9 KB apart (elasticity M = 100, level 0: Gershgorin 2.2 ms,
// prolongation values 9.3 ms for 1.9 GB of blocks).  Every sum keeps the order of the sequential loops (so the numbers
// are the generic kernels', bit for bit): what runs in parallel are the blocks, what adds them up is one lane in order.
// ---------------------------------------------------------------------------------------------
constexpr int kRowCap = 56;            // blocks of a row parked at a time (504 doubles per wave)

__device__ __forceinline__ void park_blocks3(const double *__restrict__ bval, int j0, int cnt, double *lds, int lane)
{
    const double *src = bval + (size_t)j0 * 9;
    for (int t = lane; t < cnt * 9; t += 64) lds[t] = src[t];
}

// (rows != nullptr: the bound of the nb listed block rows only -- the representatives of an operator's block-row kinds,
// whose bounds are those of all rows)
__global__ __launch_bounds__(kBlock) void block_gershgorin3_kernel(int nb, const int *__restrict__ bptr,
                                                                    const double *__restrict__ bval,
                                                                    const int *__restrict__ didx,
                                                                    double *__restrict__ partials,
                                                                    const int *__restrict__ rows_list)
{
    __shared__ double park[kBlock / 64][kRowCap * 9];
    __shared__ double fro[kBlock / 64][kRowCap];
    __shared__ double rowsum[kBlock / 64][64];
    __shared__ double red[kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = gridDim.x * (kBlock / 64);
    double m = 0.0;
    // a wave takes 64 consecutive block rows: their sums of block norms one row at a time (the row parked by all lanes,
    // the norms added in block order by one), then the 64 diagonal inverses side by side
    for (int i0 = (blockIdx.x * (kBlock / 64) + wave) * 64; i0 < nb; i0 += nwaves * 64) {
        const int rows = min(64, nb - i0);
        for (int rr = 0; rr < rows; ++rr) {
            const int i = rows_list ? rows_list[i0 + rr] : i0 + rr;
            const int jb = bptr[i], je = bptr[i + 1];
            double s = 0.0; // (lane 0's)
            for (int j0 = jb; j0 < je; j0 += kRowCap) {
                const int cnt = min(kRowCap, je - j0);
                park_blocks3(bval, j0, cnt, park[wave], lane);
                PS_WAVE_SYNC();
                if (lane < cnt) fro[wave][lane] = fro_norm(9, park[wave] + lane * 9);
                PS_WAVE_SYNC();
                if (lane == 0)
                    for (int t = 0; t < cnt; ++t) s += fro[wave][t];
                PS_WAVE_SYNC();
            }
            if (lane == 0) rowsum[wave][rr] = s;
        }
        PS_WAVE_SYNC();
        if (lane < rows) {
            const int i = rows_list ? rows_list[i0 + lane] : i0 + lane;
            double dia[9], inv[9];
            for (int k = 0; k < 9; ++k) dia[k] = (k % 4 == 0) ? 1.0 : 0.0;
            if (didx[i] >= 0)
                for (int k = 0; k < 9; ++k) dia[k] = bval[(size_t)didx[i] * 9 + k];
            invert_block_static<3>(dia, inv);
            m = fmax(m, rowsum[wave][lane] * fro_norm(9, inv));
        }
        PS_WAVE_SYNC();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmax(m, red[w]);
        partials[blockIdx.x] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// b = 3, round 6: one thread per BLOCK, then one thread per block row.  The wave-per-row kernels above are chains of
// dependent LDS steps with 27 of 64 lanes at work (level 0 of configs[2] under a random numbering: Gershgorin bound 2.4 ms,
// strength flags 2.6 ms for 1.9 GB of blocks: 0.8 TB/s).  What they compute per block -- a Frobenius norm, "is not exactly
// zero" -- needs nothing of the row: phase 1 streams the blocks, 256 per workgroup step through LDS by whole-line loads,
// one number (a flag) per block out; phase 2 walks the rows over those.  Every sum keeps the sequential loops' order.
// ---------------------------------------------------------------------------------------------
// MODE 0: out_d[j] = fro_norm(block j); MODE 1: flag[j] = 0 < trace(V V) (eps_strong = 0: "strong" before the diagonal test)
template <int MODE>
__global__ __launch_bounds__(kBlock) void block_stream3_kernel(int64_t nnzb, const double *__restrict__ bval,
                                                                double *__restrict__ out_d, unsigned char *__restrict__ flag)
{
    __shared__ double tile[kBlock * 9];
    for (int64_t j0 = (int64_t)blockIdx.x * kBlock; j0 < nnzb; j0 += (int64_t)gridDim.x * kBlock) {
        const int cnt = (int)min<int64_t>(kBlock, nnzb - j0);
        const double *src = bval + j0 * 9;
        for (int t = threadIdx.x; t < cnt * 9; t += kBlock) tile[t] = src[t];
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const double *v = tile + 9 * threadIdx.x;
            if (MODE == 0) out_d[j0 + threadIdx.x] = fro_norm(9, v);
            else flag[j0 + threadIdx.x] = (0.0 < trace_of_product(3, v, v)) ? 1 : 0;
        }
        __syncthreads();
    }
}

// rho bound of a block row: (sum of its blocks' norms, in block order) x norm of its inverted diagonal block
__global__ __launch_bounds__(kBlock) void block_gershgorin3_rows_kernel(int nb, const int *__restrict__ bptr,
                                                                         const double *__restrict__ bval,
                                                                         const double *__restrict__ fro,
                                                                         const int *__restrict__ didx,
                                                                         double *__restrict__ partials)
{
    __shared__ double red[kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double m = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double s = 0.0;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j) s += fro[j];
        double dia[9], inv[9];
        for (int k = 0; k < 9; ++k) dia[k] = (k % 4 == 0) ? 1.0 : 0.0;
        if (didx[i] >= 0)
            for (int k = 0; k < 9; ++k) dia[k] = bval[(size_t)didx[i] * 9 + k];
        invert_block_static<3>(dia, inv);
        m = fmax(m, s * fro_norm(9, inv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = fmax(m, red[w]);
        partials[blockIdx.x] = m;
    }
}

// eps_strong = 0: a block is strong if it is off the diagonal and not exactly zero; cnt[i] = strong blocks + the diagonal
__global__ __launch_bounds__(kBlock) void block_strong_rows_kernel(int nb, const int *__restrict__ bptr,
                                                                    const int *__restrict__ bcol,
                                                                    unsigned char *__restrict__ strong, int *__restrict__ cnt)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        int w = 0;
        for (int j = bptr[i]; j < bptr[i + 1]; ++j) {
            const bool diag = bcol[j] == i;
            const bool s = !diag && strong[j] != 0;
            strong[j] = s ? 1 : 0;
            if (s || diag) ++w;
        }
        cnt[i] = w;
    }
}

// Round 4 (last pass): a HALF wave per block row.  None of the row's phases has work for more than 36 lanes (9 for the sums of the
// filtered diagonal, 1 for its inverse, one per block for the products, one per entry of the row of P for the accumulation),
// and every phase is a chain of dependent LDS operations: two rows per wave double what a CU has in flight (level 0 of
// configs[2]: 3.36 ms for 2 GB of blocks).  Same operations in the same order per row.
constexpr int kPHalf = 32;     // lanes per block row
constexpr int kPRowCapH = 28;  // blocks of a row parked at a time by a half wave
constexpr int kPRowCap = 12;   // blocks of a row of P accumulated in LDS (longer rows accumulate in place)

__global__ __launch_bounds__(kBlock) void block_prolongation_values3_kernel(
    int nb, const int *__restrict__ bptr, const int *__restrict__ bcol, const double *__restrict__ bval,
    const unsigned char *__restrict__ strong, const int *__restrict__ id, double omega, const int *__restrict__ pbptr,
    const int *__restrict__ pbcol, double *__restrict__ pbval)
{
    constexpr int NG = kBlock / kPHalf;
    __shared__ double park[NG][kPRowCapH * 9]; // the row's blocks, then (in place) their contributions
    __shared__ int tgt[NG][kPRowCapH];         // aggregate of the block's column (-1: contributes nothing)
    __shared__ unsigned char flt[NG][kPRowCapH]; // the block belongs to the filtered diagonal
    __shared__ double dsh[NG][9];              // -omega D^-1
    __shared__ double pacc[NG][kPRowCap * 9];
    __shared__ int pcl[NG][kPRowCap];      // the columns of the row of P
    __shared__ unsigned msk[NG][kPRowCap]; // parked blocks feeding each of them
    const int g = threadIdx.x / kPHalf, lane = threadIdx.x % kPHalf, ngroups = gridDim.x * NG;
    for (int i = blockIdx.x * NG + g; i < nb; i += ngroups) {
        const int jb = bptr[i], je = bptr[i + 1];
        const int pb = pbptr[i], np = pbptr[i + 1] - pb;
        const bool in_lds = np <= kPRowCap;
        double *acc = in_lds ? pacc[g] : pbval + (size_t)pb * 9;
        for (int t = lane; t < np * 9; t += kPHalf) acc[t] = 0.0;
        // pass 1: the filtered diagonal (diagonal + weak blocks), every component summed in block order by one lane
        double dsum = 0.0; // (lanes 0..8)
        for (int j0 = jb; j0 < je; j0 += kPRowCapH) {
            const int cnt = min(kPRowCapH, je - j0);
            {
                const double *src = bval + (size_t)j0 * 9;
                for (int t = lane; t < cnt * 9; t += kPHalf) park[g][t] = src[t];
            }
            if (lane < cnt) flt[g][lane] = (bcol[j0 + lane] == i || !strong[j0 + lane]) ? 1 : 0;
            PS_WAVE_SYNC();
            if (lane < 9)
                for (int t = 0; t < cnt; ++t)
                    if (flt[g][t]) dsum += park[g][t * 9 + lane];
            PS_WAVE_SYNC();
        }
        if (lane < 9) dsh[g][lane] = dsum;
        PS_WAVE_SYNC();
        if (lane == 0) {
            double dia[9], dinv[9];
            for (int k = 0; k < 9; ++k) dia[k] = dsh[g][k];
            invert_block_static<3>(dia, dinv);
            for (int k = 0; k < 9; ++k) dsh[g][k] = dinv[k] * -omega;
        }
        PS_WAVE_SYNC();
        // pass 2: every strong / diagonal block's contribution to the aggregate of its column, added in block order
        for (int j0 = jb; j0 < je; j0 += kPRowCapH) {
            const int cnt = min(kPRowCapH, je - j0);
            if (je - jb > kPRowCapH) { // (a row that fits is still parked)
                const double *src = bval + (size_t)j0 * 9;
                for (int t = lane; t < cnt * 9; t += kPHalf) park[g][t] = src[t];
                PS_WAVE_SYNC();
            }
            if (lane < cnt) {
                const int ca = bcol[j0 + lane];
                int cp = -1;
                if (ca == i || strong[j0 + lane]) cp = id[ca];
                double *Y = park[g] + lane * 9, v[9];
                if (ca == i) {
                    for (int k = 0; k < 9; ++k) v[k] = (k % 4 == 0) ? (1.0 - omega) : 0.0;
                } else {
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) {
                            double sm = 0.0;
                            for (int k = 0; k < 3; ++k) sm += dsh[g][r * 3 + k] * Y[k * 3 + c];
                            v[r * 3 + c] = sm;
                        }
                }
                for (int k = 0; k < 9; ++k) Y[k] = v[k];
                tgt[g][lane] = cp < 0 ? -1 : cp;
            }
            PS_WAVE_SYNC();
            if (in_lds) {
                // (round 6) which parked blocks go to which of the row's (at most kPRowCap) entries of P, as bit masks: every
                // block's lane looks its aggregate up in the row's columns once, one ballot per entry collects the lanes; an
                // accumulator then walks only the blocks that feed it, in block order as before (the scan over all parked
                // blocks for every one of the 9 np accumulators was most of this kernel: 27 blocks x 72 accumulators per node)
                if (lane < np) pcl[g][lane] = pbcol[pb + lane];
                PS_WAVE_SYNC();
                int myslot = -1;
                if (lane < cnt) {
                    const int want = tgt[g][lane];
                    if (want >= 0)
                        for (int k = 0; k < np; ++k)
                            if (pcl[g][k] == want) {
                                myslot = k;
                                break;
                            }
                }
                for (int k = 0; k < np; ++k) {
                    const unsigned long long bal = __ballot(myslot == k);
                    if (lane == 0) msk[g][k] = (unsigned)(bal >> ((threadIdx.x & 32) ? 32 : 0));
                }
                PS_WAVE_SYNC();
                for (int t = lane; t < np * 9; t += kPHalf) {
                    const int k = t / 9, q = t - k * 9;
                    double a = acc[t];
                    unsigned m = msk[g][k];
                    while (m) {
                        const int u = __ffs((int)m) - 1;
                        m &= m - 1;
                        a += park[g][u * 9 + q];
                    }
                    acc[t] = a;
                }
            } else {
                for (int t = lane; t < np * 9; t += kPHalf) {
                    const int k = t / 9, q = t - k * 9, want = pbcol[pb + k];
                    double a = acc[t];
                    for (int u = 0; u < cnt; ++u)
                        if (tgt[g][u] == want) a += park[g][u * 9 + q];
                    acc[t] = a;
                }
            }
            PS_WAVE_SYNC();
        }
        if (in_lds)
            for (int t = lane; t < np * 9; t += kPHalf) pbval[(size_t)pb * 9 + t] = acc[t];
        PS_WAVE_SYNC();
    }
}

// block CSR -> scalar CSR with full b x b blocks (explicit zeros kept)
__global__ __launch_bounds__(kBlock) void expand_block_ptr_kernel(int nb, int b, const int *__restrict__ pbptr,
                                                                   int *__restrict__ ptr)
{
    for (int s = blockIdx.x * kBlock + threadIdx.x; s <= nb * b; s += gridDim.x * kBlock) {
        if (s == nb * b) {
            ptr[s] = pbptr[nb] * b * b;
            continue;
        }
        const int i = s / b, r = s % b;
        const int len = pbptr[i + 1] - pbptr[i];
        ptr[s] = pbptr[i] * b * b + r * len * b;
    }
}

__global__ __launch_bounds__(kBlock) void expand_block_entries_kernel(int nb, int b, const int *__restrict__ pbptr,
                                                                       const int *__restrict__ pbcol,
                                                                       const double *__restrict__ pbval,
                                                                       int *__restrict__ col, double *__restrict__ val,
                                                                       bool with_cols)
{
    // 32 lanes per block row stride its len * b * b scalar entries in the order they are stored (scalar row r of the block
    // row, then block k, then column c): consecutive lanes write consecutive entries.  (Round 4; one thread per block row
    // wrote b strided runs: P_0 of configs[2] 2.0 ms.)
    constexpr int G = 32;
    const int bb = b * b, lane = threadIdx.x % G;
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / G; i < nb; i += gridDim.x * (kBlock / G)) {
        const int pb = pbptr[i], len = pbptr[i + 1] - pb;
        const int total = len * bb, rowlen = len * b;
        const size_t base = (size_t)pb * bb;
        for (int q = lane; q < total; q += G) {
            const int r = q / rowlen, rem = q - r * rowlen, k = rem / b, c = rem - k * b;
            if (with_cols) col[base + q] = pbcol[pb + k] * b + c;
            if (val) val[base + q] = pbval[(size_t)(pb + k) * bb + r * b + c];
        }
    }
}

__global__ __launch_bounds__(kBlock) void count_flag_changes_kernel(int64_t n, const unsigned char *__restrict__ a,
                                                                     const unsigned char *__restrict__ b2,
                                                                     int *__restrict__ count)
{
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        c += a[i] != b2[i];
    if (c) atomicAdd(count, c);
}

} // namespace

int64_t device_block_graph(const Launch &L, const CsrDev &A, int b, BlockGraph &G, SymbolicScratch &S)
{
    PS_REQUIRE(b >= 2 && b <= kMaxB && A.n % b == 0, PSOLVE_HIP_EINVAL, "block_size does not divide the matrix size");
    const int nb = A.n / b;
    G.nb = nb;
    G.b = b;
    G.rowstart.ensure((size_t)nb + 1);
    hipLaunchKernelGGL(strided_ptr_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, b, A.rowptr, G.rowstart.ptr);
    PS_HIP_CHECK(hipGetLastError());
    // block row ib = the scalar entries [rowptr[ib b], rowptr[(ib + 1) b)), columns folded by / b
    G.nnzb = device_spgemm_symbolic(L, nb, G.rowstart.ptr, A.col, nullptr, nullptr, (A.n_ext + b - 1) / b, G.ptr, G.col,
                                    S, b);
    return G.nnzb;
}

void device_block_values(const Launch &L, const CsrDev &A, BlockGraph &G)
{
    const int bb = G.b * G.b;
    G.val.ensure((size_t)G.nnzb * bb + 4);
    G.didx.ensure((size_t)G.nb + 1);
    (void)bb; // (no memset: block_values_kernel zeroes the rows it does not write in full)
    // 32 lanes per block row: eight rows per workgroup step -- a launch fitted to a small level's vectors (level 1 of
    // configs[2]: 112 workgroups for 114 444 block rows of ~77 blocks) would walk 128 rows per lane group
    const int grid = std::max(L.grid, std::min(8 * L.num_cus, (G.nb + 7) / 8));
    if (G.b == 3)
        hipLaunchKernelGGL(block_values_kernel<3>, dim3(grid), dim3(kBlock), 0, L.stream, G.nb, G.b, A.rowptr, A.col, A.val,
                           G.ptr.ptr, G.col.ptr, G.val.ptr, G.didx.ptr);
    else
        hipLaunchKernelGGL(block_values_kernel<0>, dim3(grid), dim3(kBlock), 0, L.stream, G.nb, G.b, A.rowptr, A.col, A.val,
                           G.ptr.ptr, G.col.ptr, G.val.ptr, G.didx.ptr);
    PS_HIP_CHECK(hipGetLastError());
}

// the strong flags of every block and the per-row counts of the compacted graph, 32 lanes per block row (the flags
// do not depend on each other; a coarse block row has a hundred blocks)
__global__ __launch_bounds__(kBlock) void block_strong_flags_kernel(int nb, int b, const int *__restrict__ bptr,
                                                                     const int *__restrict__ bcol,
                                                                     const double *__restrict__ bval,
                                                                     const int *__restrict__ didx, double eps2,
                                                                     unsigned char *__restrict__ strong,
                                                                     int *__restrict__ cnt)
{
    constexpr int G = 32;
    const int bb = b * b, lane = threadIdx.x % G;
    const int groups = gridDim.x * kBlock / G;
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / G; i < nb; i += groups) {
        // eps = 0 (AMGCL's default, AMGCL.cpp:32-65): "strong" = an off-diagonal block that is not exactly zero; the
        // diagonal blocks of the row and of every column (a 72-byte gather per block) are not needed then
        const double *di = (eps2 != 0.0 && didx[i] >= 0) ? bval + (size_t)didx[i] * bb : nullptr;
        int w = 0;
        for (int j = bptr[i] + lane; j < bptr[i + 1]; j += G) {
            const int c = bcol[j];
            const double *dc = (eps2 != 0.0 && didx[c] >= 0) ? bval + (size_t)didx[c] * bb : nullptr;
            const bool s = block_is_strong(b, i, c, bval + (size_t)j * bb, di, dc, eps2);
            strong[j] = s ? 1 : 0;
            if (s || c == i) ++w;
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) w += __shfl_xor(w, off);
        if (lane == 0) cnt[i] = w;
    }
}

void device_block_strong_flags(const Launch &L, const BlockGraph &G, double eps_strong, unsigned char *flags, int *cnt)
{
    if (G.b == 3 && eps_strong == 0.0 && G.nnzb > 0) { // two phases (round 6): a flag per block streamed, then the rows
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(L.num_cus * 8, (G.nnzb + kBlock - 1) / kBlock));
        hipLaunchKernelGGL(block_stream3_kernel<1>, dim3(grid), dim3(kBlock), 0, L.stream, G.nnzb, G.val.ptr, (double *)nullptr, flags);
        hipLaunchKernelGGL(block_strong_rows_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.ptr.ptr, G.col.ptr, flags, cnt);
        PS_HIP_CHECK(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(block_strong_flags_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.b, G.ptr.ptr, G.col.ptr,
                       G.val.ptr, G.didx.ptr, eps_strong * eps_strong, flags, cnt);
    PS_HIP_CHECK(hipGetLastError());
}

int64_t device_block_strength_graph(const Launch &L, BlockGraph &G, double eps_strong, DeviceBuffer<int> &sptr,
                                    DeviceBuffer<int> &scol, int *id0, SymbolicScratch &S)
{
    G.strong.ensure((size_t)G.nnzb + 4);
    sptr.ensure((size_t)G.nb + 1);
    device_block_strong_flags(L, G, eps_strong, G.strong.ptr, sptr.ptr);
    const int64_t total = device_exclusive_scan(L, sptr.ptr, G.nb, S);
    scol.ensure((size_t)total + 4);
    hipLaunchKernelGGL(block_strength_kernel<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.b, G.ptr.ptr,
                       G.col.ptr, G.val.ptr, G.didx.ptr, eps_strong * eps_strong, G.strong.ptr, sptr.ptr, scol.ptr, id0);
    PS_HIP_CHECK(hipGetLastError());
    return total;
}

int device_block_flag_changes(const Launch &L, int64_t n, const unsigned char *a, const unsigned char *b,
                              SymbolicScratch &S)
{
    S.counters.ensure(16);
    S.host.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(S.counters.ptr, 0, sizeof(int), L.stream));
    hipLaunchKernelGGL(count_flag_changes_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, a, b, S.counters.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.counters.ptr, sizeof(int), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    return *reinterpret_cast<const int *>(S.host.ptr);
}

double device_block_gershgorin(const Launch &L, const BlockGraph &G, double *partials, const int *rows_list, int n_list)
{
    if (G.b == 3 && !(rows_list && n_list > 0) && G.nnzb > 0) { // two phases (round 6): the blocks' norms streamed, then the rows
        G.per_block.ensure((size_t)G.nnzb + 4);
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(L.num_cus * 8, (G.nnzb + kBlock - 1) / kBlock));
        hipLaunchKernelGGL(block_stream3_kernel<0>, dim3(grid), dim3(kBlock), 0, L.stream, G.nnzb, G.val.ptr, G.per_block.ptr,
                           (unsigned char *)nullptr);
        hipLaunchKernelGGL(block_gershgorin3_rows_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.ptr.ptr, G.val.ptr,
                           G.per_block.ptr, G.didx.ptr, partials);
    } else if (G.b == 3)
        hipLaunchKernelGGL(block_gershgorin3_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, (rows_list && n_list > 0) ? n_list : G.nb,
                           G.ptr.ptr, G.val.ptr, G.didx.ptr, partials, (rows_list && n_list > 0) ? rows_list : nullptr);
    else
        hipLaunchKernelGGL(block_gershgorin_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.b, G.ptr.ptr,
                           G.val.ptr, G.didx.ptr, partials);
    PS_HIP_CHECK(hipGetLastError());
    std::vector<double> h((size_t)L.grid);
    PS_HIP_CHECK(hipMemcpyAsync(h.data(), partials, h.size() * sizeof(double), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    double m = 0.0;
    for (double v : h) m = std::max(m, v);
    return m;
}

void launch_block_prolongation_values(const Launch &L, const BlockGraph &G, const int *id, double omega,
                                      const int *pbptr, const int *pbcol, double *pbval)
{
    if (G.b == 3)
        hipLaunchKernelGGL(block_prolongation_values3_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.ptr.ptr,
                           G.col.ptr, G.val.ptr, G.strong.ptr, id, omega, pbptr, pbcol, pbval);
    else
        hipLaunchKernelGGL(block_prolongation_values_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, G.nb, G.b, G.ptr.ptr,
                           G.col.ptr, G.val.ptr, G.strong.ptr, id, omega, pbptr, pbcol, pbval);
    PS_HIP_CHECK(hipGetLastError());
}

// rows stay as they are, every (block) column c becomes the b scalar columns c b .. c b + b - 1
__global__ __launch_bounds__(kBlock) void expand_block_columns_kernel(int n, int b, const int *__restrict__ fptr,
                                                                       const int *__restrict__ fcol, int *__restrict__ ptr,
                                                                       int *__restrict__ col)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i <= n; i += gridDim.x * kBlock) {
        ptr[i] = fptr[i] * b;
        if (i == n) continue;
        for (int k = fptr[i]; k < fptr[i + 1]; ++k)
            for (int c = 0; c < b; ++c) col[(size_t)k * b + c] = fcol[k] * b + c;
    }
}

void launch_expand_block_columns(const Launch &L, int n, int b, const int *fptr, const int *fcol, int *ptr, int *col)
{
    hipLaunchKernelGGL(expand_block_columns_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, b, fptr, fcol, ptr, col);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_expand_block_csr(const Launch &L, int nb, int b, const int *pbptr, const int *pbcol, const double *pbval,
                             int *ptr, int *col, double *val)
{
    if (ptr) hipLaunchKernelGGL(expand_block_ptr_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, b, pbptr, ptr);
    hipLaunchKernelGGL(expand_block_entries_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, b, pbptr, pbcol, pbval,
                       col, val, col != nullptr);
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
