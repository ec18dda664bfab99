// amg_symbolic.hip -- device-side symbolic setup of the smoothed-aggregation hierarchy (see amg_symbolic.hpp).
//
// Row-wise sparse set union is the one primitive behind all three patterns (P, A P, R A P): the
// candidates of row i of C = A * B are the concatenated rows B(k,:), k in A(i,:).  A group of lanes
// (16 / 64 / 256, picked per row from an upper bound on its size) owns one row: candidates go through
// a hash set in LDS (atomicCAS), the distinct keys are compacted and sorted by a bitonic network in
// LDS, and rows too wide for LDS use a hash set in HBM.  Pass 1 counts, a device scan turns counts
// into row pointers, pass 2 fills.  Every row's result is a sorted set, hence independent of the order
// in which lanes and workgroups ran: the patterns are deterministic and equal to the host's.
#include "amg_symbolic.hpp"

#include <algorithm>
#include <climits>

namespace psolve {

namespace {

constexpr int kEmpty = 0x7fffffff;
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;
// per-row bounds of the four row-set tiers (lanes per row / hash slots): 16/128, 64/512, 256/4096, HBM
constexpr int kTier0 = 96, kTier1 = 384, kTier2 = 3072;
// round 4: two more tiers.  Rows of at most kTierTiny candidates take 8 lanes and 64 hash slots (tier value 4) -- the
// prolongation and A P of a stencil-like level 0: half the lanes, clears and scans per row of the 16 / 128 tier; rows between
// kTier0 and kTierMid take 32 lanes and 256 slots (tier value 5) instead of 64 / 512 -- R (A P) of such a level
constexpr int kTierTiny = 40, kTierMid = 192;
constexpr int kSortLds = 8192; // widest row the HBM tier still sorts in LDS

} // namespace
namespace {

template <int GROUP>
__device__ __forceinline__ void group_sync()
{
    if constexpr (GROUP >= kBlock) {
        __syncthreads();
    } else {
        // lanes of one wave: LDS operations complete in issue order, only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- exclusive scan -------------------------------------------------------------------------------
__device__ long long block_exclusive_scan(long long v, long long *sh, long long *total)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const long long add = (t >= off) ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const long long incl = sh[t];
    if (total) *total = sh[kBlock - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(kBlock) void scan_block_sums_kernel(int64_t n, const int *__restrict__ data,
                                                                  long long *__restrict__ bsum)
{
    __shared__ long long sh[kBlock];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    long long s = 0;
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += data[base + k];
    long long total;
    (void)block_exclusive_scan(s, sh, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one workgroup: bsum[0..nb) -> exclusive prefix, bsum[nb] = total
__global__ __launch_bounds__(kBlock) void scan_sums_kernel(int64_t nb, long long *__restrict__ bsum)
{
    __shared__ long long sh[kBlock];
    long long carry = 0;
    for (int64_t base = 0; base < nb; base += kBlock) {
        const int64_t i = base + threadIdx.x;
        const long long v = i < nb ? bsum[i] : 0;
        long long total;
        const long long ex = block_exclusive_scan(v, sh, &total);
        if (i < nb) bsum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}

// in place: data[i] = sum of counts before i, for i in [0, n]
__global__ __launch_bounds__(kBlock) void scan_final_kernel(int64_t n, int *__restrict__ data,
                                                             const long long *__restrict__ bsum)
{
    __shared__ long long sh[kBlock];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems];
    long long s = 0;
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? data[base + k] : 0;
        s += v[k];
    }
    long long run = bsum[blockIdx.x] + block_exclusive_scan(s, sh, nullptr);
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k <= n) data[base + k] = (int)run;
        run += v[k];
    }
}

// ---- strength of connection --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void extract_diag_kernel(int n, const int *__restrict__ rowptr,
                                                               const int *__restrict__ col,
                                                               const double *__restrict__ val, double *__restrict__ dia)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        double d = 0.0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
            if (col[j] == i) {
                d = val[j];
                break;
            }
        dia[i] = d;
    }
}

__device__ __forceinline__ bool keep_entry(int i, int c, double v, double eps_dia_i, const double *dia)
{
    if (c == i) return true;
    const double rhs = v * v;
    // amgcl/coarsening/plain_aggregates.hpp: eps^2 a_ii a_jj < a_ij^2
    return (eps_dia_i != 0.0 ? eps_dia_i * dia[c] : 0.0) < rhs;
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void strength_kernel(int n, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ dia, double eps2,
                                                           int *__restrict__ sptr, int *__restrict__ scol,
                                                           int *__restrict__ id0)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double eps_dia_i = eps2 * dia[i];
        int w = FILL ? sptr[i] : 0;
        bool any = false;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int c = col[j];
            if (keep_entry(i, c, val[j], eps_dia_i, dia)) {
                if (FILL) scol[w] = c;
                ++w;
                any = any || c != i;
            }
        }
        if (!FILL) sptr[i] = w;
        else id0[i] = any ? -1 : -2; // plain_aggregates: undefined / removed (no strong connection at all)
    }
}

// the same with G lanes per row (wide rows: level 1 of the 216^3 hierarchy, 31 entries per row, 1.2 + 1.6 ms with one lane):
// the lanes take G consecutive entries at a time, the kept ones are written in entry order (ballot + popcount below the lane)
template <bool FILL, int G>
__global__ __launch_bounds__(kBlock) void strength_group_kernel(int n, const int *__restrict__ rowptr,
                                                                 const int *__restrict__ col,
                                                                 const double *__restrict__ val,
                                                                 const double *__restrict__ dia, double eps2,
                                                                 int *__restrict__ sptr, int *__restrict__ scol,
                                                                 int *__restrict__ id0)
{
    const int lane = threadIdx.x % G, gbase = (threadIdx.x & 63) / G * G;
    const unsigned long long gmask = (1ull << G) - 1ull;
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / G; i < n; i += gridDim.x * (kBlock / G)) {
        const double eps_dia_i = eps2 * dia[i];
        int w = FILL ? sptr[i] : 0;
        bool any = false;
        const int b = rowptr[i], e = rowptr[i + 1];
        for (int j0 = b; j0 < e; j0 += G) { // (uniform over the group)
            const int j = j0 + lane;
            bool keep = false;
            int c = 0;
            if (j < e) {
                c = col[j];
                keep = keep_entry(i, c, val[j], eps_dia_i, dia);
            }
            const unsigned m = (unsigned)((__ballot(keep) >> gbase) & gmask);
            if (FILL && keep) scol[w + __popc(m & ((1u << lane) - 1u))] = c;
            w += __popc(m);
            any = any || (keep && c != i);
        }
        const bool gany = ((__ballot(any) >> gbase) & gmask) != 0ull;
        if (lane == 0) {
            if (!FILL) sptr[i] = w;
            else id0[i] = gany ? -1 : -2;
        }
    }
}

// rows of A restricted to columns < ncols (a shard's diagonal block: the halo columns are dropped)
template <bool FILL>
__global__ __launch_bounds__(kBlock) void column_filter_kernel(int n, int ncols, const int *__restrict__ rowptr,
                                                                const int *__restrict__ col,
                                                                const double *__restrict__ val,
                                                                int *__restrict__ optr, int *__restrict__ ocol,
                                                                double *__restrict__ oval)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int w = FILL ? optr[i] : 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j)
            if (col[j] < ncols) {
                if (FILL) {
                    ocol[w] = col[j];
                    oval[w] = val[j];
                }
                ++w;
            }
        if (!FILL) optr[i] = w;
    }
}

// ---- row sets ----------------------------------------------------------------------------------------
struct SymArgs {
    int n;
    const int *aptr, *acol;
    const int *bptr, *bcol; // bptr == nullptr: B is a map (row c = {bcol[c]} if >= 0)
    const unsigned char *tier;
    int div; // bptr == bcol == nullptr: row c = {c / div}  (scalar columns -> block columns)
};

// upper bound on |row i of C| -> tier; counters[0..3] rows per tier, [4] widest bound in tier 3,
// list3 = rows of tier 3 (any order)
__global__ __launch_bounds__(kBlock) void rowset_bound_kernel(SymArgs a, int ncols_c, int t3_above, int *__restrict__ ub,
                                                               unsigned char *__restrict__ tier,
                                                               int *__restrict__ counters, int *__restrict__ list3)
{
    const int lane = threadIdx.x & 63;
    const int rounds = (a.n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    int c0 = 0, c1 = 0, c2 = 0, c5 = 0, c6 = 0, mx = 0;
    for (int r = 0; r < rounds; ++r) {
        const int i = (r * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        int t = -1, b = 0;
        if (i < a.n) {
            long long m = 0;
            const int ab = a.aptr[i], ae = a.aptr[i + 1];
            if (a.bptr) {
                for (int j = ab; j < ae; ++j) {
                    const int c = a.acol[j];
                    m += a.bptr[c + 1] - a.bptr[c];
                }
            } else {
                m = ae - ab;
            }
            b = (int)(m < (long long)ncols_c ? m : (long long)ncols_c);
            t = b <= kTierTiny ? 4 : b <= kTier0 ? 0 : b <= kTierMid ? 5 : b <= kTier1 ? 1 : b <= t3_above ? 2 : 3; // (t3_above <= kTier2)
            ub[i] = b;
            tier[i] = (unsigned char)t;
            c0 += t == 0;
            c1 += t == 1;
            c2 += t == 2;
            c5 += t == 4;
            c6 += t == 5;
        }
        const unsigned long long m3 = __ballot(t == 3);
        if (m3) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[3], __popcll(m3));
            base = __shfl(base, 0);
            if (t == 3) {
                list3[base + __popcll(m3 & ((1ull << lane) - 1ull))] = i;
                mx = max(mx, b);
            }
        }
    }
    if (c0) atomicAdd(&counters[0], c0);
    if (c1) atomicAdd(&counters[1], c1);
    if (c2) atomicAdd(&counters[2], c2);
    if (c5) atomicAdd(&counters[5], c5);
    if (c6) atomicAdd(&counters[6], c6);
    if (mx) atomicMax(&counters[4], mx);
}

// returns 1 when `key` was not in the set yet
template <bool GLOBAL>
__device__ __forceinline__ int set_insert(int *tab, unsigned mask, int shift, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) >> shift;
    for (;;) {
        int cur;
        if (GLOBAL) cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else cur = *(volatile int *)&tab[h];
        if (cur == key) return 0;
        if (cur == kEmpty) {
            const int old = atomicCAS(&tab[h], kEmpty, key);
            if (old == kEmpty) return 1;
            if (old == key) return 0;
        }
        h = (h + 1) & mask;
    }
}

template <int GROUP, bool GLOBAL>
__device__ __forceinline__ int insert_row_candidates(const SymArgs &a, int i, int lane, int *tab, unsigned mask,
                                                     int shift)
{
    int added = 0;
    const int ab = a.aptr[i], ae = a.aptr[i + 1];
    for (int ja = ab + lane; ja < ae; ja += GROUP) {
        const int c = a.acol[ja];
        if (a.bptr) {
            const int be = a.bptr[c + 1];
            for (int jb = a.bptr[c]; jb < be; ++jb) added += set_insert<GLOBAL>(tab, mask, shift, a.bcol[jb]);
        } else if (a.bcol) {
            const int v = a.bcol[c];
            if (v >= 0) added += set_insert<GLOBAL>(tab, mask, shift, v);
        } else {
            added += set_insert<GLOBAL>(tab, mask, shift, c / a.div);
        }
    }
    return added;
}

// ascending bitonic sort of a[0..p2), p2 a power of two, by GROUP lanes
template <int GROUP>
__device__ __forceinline__ void group_bitonic(int *a, int p2, int lane)
{
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (p2 >> 1); t += GROUP) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const int x = a[lo], y = a[hi];
                if ((x > y) == up) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
            group_sync<GROUP>();
        }
}

__device__ __forceinline__ int next_pow2(int u)
{
    int p = 1;
    while (p < u) p <<= 1;
    return p;
}

template <int H>
struct Log2 {
    static constexpr int value = 1 + Log2<H / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr int value = 0;
};

template <int GROUP, int H, int TIER, bool FILL>
__global__ __launch_bounds__(kBlock) void rowset_lds_kernel(SymArgs a, int *__restrict__ cnt,
                                                             const int *__restrict__ cptr, int *__restrict__ ccol)
{
    constexpr int GPB = kBlock / GROUP;
    constexpr int SHIFT = 32 - Log2<H>::value;
    __shared__ int tab[GPB * H];
    __shared__ int lst[FILL ? GPB * H : 1];
    __shared__ int lcount[GPB];
    const int g = threadIdx.x / GROUP, lane = threadIdx.x % GROUP;
    int *mytab = tab + g * H;
    int *mylst = lst + (FILL ? g * H : 0);
    for (int i = blockIdx.x * GPB + g; i < a.n; i += gridDim.x * GPB) {
        if (a.tier[i] != TIER) continue; // uniform over the group (and over the workgroup when GROUP == 256)
        for (int s = lane; s < H; s += GROUP) mytab[s] = kEmpty;
        if (lane == 0) lcount[g] = 0;
        group_sync<GROUP>();
        const int added = insert_row_candidates<GROUP, false>(a, i, lane, mytab, H - 1, SHIFT);
        if (!FILL) {
            if (added) atomicAdd(&lcount[g], added);
            group_sync<GROUP>();
            if (lane == 0) cnt[i] = lcount[g];
            group_sync<GROUP>();
        } else {
            group_sync<GROUP>();
            for (int s = lane; s < H; s += GROUP) {
                const int k = mytab[s];
                if (k != kEmpty) mylst[atomicAdd(&lcount[g], 1)] = k;
            }
            group_sync<GROUP>();
            const int u = lcount[g];
            const int p2 = next_pow2(u);
            for (int s = u + lane; s < p2; s += GROUP) mylst[s] = kEmpty;
            group_sync<GROUP>();
            group_bitonic<GROUP>(mylst, p2, lane);
            const int cb = cptr[i];
            for (int s = lane; s < u; s += GROUP) ccol[cb + s] = mylst[s];
            group_sync<GROUP>();
        }
    }
}

// rows too wide for LDS: one workgroup per row, hash set of pow2 >= 2 * bound slots in HBM
template <bool FILL>
__global__ __launch_bounds__(kBlock) void rowset_global_kernel(SymArgs a, int nlist, const int *__restrict__ list,
                                                                const int *__restrict__ ub, int *scratch,
                                                                long long stride, int *__restrict__ cnt,
                                                                const int *__restrict__ cptr, int *__restrict__ ccol)
{
    __shared__ int lst[FILL ? kSortLds : 1];
    __shared__ int lcount;
    int *tab = scratch + (long long)blockIdx.x * stride;
    for (int r = blockIdx.x; r < nlist; r += gridDim.x) {
        const int i = list[r];
        int logt = 13;
        while (logt < 30 && (1ll << logt) < 2ll * ub[i]) ++logt;
        const int ts = 1 << logt;
        for (int s = threadIdx.x; s < ts; s += kBlock) tab[s] = kEmpty;
        if (threadIdx.x == 0) lcount = 0;
        __threadfence();
        __syncthreads();
        const int added = insert_row_candidates<kBlock, true>(a, i, threadIdx.x, tab, (unsigned)ts - 1u, 32 - logt);
        if (!FILL) {
            if (added) atomicAdd(&lcount, added);
            __syncthreads();
            if (threadIdx.x == 0) cnt[i] = lcount;
            __syncthreads();
            continue;
        }
        __threadfence();
        __syncthreads();
        const int cb = cptr[i], u = cptr[i + 1] - cb;
        if (u <= kSortLds) {
            for (int s = threadIdx.x; s < ts; s += kBlock) {
                const int k = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k != kEmpty) lst[atomicAdd(&lcount, 1)] = k;
            }
            __syncthreads();
            const int p2 = next_pow2(u);
            for (int s = u + threadIdx.x; s < p2; s += kBlock) lst[s] = kEmpty;
            __syncthreads();
            group_bitonic<kBlock>(lst, p2, threadIdx.x);
            for (int s = threadIdx.x; s < u; s += kBlock) ccol[cb + s] = lst[s];
        } else {
            // correctness path for enormous rows: sort the whole table in HBM (empties sort last)
            group_bitonic<kBlock>(tab, ts, threadIdx.x);
            for (int s = threadIdx.x; s < u; s += kBlock) ccol[cb + s] = tab[s];
        }
        __syncthreads();
    }
}

// wide rows of a product with FEW columns (R (A P) onto a coarse level: a row of R_1 of the 256^3 hierarchy unions 45 rows
// of ~100 entries into a few hundred distinct columns out of 44 545): the set is a bitmap of the columns in LDS.  One
// workgroup per row; a WAVE takes an entry of A's row and its lanes stride the row of B (the hash tiers give an entry of
// A to a lane, which leaves most of 256 lanes idle on such rows); counting is a popcount, and the bits come out in
// ascending order, so there is nothing to sort.  Same sorted set as the hash tiers.
constexpr int kBitmapWords = 15360; // 60 KiB of LDS: products with up to 491 520 columns

template <bool FILL>
__global__ __launch_bounds__(kBlock) void rowset_bitmap_kernel(SymArgs a, int nlist, const int *__restrict__ list,
                                                                int ncols_c, int *__restrict__ cnt,
                                                                const int *__restrict__ cptr, int *__restrict__ ccol)
{
    extern __shared__ unsigned bits[];
    __shared__ int wsum[kBlock / 64];
    const int nw = (ncols_c + 31) >> 5;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int chunk = (nw + kBlock - 1) / kBlock; // words per thread, contiguous: thread t emits before thread t + 1
    for (int r = blockIdx.x; r < nlist; r += gridDim.x) {
        const int i = list[r];
        for (int w = tid; w < nw; w += kBlock) bits[w] = 0u;
        __syncthreads();
        const int ab = a.aptr[i], ae = a.aptr[i + 1];
        if (a.bptr) {
            for (int ja = ab + wave; ja < ae; ja += kBlock / 64) {
                const int c = a.acol[ja];
                const int be = a.bptr[c + 1];
                for (int jb = a.bptr[c] + lane; jb < be; jb += 64) {
                    const int v = a.bcol[jb];
                    atomicOr(&bits[v >> 5], 1u << (v & 31));
                }
            }
        } else {
            for (int ja = ab + tid; ja < ae; ja += kBlock) {
                const int c = a.acol[ja];
                const int v = a.bcol ? a.bcol[c] : c / a.div;
                if (v >= 0) atomicOr(&bits[v >> 5], 1u << (v & 31));
            }
        }
        __syncthreads();
        const int w0 = min(tid * chunk, nw), w1 = min(w0 + chunk, nw);
        int mine = 0;
        for (int w = w0; w < w1; ++w) mine += __popc(bits[w]);
        // exclusive scan of `mine` over the workgroup: inside the wave by shuffles, across the four waves through LDS
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = incl - mine, total = 0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) {
            if (k < wave) before += wsum[k];
            total += wsum[k];
        }
        if (!FILL) {
            if (tid == 0) cnt[i] = total;
        } else {
            int o = cptr[i] + before;
            for (int w = w0; w < w1; ++w) {
                unsigned m = bits[w];
                while (m) {
                    const int b = __ffs(m) - 1;
                    ccol[o++] = (w << 5) + b;
                    m &= m - 1u;
                }
            }
        }
        __syncthreads(); // (bits and wsum are reused by the next row)
    }
}

// ---- transpose ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void col_histogram_kernel(int64_t nnz, const int *__restrict__ col,
                                                                int *__restrict__ cnt)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock)
        atomicAdd(&cnt[col[e]], 1);
}

__global__ __launch_bounds__(kBlock) void col_scatter_kernel(int64_t nnz, const int *__restrict__ col,
                                                              const int *__restrict__ rptr, int *__restrict__ cursor,
                                                              int *__restrict__ tmp)
{
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kBlock) {
        const int c = col[e];
        tmp[rptr[c] + atomicAdd(&cursor[c], 1)] = (int)e;
    }
}

// counters[0] = longest row, counters[1] = rows longer than `big`, list = those rows
__global__ __launch_bounds__(kBlock) void rowlen_stats_kernel(int nrows, const int *__restrict__ rptr, int big,
                                                               int *__restrict__ counters, int *__restrict__ list)
{
    int mx = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nrows; i += gridDim.x * kBlock) {
        const int len = rptr[i + 1] - rptr[i];
        mx = max(mx, len);
        if (len > big) list[atomicAdd(&counters[1], 1)] = i;
    }
    if (mx) atomicMax(&counters[0], mx);
}

__device__ __forceinline__ int row_of_entry(const int *__restrict__ pptr, int n, int e)
{
    int lo = 0, hi = n; // largest i with pptr[i] <= e
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1); // lo + hi can pass 2^31
        if (pptr[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// sorts the entry ids of every R row whose length is in (lo_len, hi_len] and writes rcol / r_from_p
template <int GROUP, int CAP>
__global__ __launch_bounds__(kBlock) void rowsort_lds_kernel(int nrows, const int *__restrict__ rptr, int lo_len,
                                                              int hi_len, const int *__restrict__ tmp,
                                                              const int *__restrict__ pptr, int n,
                                                              int *__restrict__ rcol, int *__restrict__ r_from_p)
{
    constexpr int GPB = kBlock / GROUP;
    __shared__ int lst[GPB * CAP];
    const int g = threadIdx.x / GROUP, lane = threadIdx.x % GROUP;
    int *mylst = lst + g * CAP;
    for (int i = blockIdx.x * GPB + g; i < nrows; i += gridDim.x * GPB) {
        const int rb = rptr[i], u = rptr[i + 1] - rb;
        if (u <= lo_len || u > hi_len) continue;
        const int p2 = next_pow2(u);
        for (int s = lane; s < p2; s += GROUP) mylst[s] = s < u ? tmp[rb + s] : kEmpty;
        group_sync<GROUP>();
        group_bitonic<GROUP>(mylst, p2, lane);
        for (int s = lane; s < u; s += GROUP) {
            const int e = mylst[s];
            r_from_p[rb + s] = e;
            rcol[rb + s] = row_of_entry(pptr, n, e);
        }
        group_sync<GROUP>();
    }
}

__global__ __launch_bounds__(kBlock) void rowsort_global_kernel(int nlist, const int *__restrict__ list,
                                                                 const int *__restrict__ rptr,
                                                                 const int *__restrict__ tmp, int *scratch,
                                                                 long long stride, const int *__restrict__ pptr, int n,
                                                                 int *__restrict__ rcol, int *__restrict__ r_from_p)
{
    int *buf = scratch + (long long)blockIdx.x * stride;
    for (int r = blockIdx.x; r < nlist; r += gridDim.x) {
        const int i = list[r];
        const int rb = rptr[i], u = rptr[i + 1] - rb;
        const int p2 = next_pow2(u);
        for (int s = threadIdx.x; s < p2; s += kBlock) buf[s] = s < u ? tmp[rb + s] : kEmpty;
        __syncthreads();
        group_bitonic<kBlock>(buf, p2, threadIdx.x);
        for (int s = threadIdx.x; s < u; s += kBlock) {
            const int e = buf[s];
            r_from_p[rb + s] = e;
            rcol[rb + s] = row_of_entry(pptr, n, e);
        }
        __syncthreads();
    }
}

void read_counters(const Launch &L, SymbolicScratch &S, int n, int *out)
{
    S.host.ensure(16);
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.counters.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    const int *h = reinterpret_cast<const int *>(S.host.ptr);
    for (int k = 0; k < n; ++k) out[k] = h[k];
}

} // namespace

int64_t device_exclusive_scan(const Launch &L, int *data, int64_t n, SymbolicScratch &S)
{
    const int64_t nb = (n + 1 + kScanTile - 1) / kScanTile;
    S.bsum.ensure((size_t)nb + 1);
    S.host.ensure(16);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(kBlock), 0, L.stream, n, data, S.bsum.ptr);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, L.stream, nb, S.bsum.ptr);
    hipLaunchKernelGGL(scan_final_kernel, dim3((unsigned)nb), dim3(kBlock), 0, L.stream, n, data, S.bsum.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.bsum.ptr + nb, sizeof(long long), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    const long long total = S.host.ptr[0];
    PS_REQUIRE(total < (long long)INT32_MAX, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
    return (int64_t)total;
}

void launch_extract_diagonal(const Launch &L, const CsrDev &A, double *dia)
{
    hipLaunchKernelGGL(extract_diag_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val, dia);
    PS_HIP_CHECK(hipGetLastError());
}

int64_t device_diagonal_block(const Launch &L, const CsrDev &A, DeviceBuffer<int> &ptr, DeviceBuffer<int> &col,
                              DeviceBuffer<double> &val, SymbolicScratch &S)
{
    ptr.ensure((size_t)A.n + 1);
    hipLaunchKernelGGL(column_filter_kernel<false>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.n, A.rowptr, A.col,
                       A.val, ptr.ptr, (int *)nullptr, (double *)nullptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t total = device_exclusive_scan(L, ptr.ptr, A.n, S);
    col.ensure((size_t)total + 4);
    val.ensure((size_t)total + 4);
    hipLaunchKernelGGL(column_filter_kernel<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.n, A.rowptr, A.col,
                       A.val, ptr.ptr, col.ptr, val.ptr);
    PS_HIP_CHECK(hipGetLastError());
    return total;
}

int64_t device_strength_graph(const Launch &L, const CsrDev &A, double eps_strong, const double *dia,
                              DeviceBuffer<int> &sptr, DeviceBuffer<int> &scol, int *id0, SymbolicScratch &S)
{
    const double eps2 = eps_strong * eps_strong;
    sptr.ensure((size_t)A.n + 1);
    const bool wide = A.nnz > 12ll * (int64_t)A.n; // several lanes per row
    if (wide)
        hipLaunchKernelGGL((strength_group_kernel<false, 16>), dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col,
                           A.val, dia, eps2, sptr.ptr, (int *)nullptr, (int *)nullptr);
    else
        hipLaunchKernelGGL(strength_kernel<false>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           dia, eps2, sptr.ptr, (int *)nullptr, (int *)nullptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t total = device_exclusive_scan(L, sptr.ptr, A.n, S);
    scol.ensure((size_t)total + 4);
    if (wide)
        hipLaunchKernelGGL((strength_group_kernel<true, 16>), dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col,
                           A.val, dia, eps2, sptr.ptr, scol.ptr, id0);
    else
        hipLaunchKernelGGL(strength_kernel<true>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           dia, eps2, sptr.ptr, scol.ptr, id0);
    PS_HIP_CHECK(hipGetLastError());
    return total;
}

int64_t device_spgemm_symbolic(const Launch &L, int n, const int *aptr, const int *acol, const int *bptr,
                               const int *bcol, int ncols_c, DeviceBuffer<int> &cptr, DeviceBuffer<int> &ccol,
                               SymbolicScratch &S, int div)
{
    hipStream_t s = L.stream;
    S.cand.ensure((size_t)n + 1);
    S.tier.ensure((size_t)n + 1);
    S.tmp.ensure((size_t)n + 1);
    S.counters.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(S.counters.ptr, 0, 16 * sizeof(int), s));
    SymArgs a{n, aptr, acol, bptr, bcol, S.tier.ptr, div > 0 ? div : 1};
    // products with few columns: the rows the 64-lane hash tier does not take go to the bitmap kernel
    const bool bitmap = L.lab.symbolic_bitmap && ncols_c <= 32 * kBitmapWords;
    hipLaunchKernelGGL(rowset_bound_kernel, dim3(L.grid), dim3(kBlock), 0, s, a, ncols_c, bitmap ? kTier1 : kTier2, S.cand.ptr,
                       S.tier.ptr, S.counters.ptr, S.tmp.ptr);
    PS_HIP_CHECK(hipGetLastError());
    int c[7];
    read_counters(L, S, 7, c);
    int grid3 = 0;
    long long stride3 = 0;
    const size_t bitmap_lds = (size_t)((ncols_c + 31) / 32) * sizeof(unsigned);
    if (c[3] > 0 && bitmap) {
        grid3 = std::max(1, std::min(c[3], L.num_cus * (int)std::max<size_t>(1, std::min<size_t>(8, (128 * 1024) / std::max<size_t>(bitmap_lds, 1)))));
    } else if (c[3] > 0) {
        stride3 = 1 << 13;
        while (stride3 < 2ll * c[4]) stride3 <<= 1;
        const long long budget = 1ll << 28; // ints (1 GiB)
        grid3 = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(c[3], 1024), budget / stride3));
        S.table.ensure((size_t)(stride3 * grid3));
    }
    cptr.ensure((size_t)n + 1);
    const dim3 g(L.grid), blk(kBlock);
#define PS_ROWSET(FILL, CNT, CPTR, CCOL)                                                                            \
    do {                                                                                                            \
        if (c[5]) hipLaunchKernelGGL((rowset_lds_kernel<8, 64, 4, FILL>), g, blk, 0, s, a, CNT, CPTR, CCOL);        \
        if (c[0]) hipLaunchKernelGGL((rowset_lds_kernel<16, 128, 0, FILL>), g, blk, 0, s, a, CNT, CPTR, CCOL);      \
        if (c[6]) hipLaunchKernelGGL((rowset_lds_kernel<32, 256, 5, FILL>), g, blk, 0, s, a, CNT, CPTR, CCOL);      \
        if (c[1]) hipLaunchKernelGGL((rowset_lds_kernel<64, 512, 1, FILL>), g, blk, 0, s, a, CNT, CPTR, CCOL);      \
        if (c[2]) hipLaunchKernelGGL((rowset_lds_kernel<256, 4096, 2, FILL>), g, blk, 0, s, a, CNT, CPTR, CCOL);    \
        if (c[3] && bitmap)                                                                                         \
            hipLaunchKernelGGL((rowset_bitmap_kernel<FILL>), dim3(grid3), blk, bitmap_lds, s, a, c[3], S.tmp.ptr, ncols_c, \
                               CNT, CPTR, CCOL);                                                                    \
        else if (c[3])                                                                                              \
            hipLaunchKernelGGL((rowset_global_kernel<FILL>), dim3(grid3), blk, 0, s, a, c[3], S.tmp.ptr, S.cand.ptr, \
                               S.table.ptr, stride3, CNT, CPTR, CCOL);                                              \
        PS_HIP_CHECK(hipGetLastError());                                                                            \
    } while (0)
    PS_ROWSET(false, cptr.ptr, (const int *)nullptr, (int *)nullptr);
    const int64_t total = device_exclusive_scan(L, cptr.ptr, n, S);
    ccol.ensure((size_t)total + 4);
    PS_ROWSET(true, (int *)nullptr, cptr.ptr, ccol.ptr);
#undef PS_ROWSET
    return total;
}

void device_transpose_pattern(const Launch &L, int n, int ncols, const int *pptr, const int *pcol, int64_t nnz,
                              DeviceBuffer<int> &rptr, DeviceBuffer<int> &rcol, DeviceBuffer<int> &r_from_p,
                              SymbolicScratch &S)
{
    hipStream_t s = L.stream;
    rptr.ensure((size_t)ncols + 1);
    rcol.ensure((size_t)nnz + 4);
    r_from_p.ensure((size_t)nnz + 4);
    S.tmp.ensure((size_t)std::max<int64_t>(nnz, ncols) + 4);
    S.cursor.ensure((size_t)ncols + 1);
    S.cand.ensure((size_t)ncols + 1); // list of long rows
    S.counters.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(rptr.ptr, 0, ((size_t)ncols + 1) * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(S.cursor.ptr, 0, ((size_t)ncols + 1) * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(S.counters.ptr, 0, 16 * sizeof(int), s));
    const dim3 g(L.grid), blk(kBlock);
    hipLaunchKernelGGL(col_histogram_kernel, g, blk, 0, s, nnz, pcol, rptr.ptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t total = device_exclusive_scan(L, rptr.ptr, ncols, S);
    PS_REQUIRE(total == nnz, PSOLVE_HIP_EINVAL, "transpose: column index out of range");
    hipLaunchKernelGGL(col_scatter_kernel, g, blk, 0, s, nnz, pcol, rptr.ptr, S.cursor.ptr, S.tmp.ptr);
    constexpr int kBig = 4096;
    hipLaunchKernelGGL(rowlen_stats_kernel, g, blk, 0, s, ncols, rptr.ptr, kBig, S.counters.ptr, S.cand.ptr);
    PS_HIP_CHECK(hipGetLastError());
    int c[2];
    read_counters(L, S, 2, c);
    hipLaunchKernelGGL((rowsort_lds_kernel<16, 128>), g, blk, 0, s, ncols, rptr.ptr, 0, 128, S.tmp.ptr, pptr, n,
                       rcol.ptr, r_from_p.ptr);
    if (c[0] > 128)
        hipLaunchKernelGGL((rowsort_lds_kernel<64, 512>), g, blk, 0, s, ncols, rptr.ptr, 128, 512, S.tmp.ptr, pptr, n,
                           rcol.ptr, r_from_p.ptr);
    if (c[0] > 512)
        hipLaunchKernelGGL((rowsort_lds_kernel<256, kBig>), g, blk, 0, s, ncols, rptr.ptr, 512, kBig, S.tmp.ptr, pptr,
                           n, rcol.ptr, r_from_p.ptr);
    if (c[1] > 0) {
        long long stride = 1;
        while (stride < c[0]) stride <<= 1;
        const long long budget = 1ll << 28;
        const int grid = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(c[1], 256), budget / stride));
        S.table.ensure((size_t)(stride * grid));
        hipLaunchKernelGGL(rowsort_global_kernel, dim3(grid), blk, 0, s, c[1], S.cand.ptr, rptr.ptr, S.tmp.ptr,
                           S.table.ptr, stride, pptr, n, rcol.ptr, r_from_p.ptr);
    }
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
