// reorder.hip -- Cuthill-McKee renumbering on the device, level by level (see reorder.hpp for the definition and the
// reference precedent: MASSolver renumbers the system by a graph partition before it builds its preconditioner,
// /root/reference/src/polysolve/linear/mas_utils/GraphPartition.cpp:240-243).
//
// The sequential definition -- a dequeued vertex appends its unvisited neighbours in row order -- has a closed form per
// breadth-first level: a vertex of the next level belongs to the frontier vertex of the SMALLEST POSITION adjacent to
// it, the children of one parent keep the parent's row order, and parents are taken in position order.  So a level
// is four launches over the frontier order[lo, hi):
//   claim  every frontier position i lowers claim[w] to i for its unvisited neighbours w            (atomicMin)
//   count  position i counts the neighbours it won (claim[w] == i), marks them ~i; tile sums
//   scan   one workgroup: exclusive scan of the tile sums, next frontier = [hi, hi + total)
//   write  position i writes its children, in row order, at hi + (scan of the counts)
// [lo, hi) lives in device memory (double-buffered by level parity), so the host enqueues levels blindly in batches
// and looks at the state once per batch; levels behind the last one of a component are no-ops.
// What a level costs (~35 us; 766 levels of the 256^3 grid: 27-32 ms) is not the number of launches but the chains of
// dependent loads inside them (state -> order -> row pointer -> column -> position -> atomic: ~10 us per kernel whatever
// the frontier's size).  Tried and dropped: the scan folded into the count kernel by a ticket (the last workgroup to
// finish scans: three launches, 29-34 ms: no gain); write + the next level's claims in one launch (two launches per
// level, but one thread then walks the rows of all its children: 35-66 ms).
#include <algorithm>
#include <climits>

#include "reorder.hpp"

namespace psolve {

namespace {

static_assert(kBlock == 256, "the block scans below assume four waves of 64");

// st: [0..1] lo by parity, [2..3] hi by parity, [4] levels with a non-empty frontier, [5] the widest frontier so far
enum { ST_LO = 0, ST_HI = 2, ST_LEVELS = 4, ST_WIDEST = 5, ST_COUNT = 8 };

__device__ __forceinline__ int block_exclusive_scan(int v, int *sh, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int k = 0; k < kBlock / 64; ++k) {
        const int s = sh[k];
        if (k < w) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + x - v;
}

// pos = -1, claim = "nobody", flag[i] = 1 for rows without an off-diagonal entry
__global__ __launch_bounds__(kBlock) void cm_init_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ col,
                                                         int *__restrict__ pos, int *__restrict__ claim,
                                                         int *__restrict__ flag)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        bool iso = true;
        for (int k = ptr[i], e = ptr[i + 1]; k < e; ++k)
            if (col[k] != i) {
                iso = false;
                break;
            }
        pos[i] = -1;
        claim[i] = INT_MAX;
        flag[i] = iso ? 1 : 0;
    }
}

// off = exclusive scan of the flags (n + 1 entries): flagged vertices take positions base + off[i], ascending
__global__ __launch_bounds__(kBlock) void cm_place_flagged_kernel(int n, const int *__restrict__ off, int base,
                                                                  int *__restrict__ order, int *__restrict__ pos)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        if (off[i + 1] != off[i]) {
            const int p = base + off[i];
            order[p] = i;
            pos[i] = p;
        }
}

__global__ __launch_bounds__(kBlock) void cm_flag_unvisited_kernel(int n, const int *__restrict__ pos, int *__restrict__ flag)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) flag[i] = pos[i] < 0 ? 1 : 0;
}

// key = min over the unvisited vertices of (stored entries of the row, index)
__global__ __launch_bounds__(kBlock) void cm_min_key_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ pos,
                                                            unsigned long long *key)
{
    unsigned long long best = ~0ull;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        if (pos[i] < 0) {
            const unsigned long long k = ((unsigned long long)(unsigned)(ptr[i + 1] - ptr[i]) << 32) | (unsigned)i;
            best = k < best ? k : best;
        }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long y = __shfl_xor(best, o, 64);
        best = y < best ? y : best;
    }
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(key, best);
}

__global__ void cm_place_start_kernel(const unsigned long long *key, int placed, int *order, int *pos, int *st)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int s = (int)(*key & 0xffffffffull);
        order[placed] = s;
        pos[s] = placed;
        st[ST_LO + 0] = placed;
        st[ST_HI + 0] = placed + 1;
        st[ST_LO + 1] = placed + 1;
        st[ST_HI + 1] = placed + 1;
    }
}

__global__ __launch_bounds__(kBlock) void cm_claim_kernel(int par, const int *__restrict__ st, const int *__restrict__ order,
                                                          const int *__restrict__ ptr, const int *__restrict__ col,
                                                          const int *__restrict__ pos, int *__restrict__ claim)
{
    const int lo = st[ST_LO + par], hi = st[ST_HI + par];
    for (int i = lo + blockIdx.x * kBlock + threadIdx.x; i < hi; i += gridDim.x * kBlock) {
        const int v = order[i];
        for (int k = ptr[v], e = ptr[v + 1]; k < e; ++k) {
            const int w = col[k];
            if (pos[w] < 0) atomicMin(&claim[w], i);
        }
    }
}

__global__ __launch_bounds__(kBlock) void cm_count_kernel(int par, const int *__restrict__ st, const int *__restrict__ order,
                                                          const int *__restrict__ ptr, const int *__restrict__ col,
                                                          int *__restrict__ claim, int *__restrict__ cnt,
                                                          int *__restrict__ tsum)
{
    __shared__ int sh[kBlock / 64];
    const int lo = st[ST_LO + par], hi = st[ST_HI + par];
    const int tiles = (hi - lo + kBlock - 1) / kBlock;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int i = lo + t * kBlock + threadIdx.x;
        int c = 0;
        if (i < hi) {
            const int v = order[i];
            for (int k = ptr[v], e = ptr[v + 1]; k < e; ++k) {
                const int w = col[k];
                if (claim[w] == i) { // won by this position; marked so that a repeated column is not counted twice
                    claim[w] = ~i;
                    ++c;
                }
            }
            cnt[i - lo] = c;
        }
        int total;
        (void)block_exclusive_scan(c, sh, total);
        if (threadIdx.x == 0) tsum[t] = total;
    }
}

// one workgroup: tsum <- its exclusive scan; the next level's frontier
__global__ __launch_bounds__(kBlock) void cm_scan_tiles_kernel(int par, int *__restrict__ st, int *__restrict__ tsum)
{
    __shared__ int sh[kBlock / 64];
    const int lo = st[ST_LO + par], hi = st[ST_HI + par];
    const int tiles = (hi - lo + kBlock - 1) / kBlock;
    const int per = (tiles + kBlock - 1) / kBlock;
    const int b = min(tiles, (int)threadIdx.x * per), e = min(tiles, b + per);
    int s = 0;
    for (int k = b; k < e; ++k) s += tsum[k];
    int total;
    int run = block_exclusive_scan(s, sh, total);
    for (int k = b; k < e; ++k) {
        const int v = tsum[k];
        tsum[k] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        st[ST_LO + (par ^ 1)] = hi;
        st[ST_HI + (par ^ 1)] = hi + total;
        if (hi > lo) st[ST_LEVELS] += 1;
        if (total > st[ST_WIDEST]) st[ST_WIDEST] = total;
    }
}

__global__ __launch_bounds__(kBlock) void cm_write_kernel(int par, const int *__restrict__ st, const int *__restrict__ ptr,
                                                          const int *__restrict__ col, const int *__restrict__ claim,
                                                          const int *__restrict__ cnt, const int *__restrict__ tsum,
                                                          int *__restrict__ order, int *__restrict__ pos)
{
    __shared__ int sh[kBlock / 64];
    const int lo = st[ST_LO + par], hi = st[ST_HI + par];
    const int tiles = (hi - lo + kBlock - 1) / kBlock;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int i = lo + t * kBlock + threadIdx.x;
        const int c = i < hi ? cnt[i - lo] : 0;
        int total;
        int p = hi + tsum[t] + block_exclusive_scan(c, sh, total);
        if (c > 0) {
            const int v = order[i];
            for (int k = ptr[v], e = ptr[v + 1]; k < e; ++k) {
                const int w = col[k];
                if (claim[w] == ~i && pos[w] < 0) {
                    pos[w] = p;
                    order[p] = w;
                    ++p;
                }
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void expand_node_order_kernel(int nb, int b, const int *__restrict__ order,
                                                                   int *__restrict__ dof_order,
                                                                   int *__restrict__ dof_new_of_old)
{
    const long long n = (long long)nb * b;
    for (long long j = (long long)blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long)gridDim.x * kBlock) {
        const int k = (int)(j / b), c = (int)(j % b);
        const int old = order[k] * b + c;
        dof_order[j] = old;
        dof_new_of_old[old] = (int)j;
    }
}

// the order read backwards ("reverse" Cuthill-McKee): order[k] <-> order[n - 1 - k], new_of_old[v] = n - 1 - new_of_old[v]
__global__ __launch_bounds__(kBlock) void reverse_order_kernel(int n, int *__restrict__ order, int *__restrict__ new_of_old)
{
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        if (k < n / 2) {
            const int a = order[k], b = order[n - 1 - k];
            order[k] = b;
            order[n - 1 - k] = a;
        }
        new_of_old[k] = n - 1 - new_of_old[k];
    }
}

// one workgroup per sampled group of 64 rows: among the first 64 entries of each row, the distinct gathered unknowns
// (nodes, with block value types: col / b) and the distinct lines of eight consecutive unknowns they fall into
constexpr int kSpreadSlots = 8192;
__global__ __launch_bounds__(kBlock) void gather_spread_kernel(int n, const int *__restrict__ ptr, const int *__restrict__ col,
                                                               int b, int stride, unsigned long long *acc)
{
    __shared__ int table[kSpreadSlots];
    __shared__ int counts[2];
    const int groups = (n + 63) / 64;
    for (int g = blockIdx.x * stride; g < groups; g += gridDim.x * stride) {
        if (threadIdx.x < 2) counts[threadIdx.x] = 0;
        const int row = g * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
        for (int pass = 0; pass < 2; ++pass) { // 0: unknowns, 1: lines
            for (int s = threadIdx.x; s < kSpreadSlots; s += kBlock) table[s] = -1;
            __syncthreads();
            int distinct = 0;
            if (row < n) {
                const int rb = ptr[row], len = min(ptr[row + 1] - rb, 64);
                for (int k = sub; k < len; k += 4) {
                    const int key = pass == 0 ? col[rb + k] / b : (col[rb + k] / b) >> 3;
                    unsigned h = ((unsigned)key * 2654435761u) & (kSpreadSlots - 1);
                    for (;;) {
                        const int old = atomicCAS(&table[h], -1, key);
                        if (old == -1) {
                            ++distinct;
                            break;
                        }
                        if (old == key) break;
                        h = (h + 1) & (kSpreadSlots - 1);
                    }
                }
            }
            if (distinct) atomicAdd(&counts[pass], distinct);
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            atomicAdd(&acc[0], (unsigned long long)((counts[0] + 7) / 8));
            atomicAdd(&acc[1], (unsigned long long)counts[1]);
        }
        __syncthreads();
    }
}

} // namespace

void launch_expand_node_order(const Launch &L, int nb, int b, const int *order, int *dof_order, int *dof_new_of_old)
{
    hipLaunchKernelGGL(expand_node_order_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, b, order, dof_order,
                       dof_new_of_old);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_reverse_order(const Launch &L, int n, int *order, int *new_of_old)
{
    hipLaunchKernelGGL(reverse_order_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, order, new_of_old);
    PS_HIP_CHECK(hipGetLastError());
}

double device_gather_spread(const Launch &L, int n, const int *ptr, const int *col, int b, int stride, SymbolicScratch &S)
{
    b = std::max(1, b);
    S.bsum.ensure(4);
    S.host.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(S.bsum.ptr, 0, 2 * sizeof(long long), L.stream));
    const int groups = (n + 63) / 64;
    stride = std::max(1, stride);
    const int grid = std::max(1, std::min(L.grid, (groups + stride - 1) / stride));
    hipLaunchKernelGGL(gather_spread_kernel, dim3(grid), dim3(kBlock), 0, L.stream, n, ptr, col, b, stride,
                       reinterpret_cast<unsigned long long *>(S.bsum.ptr));
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.bsum.ptr, 2 * sizeof(long long), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    const double ideal = (double)S.host.ptr[0], lines = (double)S.host.ptr[1];
    return ideal > 0.0 ? lines / ideal : 1.0;
}

void device_cuthill_mckee(const Launch &L, int n, const int *ptr, const int *col, int *order, int *new_of_old,
                          ReorderScratch &W, SymbolicScratch &S, ReorderInfo *info)
{
    hipStream_t s = L.stream;
    int *pos = new_of_old;
    W.claim.ensure((size_t)n + 1);
    W.cnt.ensure((size_t)n + 2);
    W.tsum.ensure((size_t)n / kBlock + 2);
    W.state.ensure(ST_COUNT);
    W.key.ensure(1);
    W.host.ensure(ST_COUNT);
    const dim3 g((unsigned)std::max(1, std::min(L.grid, 2048))), blk(kBlock);
    PS_HIP_CHECK(hipMemsetAsync(W.state.ptr, 0, ST_COUNT * sizeof(int), s));
    hipLaunchKernelGGL(cm_init_kernel, g, blk, 0, s, n, ptr, col, pos, W.claim.ptr, W.cnt.ptr);
    PS_HIP_CHECK(hipGetLastError());
    // 1. rows without an off-diagonal entry
    const int n_iso = (int)device_exclusive_scan(L, W.cnt.ptr, n, S);
    if (n_iso > 0) hipLaunchKernelGGL(cm_place_flagged_kernel, g, blk, 0, s, n, W.cnt.ptr, 0, order, pos);
    int placed = n_iso, comps = 0, leftover = 0, widest = 1;
    while (placed < n) {
        if (comps == kReorderMaxComponents) { // 4. the rest in index order
            hipLaunchKernelGGL(cm_flag_unvisited_kernel, g, blk, 0, s, n, pos, W.cnt.ptr);
            leftover = (int)device_exclusive_scan(L, W.cnt.ptr, n, S);
            PS_REQUIRE(placed + leftover == n, PSOLVE_HIP_EINVAL, "reorder: vertex count does not add up");
            hipLaunchKernelGGL(cm_place_flagged_kernel, g, blk, 0, s, n, W.cnt.ptr, placed, order, pos);
            placed = n;
            break;
        }
        // 2. start vertex
        PS_HIP_CHECK(hipMemsetAsync(W.key.ptr, 0xff, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(cm_min_key_kernel, g, blk, 0, s, n, ptr, pos, W.key.ptr);
        hipLaunchKernelGGL(cm_place_start_kernel, dim3(1), dim3(64), 0, s, W.key.ptr, placed, order, pos, W.state.ptr);
        ++comps;
        // 3. levels, enqueued in batches; the state is read once per batch.  The kernels stride over the frontier with
        // whatever grid they get; it is sized for a few times the widest frontier seen so far (a level of a mesh is a
        // surface: a few hundred workgroups at most, and a launch of 2048 mostly idle ones costs three times as much)
        int par = 0, batch = 16;
        for (;;) {
            const int want = (int)std::min<long long>(g.x, std::max<long long>(32, (4ll * widest + kBlock - 1) / kBlock));
            const dim3 gl((unsigned)want);
            for (int b = 0; b < batch; ++b) {
                hipLaunchKernelGGL(cm_claim_kernel, gl, blk, 0, s, par, W.state.ptr, order, ptr, col, pos, W.claim.ptr);
                hipLaunchKernelGGL(cm_count_kernel, gl, blk, 0, s, par, W.state.ptr, order, ptr, col, W.claim.ptr,
                                   W.cnt.ptr, W.tsum.ptr);
                hipLaunchKernelGGL(cm_scan_tiles_kernel, dim3(1), blk, 0, s, par, W.state.ptr, W.tsum.ptr);
                hipLaunchKernelGGL(cm_write_kernel, gl, blk, 0, s, par, W.state.ptr, ptr, col, W.claim.ptr, W.cnt.ptr,
                                   W.tsum.ptr, order, pos);
                par ^= 1;
            }
            PS_HIP_CHECK(hipGetLastError());
            PS_HIP_CHECK(hipMemcpyAsync(W.host.ptr, W.state.ptr, ST_COUNT * sizeof(int), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
            const int lo = W.host.ptr[ST_LO + par], hi = W.host.ptr[ST_HI + par];
            widest = std::max(widest, W.host.ptr[ST_WIDEST]);
            PS_REQUIRE(lo >= placed && hi >= lo && hi <= n, PSOLVE_HIP_EINVAL,
                       "reorder: breadth-first search left its bounds (duplicate column indices in a row?)");
            if (lo == hi) {
                placed = hi;
                break;
            }
            batch = 64;
        }
    }
    PS_HIP_CHECK(hipStreamSynchronize(s));
    if (info) {
        info->levels = W.host.ptr[ST_LEVELS];
        info->components = comps;
        info->isolated = n_iso;
        info->leftover = leftover;
        if (comps == 0) info->levels = 0;
    }
}

} // namespace psolve
