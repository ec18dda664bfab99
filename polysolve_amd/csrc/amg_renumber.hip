// amg_renumber.hip -- locality renumbering of the coarse AMG levels, applied after the hierarchy is built.
//
// Why: AMGCL numbers the aggregates of a level in the order its greedy sweep creates them.  For the level-1
// operator of a 3-D grid that order interleaves neighbouring lines of aggregates, and a gather instruction of the
// wide-row products touches 16-25 distinct 128-byte lines of x instead of the ~8 a compact numbering needs
// (profiles/r02_spmv_lab.md section 5b: the coarse-level Chebyshev product at 55 % of the HBM roofline).
// What: every level l >= 1 is renumbered by  (new id of the node's aggregate on level l + 1, old id)  -- the nodes of
// one coarse aggregate become consecutive, recursively from the coarsest level down, which is a space-filling order
// built from nothing but the aggregate maps the setup already has (no coordinates).  The hierarchy is the same
// hierarchy: A_l -> Pi_l A_l Pi_l^T, P_l -> Pi_l P_l Pi_{l+1}^T, R_l = P_l^T, the same aggregates (the reference's,
// /root/reference/src/polysolve/linear/AMGCL.cpp:32-65 -> amgcl plain_aggregates), the same numbers; only the order
// of the entries inside a row -- and with it the order of the additions of a row sum -- changes.
#include <algorithm>

#include "amg_symbolic.hpp"

namespace psolve {

namespace {

constexpr unsigned long long kEmpty64 = ~0ull;

template <int GROUP>
__device__ __forceinline__ void rn_group_sync()
{
    if constexpr (GROUP >= kBlock) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ascending bitonic sort of a[0..p2) (64-bit keys), p2 a power of two, by GROUP lanes
template <int GROUP>
__device__ __forceinline__ void rn_bitonic64(unsigned long long *a, int p2, int lane)
{
    for (int k = 2; k <= p2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (p2 >> 1); t += GROUP) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long x = a[lo], y = a[hi];
                if ((x > y) == up) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
            rn_group_sync<GROUP>();
        }
}

__device__ __forceinline__ int rn_pow2(int u)
{
    int p = 1;
    while (p < u) p <<= 1;
    return p;
}

__global__ __launch_bounds__(kBlock) void iota_kernel(int n, int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = i;
}

// key[i] = new id of node i's aggregate (parent_new == nullptr: the aggregate id itself); nodes without an aggregate
// (id < 0: rows the strength graph isolates) go behind everybody else
__global__ __launch_bounds__(kBlock) void parent_key_kernel(int n, const int *__restrict__ id,
                                                             const int *__restrict__ parent_new, int n_parent,
                                                             int *__restrict__ key)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int a = id[i];
        key[i] = a < 0 ? n_parent : (parent_new ? parent_new[a] : a);
    }
}

// order[k] = old id of the node at new position k  ->  new_of_old[order[k]] = k
__global__ __launch_bounds__(kBlock) void invert_perm_kernel(int n, const int *__restrict__ order, int *__restrict__ new_of_old)
{
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) new_of_old[order[k]] = k;
}

// out[new_of_old ? new_of_old[i] : i] = value_map ? (id[i] < 0 ? id[i] : value_map[id[i]]) : id[i]
__global__ __launch_bounds__(kBlock) void relabel_ids_kernel(int n, const int *__restrict__ id,
                                                              const int *__restrict__ new_of_old,
                                                              const int *__restrict__ value_map, int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int a = id[i];
        out[new_of_old ? new_of_old[i] : i] = (a < 0 || !value_map) ? a : value_map[a];
    }
}

__global__ __launch_bounds__(kBlock) void permute_f64_kernel(int n, const double *__restrict__ in,
                                                              const int *__restrict__ new_of_old, double *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[new_of_old[i]] = in[i];
}

// cnt[new row] = length of the old row; counters[0] = longest row, counters[1] = rows longer than `big` (listed)
__global__ __launch_bounds__(kBlock) void permuted_counts_kernel(int n, const int *__restrict__ ptr,
                                                                  const int *__restrict__ row_new, int big,
                                                                  int *__restrict__ cnt, int *__restrict__ counters,
                                                                  int *__restrict__ list)
{
    int mx = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int len = ptr[i + 1] - ptr[i];
        cnt[row_new ? row_new[i] : i] = len;
        mx = max(mx, len);
        if (len > big) list[atomicAdd(&counters[1], 1)] = i;
    }
    if (mx) atomicMax(&counters[0], mx);
}

// rows of length in (lo_len, hi_len]: GROUP lanes sort the row's (new column, position) pairs in LDS and write the
// row at its new place
template <int GROUP, int CAP>
__global__ __launch_bounds__(kBlock) void permute_rows_lds_kernel(int n, const int *__restrict__ ptr,
                                                                   const int *__restrict__ col,
                                                                   const double *__restrict__ val, int lo_len, int hi_len,
                                                                   const int *__restrict__ row_new,
                                                                   const int *__restrict__ col_new,
                                                                   const int *__restrict__ optr, int *__restrict__ ocol,
                                                                   double *__restrict__ oval, int *__restrict__ omap)
{
    constexpr int GPB = kBlock / GROUP;
    __shared__ unsigned long long lst[GPB * CAP];
    const int g = threadIdx.x / GROUP, lane = threadIdx.x % GROUP;
    unsigned long long *mine = lst + g * CAP;
    for (int i = blockIdx.x * GPB + g; i < n; i += gridDim.x * GPB) {
        const int rb = ptr[i], u = ptr[i + 1] - rb;
        if (u <= lo_len || u > hi_len) continue;
        const int p2 = rn_pow2(u);
        for (int s = lane; s < p2; s += GROUP) {
            unsigned long long k = kEmpty64;
            if (s < u) {
                const int c = col[rb + s];
                k = ((unsigned long long)(unsigned)(col_new ? col_new[c] : c) << 32) | (unsigned)s;
            }
            mine[s] = k;
        }
        rn_group_sync<GROUP>();
        rn_bitonic64<GROUP>(mine, p2, lane);
        const int ob = optr[row_new ? row_new[i] : i];
        for (int s = lane; s < u; s += GROUP) {
            const unsigned long long k = mine[s];
            ocol[ob + s] = (int)(k >> 32);
            if (oval) oval[ob + s] = val[rb + (int)(k & 0xffffffffu)];
            if (omap) omap[ob + s] = rb + (int)(k & 0xffffffffu); // where this entry came from: a refresh only gathers values
        }
        rn_group_sync<GROUP>();
    }
}

// rows too long for LDS: one workgroup per row, sorted in a slice of HBM scratch
__global__ __launch_bounds__(kBlock) void permute_rows_global_kernel(int nlist, const int *__restrict__ list,
                                                                      const int *__restrict__ ptr,
                                                                      const int *__restrict__ col,
                                                                      const double *__restrict__ val,
                                                                      unsigned long long *scratch, long long stride,
                                                                      const int *__restrict__ row_new,
                                                                      const int *__restrict__ col_new,
                                                                      const int *__restrict__ optr, int *__restrict__ ocol,
                                                                      double *__restrict__ oval)
{
    unsigned long long *buf = scratch + (long long)blockIdx.x * stride;
    for (int r = blockIdx.x; r < nlist; r += gridDim.x) {
        const int i = list[r];
        const int rb = ptr[i], u = ptr[i + 1] - rb;
        const int p2 = rn_pow2(u);
        for (int s = threadIdx.x; s < p2; s += kBlock) {
            unsigned long long k = kEmpty64;
            if (s < u) {
                const int c = col[rb + s];
                k = ((unsigned long long)(unsigned)(col_new ? col_new[c] : c) << 32) | (unsigned)s;
            }
            buf[s] = k;
        }
        __syncthreads();
        rn_bitonic64<kBlock>(buf, p2, threadIdx.x);
        const int ob = optr[row_new ? row_new[i] : i];
        for (int s = threadIdx.x; s < u; s += kBlock) {
            const unsigned long long k = buf[s];
            ocol[ob + s] = (int)(k >> 32);
            if (oval) oval[ob + s] = val[rb + (int)(k & 0xffffffffu)];
        }
        __syncthreads();
    }
}

} // namespace

void launch_iota(const Launch &L, int n, int *out)
{
    hipLaunchKernelGGL(iota_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, out);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_relabel_ids(const Launch &L, int n, const int *id, const int *new_of_old, const int *value_map, int *out)
{
    hipLaunchKernelGGL(relabel_ids_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, id, new_of_old, value_map, out);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_permute_f64(const Launch &L, int n, const double *in, const int *new_of_old, double *out)
{
    hipLaunchKernelGGL(permute_f64_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, in, new_of_old, out);
    PS_HIP_CHECK(hipGetLastError());
}

void device_order_by_parent(const Launch &L, int n, const int *id, const int *parent_new, int n_parent,
                            DeviceBuffer<int> &new_of_old, SymbolicScratch &S, DeviceBuffer<int> &w_key,
                            DeviceBuffer<int> &w_iota, DeviceBuffer<int> &w_ptr, DeviceBuffer<int> &w_order,
                            DeviceBuffer<int> &w_map)
{
    // the members of every parent, ascending, parent after parent = the transpose of the one-entry-per-row matrix
    // (node i, key[i]); its column array IS the new order
    w_key.ensure((size_t)n + 4);
    w_iota.ensure((size_t)n + 2);
    new_of_old.ensure((size_t)n + 1);
    hipLaunchKernelGGL(parent_key_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, id, parent_new, n_parent, w_key.ptr);
    hipLaunchKernelGGL(iota_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n + 1, w_iota.ptr);
    PS_HIP_CHECK(hipGetLastError());
    device_transpose_pattern(L, n, n_parent + 1, w_iota.ptr, w_key.ptr, n, w_ptr, w_order, w_map, S);
    hipLaunchKernelGGL(invert_perm_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, w_order.ptr, new_of_old.ptr);
    PS_HIP_CHECK(hipGetLastError());
}

void device_permute_csr(const Launch &L, int n, int64_t nnz, const int *ptr, const int *col, const double *val,
                        const int *row_new, const int *col_new, DeviceBuffer<int> &optr, DeviceBuffer<int> &ocol,
                        DeviceBuffer<double> *oval, SymbolicScratch &S, DeviceBuffer<int> *omap)
{
    hipStream_t s = L.stream;
    optr.ensure((size_t)n + 1);
    ocol.ensure((size_t)nnz + 4);
    if (oval) oval->ensure((size_t)nnz + 4);
    S.cand.ensure((size_t)n + 1); // list of long rows
    S.counters.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(S.counters.ptr, 0, 16 * sizeof(int), s));
    constexpr int kBig = 2048;
    const dim3 g(L.grid), blk(kBlock);
    hipLaunchKernelGGL(permuted_counts_kernel, g, blk, 0, s, n, ptr, row_new, kBig, optr.ptr, S.counters.ptr, S.cand.ptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t total = device_exclusive_scan(L, optr.ptr, n, S);
    PS_REQUIRE(total == nnz, PSOLVE_HIP_EINVAL, "permute_csr: row pointers do not add up");
    S.host.ensure(16);
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.counters.ptr, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int *hc = reinterpret_cast<const int *>(S.host.ptr);
    const int longest = hc[0], nbig = hc[1];
    double *ov = oval ? oval->ptr : nullptr;
    // (omap: source position of every output entry -- kept by the caller for the next factorize of the same pattern, which
    // then gathers the values instead of sorting every row again; not offered for rows beyond LDS: the caller gets none)
    if (omap && nbig > 0) omap->release();
    else if (omap) omap->ensure((size_t)nnz + 4);
    int *om = (omap && nbig == 0) ? omap->ptr : nullptr;
    hipLaunchKernelGGL((permute_rows_lds_kernel<16, 64>), g, blk, 0, s, n, ptr, col, val, 0, 64, row_new, col_new, optr.ptr,
                       ocol.ptr, ov, om);
    if (longest > 64)
        hipLaunchKernelGGL((permute_rows_lds_kernel<64, 256>), g, blk, 0, s, n, ptr, col, val, 64, 256, row_new, col_new,
                           optr.ptr, ocol.ptr, ov, om);
    if (longest > 256)
        hipLaunchKernelGGL((permute_rows_lds_kernel<256, kBig>), g, blk, 0, s, n, ptr, col, val, 256, kBig, row_new,
                           col_new, optr.ptr, ocol.ptr, ov, om);
    if (nbig > 0) {
        long long stride = 1;
        while (stride < longest) stride <<= 1;
        const long long budget = 1ll << 27; // 64-bit words (1 GiB)
        const int grid = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(nbig, 256), budget / stride));
        S.table.ensure((size_t)(2 * stride * grid));
        hipLaunchKernelGGL(permute_rows_global_kernel, dim3(grid), blk, 0, s, nbig, S.cand.ptr, ptr, col, val,
                           reinterpret_cast<unsigned long long *>(S.table.ptr), stride, row_new, col_new, optr.ptr,
                           ocol.ptr, ov);
    }
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
