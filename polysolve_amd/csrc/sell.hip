// sell.hip -- builds the SELL-64-sigma copy of a CSR operator on the device (sell.hpp).
//   1. per window of kSellWindow rows: rows sorted by stored length, longest first (bitonic network in LDS, ties
//      in row order, so the result is a function of the pattern alone); slot = (row, length), slice widths;
//   2. exclusive scan of the slice sizes;
//   3. one lane per row copies its row into the slice, column-major, and fills the padding.
#include "sell.hpp"

namespace psolve {

namespace {

__global__ __launch_bounds__(kBlock) void sell_sort_kernel(int n, const int *__restrict__ rowptr, int2 *__restrict__ slot,
                                                           int *__restrict__ slice_size)
{
    __shared__ unsigned key[kSellWindow];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * kSellWindow;
    for (int i = tid; i < kSellWindow; i += kBlock) {
        const int r = row0 + i;
        const int len = r < n ? rowptr[r + 1] - rowptr[r] : 0;
        // ascending keys = descending length, then ascending row; 22 bits of length are plenty for a tie-break
        key[i] = ((unsigned)(0x3FFFFF - min(len, 0x3FFFFF)) << 10) | (unsigned)i;
    }
    __syncthreads();
    for (int k = 2; k <= kSellWindow; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < kSellWindow / 2; t += kBlock) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const bool up = (i & k) == 0;
                const unsigned a = key[i], c = key[l];
                if ((a > c) == up) {
                    key[i] = c;
                    key[l] = a;
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < kSellWindow; i += kBlock) {
        const int r = row0 + (int)(key[i] & 1023u);
        const int len = r < n ? rowptr[r + 1] - rowptr[r] : 0;
        const int s = row0 + i; // slot index
        if (s < ((n + 63) & ~63)) {
            slot[s] = make_int2(r < n ? r : -1, len);
            if ((i & 63) == 0) slice_size[s >> 6] = len * 64; // the slice's first row is its longest
        }
    }
}

template <bool VALUES_ONLY>
__global__ __launch_bounds__(kBlock) void sell_fill_kernel(int nslices, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col, const double *__restrict__ val,
                                                           const int *__restrict__ slice_ptr,
                                                           const int2 *__restrict__ slot, int *__restrict__ scol,
                                                           double *__restrict__ sval)
{
    const int lane = threadIdx.x & 63;
    const int wave_g = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
    for (int s = wave_g; s < nslices; s += nwaves) {
        const int base = slice_ptr[s], w = (slice_ptr[s + 1] - base) >> 6;
        const int2 me = slot[s * 64 + lane];
        const int rp = me.x >= 0 ? rowptr[me.x] : 0, len = me.y;
        const int padcol = len > 0 ? col[rp + len - 1] : 0; // never multiplied (the product skips k >= len)
        for (int k = 0; k < w; ++k) {
            const int dst = base + k * 64 + lane;
            if (!VALUES_ONLY) scol[dst] = k < len ? col[rp + k] : padcol;
            sval[dst] = k < len ? val[rp + k] : 0.0;
        }
    }
}

} // namespace

bool SellMatrix::build(const Launch &L, const CsrDev &A, SymbolicScratch &S, double max_fill)
{
    reset();
    if (A.n <= 0 || A.nnz <= 0) return false;
    const int nslices = (A.n + 63) / 64;
    const int nwin = (A.n + kSellWindow - 1) / kSellWindow;
    slot.ensure((size_t)nslices * 64);
    slice_ptr.ensure((size_t)nslices + 2);
    hipLaunchKernelGGL(sell_sort_kernel, dim3((unsigned)nwin), dim3(kBlock), 0, L.stream, A.n, A.rowptr, slot.ptr,
                       slice_ptr.ptr);
    PS_HIP_CHECK(hipGetLastError());
    int64_t total = 0;
    try {
        total = device_exclusive_scan(L, slice_ptr.ptr, nslices, S);
    } catch (const Error &e) {
        if (e.code == PSOLVE_HIP_ERANGE) return false;
        throw;
    }
    if ((double)total > max_fill * (double)A.nnz) return false;
    padded = total;
    col.ensure((size_t)total + 64);
    val.ensure((size_t)total + 64);
    view.nslices = nslices;
    view.slice_ptr = slice_ptr.ptr;
    view.col = col.ptr;
    view.val = val.ptr;
    view.slot = slot.ptr;
    const int grid = std::max(8, std::min(L.num_cus * 8, (nslices + 3) / 4));
    hipLaunchKernelGGL((sell_fill_kernel<false>), dim3((unsigned)grid), dim3(kBlock), 0, L.stream, nslices, A.rowptr,
                       A.col, A.val, slice_ptr.ptr, slot.ptr, col.ptr, val.ptr);
    PS_HIP_CHECK(hipGetLastError());
    valid = true;
    return true;
}

void SellMatrix::refill(const Launch &L, const CsrDev &A)
{
    PS_REQUIRE(valid && view.nslices == (A.n + 63) / 64, PSOLVE_HIP_EINVAL, "SELL refill without a matching build");
    const int grid = std::max(8, std::min(L.num_cus * 8, (view.nslices + 3) / 4));
    hipLaunchKernelGGL((sell_fill_kernel<true>), dim3((unsigned)grid), dim3(kBlock), 0, L.stream, view.nslices, A.rowptr,
                       A.col, A.val, slice_ptr.ptr, slot.ptr, col.ptr, val.ptr);
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
