// pattern.hip -- builds the pattern dictionary of a CSR operator on the device (pattern.hpp).
//   1. every row hashes (length, col - row of its entries) and enters the hash in a small open-addressing table;
//      the slot remembers the smallest row with that hash (its representative).  The rows of a wave that share a
//      hash send one lane, and a slot that already holds the hash and a smaller representative is left alone, so
//      the 16.7 M rows of a 256^3 stencil cost a few thousand atomics;
//   2. the occupied slots are numbered in slot order (pattern ids; deterministic: the order is the hash's);
//   3. every row compares itself with the representative of its slot ENTRY BY ENTRY -- equal hashes are not taken
//      for equal patterns -- and takes the slot's id;
//   4. the offsets of the representatives become the dictionary.
#include "pattern.hpp"

namespace psolve {

namespace {

constexpr int kPatSlots = 16384; // power of two, 4 x kPatMaxPatterns
constexpr int kPatProbes = 64;

__device__ __forceinline__ unsigned long long pat_mix(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 31);
}

__device__ __forceinline__ unsigned long long pat_row_hash(int r, int rs, int len, const int *__restrict__ col)
{
    unsigned long long h = pat_mix(0x243F6A8885A308D3ull, (unsigned long long)len);
    for (int j = 0; j < len; ++j) h = pat_mix(h, (unsigned long long)(unsigned)(col[rs + j] - r));
    return h | 1ull; // 0 marks an empty slot
}

// ctrl: [0] failure flags, [1] longest row, [2] patterns
__global__ __launch_bounds__(kBlock) void pat_insert_kernel(int n, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col, unsigned long long *keys,
                                                            int *rep, int *ctrl)
{
    int maxlen = 0;
    for (int r0 = blockIdx.x * kBlock; r0 < n; r0 += gridDim.x * kBlock) {
        // an operator without a dictionary (every row its own pattern) fills the table within the first few
        // thousand rows: the rest of the matrix is not worth 64 probes per row
        if (__hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const int r = r0 + threadIdx.x;
        unsigned long long h = 0;
        if (r < n) {
            const int rs = rowptr[r], len = rowptr[r + 1] - rs;
            if (len > kPatMaxLen) {
                ctrl[0] = 1;
            } else {
                maxlen = max(maxlen, len);
                h = pat_row_hash(r, rs, len, col);
            }
        }
        // one lane per distinct hash of the wave (the lowest: the smallest row)
        unsigned long long todo = __ballot(h != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const unsigned long long h0 = __shfl(h, src);
            const unsigned long long same = __ballot(h == h0);
            todo &= ~same;
            if ((int)(threadIdx.x & 63) != src) continue;
            int slot = (int)(h0 >> 20) & (kPatSlots - 1);
            bool placed = false;
            for (int p = 0; p < kPatProbes && !placed; ++p, slot = (slot + 1) & (kPatSlots - 1)) {
                unsigned long long k = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == 0) k = atomicCAS(&keys[slot], 0ull, h0), k = k == 0 ? h0 : k;
                if (k == h0) {
                    if (__hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > r) atomicMin(&rep[slot], r);
                    placed = true;
                }
            }
            if (!placed) ctrl[0] = 1; // too many patterns for the table
        }
    }
    if (maxlen > 0) atomicMax(&ctrl[1], maxlen);
}

// one workgroup: pattern ids in slot order
__global__ __launch_bounds__(kBlock) void pat_number_kernel(const unsigned long long *__restrict__ keys, int *slot_pid,
                                                            int *ctrl)
{
    __shared__ int cnt[kBlock];
    constexpr int per = kPatSlots / kBlock;
    int c = 0;
    for (int k = 0; k < per; ++k) c += keys[threadIdx.x * per + k] != 0;
    cnt[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < kBlock; ++t) {
            const int v = cnt[t];
            cnt[t] = run;
            run += v;
        }
        ctrl[2] = run;
        if (run > kPatMaxPatterns) ctrl[0] = 1;
    }
    __syncthreads();
    int run = cnt[threadIdx.x];
    for (int k = 0; k < per; ++k) {
        const int s = threadIdx.x * per + k;
        slot_pid[s] = keys[s] != 0 ? run++ : -1;
    }
}

__global__ __launch_bounds__(kBlock) void pat_assign_kernel(int n, const int *__restrict__ rowptr,
                                                            const int *__restrict__ col,
                                                            const unsigned long long *__restrict__ keys,
                                                            const int *__restrict__ rep,
                                                            const int *__restrict__ slot_pid, unsigned short *id, int *ctrl)
{
    if (ctrl[0]) return; // (the build has failed already)
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const int rs = rowptr[r], len = rowptr[r + 1] - rs;
        if (len > kPatMaxLen) return;
        const unsigned long long h = pat_row_hash(r, rs, len, col);
        int slot = (int)(h >> 20) & (kPatSlots - 1);
        bool found = false;
        for (int p = 0; p < kPatProbes && !found; ++p) {
            if (keys[slot] == h) found = true;
            else slot = (slot + 1) & (kPatSlots - 1);
        }
        bool ok = found;
        if (found) {
            const int q = rep[slot], qs = rowptr[q];
            ok = rowptr[q + 1] - qs == len;
            for (int j = 0; j < len && ok; ++j) ok = (col[rs + j] - r) == (col[qs + j] - q);
            if (ok) id[r] = (unsigned short)slot_pid[slot];
        }
        if (!ok) ctrl[0] = 1; // two different patterns under one hash (or no slot): no dictionary
    }
}

__global__ __launch_bounds__(kBlock) void pat_dictionary_kernel(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                const unsigned long long *__restrict__ keys,
                                                                const int *__restrict__ rep,
                                                                const int *__restrict__ slot_pid, int ml, int *off)
{
    for (int s = blockIdx.x * kBlock + threadIdx.x; s < kPatSlots; s += gridDim.x * kBlock) {
        if (keys[s] == 0) continue;
        const int q = rep[s], qs = rowptr[q], len = rowptr[q + 1] - qs, pid = slot_pid[s];
        for (int j = 0; j < ml; ++j) off[(size_t)pid * ml + j] = j < len ? col[qs + j] - q : 0;
    }
}

} // namespace

bool PatMatrix::build(const Launch &L, const CsrDev &A)
{
    reset();
    if (A.n <= 0 || A.nnz <= 0) return false;
    keys.ensure(kPatSlots);
    rep.ensure(kPatSlots);
    slot_pid.ensure(kPatSlots);
    ctrl.ensure(8);
    host.ensure(8);
    id.ensure((size_t)A.n + 8);
    hipStream_t s = L.stream;
    PS_HIP_CHECK(hipMemsetAsync(keys.ptr, 0, kPatSlots * sizeof(unsigned long long), s));
    PS_HIP_CHECK(hipMemsetAsync(rep.ptr, 0x7f, kPatSlots * sizeof(int), s)); // large: atomicMin finds the smallest row
    PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
    const dim3 g(L.grid), blk(kBlock);
    hipLaunchKernelGGL(pat_insert_kernel, g, blk, 0, s, A.n, A.rowptr, A.col, keys.ptr, rep.ptr, ctrl.ptr);
    hipLaunchKernelGGL(pat_number_kernel, dim3(1), blk, 0, s, keys.ptr, slot_pid.ptr, ctrl.ptr);
    hipLaunchKernelGGL(pat_assign_kernel, g, blk, 0, s, A.n, A.rowptr, A.col, keys.ptr, rep.ptr, slot_pid.ptr, id.ptr,
                       ctrl.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int failed = host.ptr[0], ml = host.ptr[1], npat = host.ptr[2];
    if (failed || ml <= 0 || npat <= 0 || npat > kPatMaxPatterns || npat * ml > kPatMaxDict) return false;
    off.ensure((size_t)npat * ml + 8);
    hipLaunchKernelGGL(pat_dictionary_kernel, dim3(std::max(1, kPatSlots / kBlock)), blk, 0, s, A.rowptr, A.col, keys.ptr,
                       rep.ptr, slot_pid.ptr, ml, off.ptr);
    PS_HIP_CHECK(hipGetLastError());
    view.id = id.ptr;
    view.off = off.ptr;
    view.ml = ml;
    view.npat = npat;
    valid = true;
    return true;
}

} // namespace psolve
