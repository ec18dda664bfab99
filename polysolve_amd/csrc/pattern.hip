// pattern.hip -- builds the pattern dictionary of a CSR operator on the device (pattern.hpp).
//   1. every row hashes (length, col - row of its entries) and enters the hash in a small open-addressing table;
//      the slot remembers the smallest row with that hash (its representative).  The rows of a wave that share a
//      hash send one lane, and a slot that already holds the hash and a smaller representative is left alone, so
//      the 16.7 M rows of a 256^3 stencil cost a few thousand atomics;
//   2. the occupied slots are numbered in slot order (pattern ids; deterministic: the order is the hash's);
//   3. every row compares itself with the representative of its slot ENTRY BY ENTRY -- equal hashes are not taken
//      for equal patterns -- and takes the slot's id;
//   4. the offsets of the representatives become the dictionary.
#include <algorithm>
#include <cstring>
#include <vector>

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

// ---- row kinds: (pattern id, values) ------------------------------------------------------------------------------
constexpr int kKindSlots = 4096; // power of two; at most kKindMax of them may be taken
constexpr int kKindMax = 1024;
constexpr int kKindProbes = 32;

__device__ __forceinline__ unsigned long long kind_row_hash(int pid, int rs, int len, const double *__restrict__ val)
{
    unsigned long long h = pat_mix(0x13198A2E03707344ull, (unsigned long long)(unsigned)pid);
    for (int j = 0; j < len; ++j) h = pat_mix(h, (unsigned long long)__double_as_longlong(val[rs + j]));
    return h | 1ull;
}

// ctrl: [0] failure, [2] kinds, [3] slots taken so far
__global__ __launch_bounds__(kBlock) void kind_insert_kernel(int n, const int *__restrict__ rowptr,
                                                             const double *__restrict__ val,
                                                             const unsigned short *__restrict__ pid,
                                                             unsigned long long *keys, int *rep, int *ctrl)
{
    for (int r0 = blockIdx.x * kBlock; r0 < n; r0 += gridDim.x * kBlock) {
        // rows that do not repeat (the usual FEM matrix) take kKindMax slots within the first thousand rows: stop reading
        if (__hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const int r = r0 + threadIdx.x;
        unsigned long long h = 0;
        if (r < n) {
            const int rs = rowptr[r];
            h = kind_row_hash(pid[r], rs, rowptr[r + 1] - rs, val);
        }
        unsigned long long todo = __ballot(h != 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const unsigned long long h0 = __shfl(h, src);
            const unsigned long long same = __ballot(h == h0);
            todo &= ~same;
            if ((int)(threadIdx.x & 63) != src) continue;
            int slot = (int)(h0 >> 20) & (kKindSlots - 1);
            bool placed = false;
            for (int p = 0; p < kKindProbes && !placed; ++p, slot = (slot + 1) & (kKindSlots - 1)) {
                unsigned long long k = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k == 0) {
                    k = atomicCAS(&keys[slot], 0ull, h0);
                    if (k == 0) {
                        k = h0;
                        if (atomicAdd(&ctrl[3], 1) >= kKindMax) ctrl[0] = 1;
                    }
                }
                if (k == h0) {
                    if (__hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > r) atomicMin(&rep[slot], r);
                    placed = true;
                }
            }
            if (!placed) ctrl[0] = 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void kind_number_kernel(const unsigned long long *__restrict__ keys, int *slot_kid,
                                                             int *ctrl)
{
    __shared__ int cnt[kBlock];
    constexpr int per = kKindSlots / kBlock;
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
        if (run > kKindMax) ctrl[0] = 1;
    }
    __syncthreads();
    int run = cnt[threadIdx.x];
    for (int k = 0; k < per; ++k) {
        const int s = threadIdx.x * per + k;
        slot_kid[s] = keys[s] != 0 ? run++ : -1;
    }
}

__global__ __launch_bounds__(kBlock) void kind_assign_kernel(int n, const int *__restrict__ rowptr,
                                                             const double *__restrict__ val,
                                                             const unsigned short *__restrict__ pid,
                                                             const unsigned long long *__restrict__ keys,
                                                             const int *__restrict__ rep, const int *__restrict__ slot_kid,
                                                             unsigned short *kind, int *ctrl)
{
    if (ctrl[0]) return;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const int rs = rowptr[r], len = rowptr[r + 1] - rs;
        const unsigned long long h = kind_row_hash(pid[r], rs, len, val);
        int slot = (int)(h >> 20) & (kKindSlots - 1);
        bool found = false;
        for (int p = 0; p < kKindProbes && !found; ++p) {
            if (keys[slot] == h) found = true;
            else slot = (slot + 1) & (kKindSlots - 1);
        }
        bool ok = found;
        if (found) {
            // equal hashes are not taken for equal rows: the pattern and every value, bit for bit
            const int q = rep[slot], qs = rowptr[q];
            ok = pid[q] == pid[r] && rowptr[q + 1] - qs == len;
            for (int j = 0; j < len && ok; ++j) ok = __double_as_longlong(val[rs + j]) == __double_as_longlong(val[qs + j]);
            if (ok) kind[r] = (unsigned short)slot_kid[slot];
        }
        if (!ok) ctrl[0] = 1;
    }
}

__global__ __launch_bounds__(kBlock) void kind_dictionary_kernel(const int *__restrict__ rowptr, const double *__restrict__ val,
                                                                 const unsigned short *__restrict__ pid,
                                                                 const int *__restrict__ off,
                                                                 const unsigned long long *__restrict__ keys,
                                                                 const int *__restrict__ rep, const int *__restrict__ slot_kid,
                                                                 int ml, int kml, double *kval, int *koff, int *klen, int *krep)
{
    for (int s = blockIdx.x * kBlock + threadIdx.x; s < kKindSlots; s += gridDim.x * kBlock) {
        if (keys[s] == 0) continue;
        const int q = rep[s], qs = rowptr[q], len = rowptr[q + 1] - qs, kid = slot_kid[s], p = pid[q];
        klen[kid] = len;
        krep[kid] = q;
        for (int j = 0; j < kml; ++j) {
            kval[(size_t)kid * kml + j] = j < len ? val[qs + j] : 0.0;
            koff[(size_t)kid * kml + j] = j < len ? off[(size_t)p * ml + j] : 0;
        }
    }
}

__global__ __launch_bounds__(kBlock) void kind_table_fill_kernel(const unsigned long long *__restrict__ keys,
                                                                 const int *__restrict__ rep, const int *__restrict__ slot_kid,
                                                                 const double *__restrict__ v, double *table)
{
    for (int s = blockIdx.x * kBlock + threadIdx.x; s < kKindSlots; s += gridDim.x * kBlock)
        if (keys[s] != 0) table[slot_kid[s]] = v[rep[s]];
}

__global__ __launch_bounds__(kBlock) void kind_table_check_kernel(int n, const unsigned short *__restrict__ kind,
                                                                  const double *__restrict__ v,
                                                                  const double *__restrict__ table, int *flag)
{
    bool bad = false;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock)
        bad = bad || __double_as_longlong(v[r]) != __double_as_longlong(table[kind[r]]);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) *flag = 1;
}

} // namespace

// ---- block-row kinds of a 3x3-block operator ------------------------------------------------------------------------
namespace {
constexpr int kBKindSlots = 1024, kBKindMax = 256, kBKindProbes = 32, kBKindMaxLen = 32;

// 32 lanes per block row: lane j takes block j of the row (72 contiguous bytes: the row's 27 blocks are read as one
// segment), the lanes' hashes -- each mixed with j -- are added up (one thread per row read 1.9 KB by itself: 4.0 ms of a
// 27 ms refresh at M = 100 went into these kernels)
__device__ __forceinline__ unsigned long long bkind_block_hash(int j, int off, const double *__restrict__ v)
{
    unsigned long long h = pat_mix(0xA4093822299F31D0ull + (unsigned long long)j, (unsigned long long)(unsigned)off);
    for (int q = 0; q < 9; ++q) h = pat_mix(h, (unsigned long long)__double_as_longlong(v[q]));
    return h;
}

// ctrl: [0] failure, [1] longest block row, [2] kinds, [3] slots taken
__global__ __launch_bounds__(kBlock) void bkind_insert_kernel(int nb, const int *__restrict__ browptr, const int *__restrict__ bcol,
                                                              const double *__restrict__ bval, unsigned long long *keys,
                                                              int *rep, int *ctrl, unsigned long long *rowhash)
{
    const int j = threadIdx.x & 31, team = threadIdx.x >> 5;
    int maxlen = 0;
    for (int r0 = blockIdx.x * (kBlock / 32); r0 < nb; r0 += gridDim.x * (kBlock / 32)) {
        if (__hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const int r = r0 + team;
        unsigned long long h = 0;
        int len = 0;
        if (r < nb) {
            const int bs = browptr[r];
            len = browptr[r + 1] - bs;
            if (len > kBKindMaxLen) ctrl[0] = 1;
            else if (j < len) h = bkind_block_hash(j, bcol[bs + j] - r, bval + (size_t)9 * (bs + j));
        }
        for (int o = 16; o > 0; o >>= 1) h += __shfl_xor(h, o, 32);
        if (r >= nb || len > kBKindMaxLen) continue;
        h = pat_mix(h, (unsigned long long)len) | 1ull;
        maxlen = max(maxlen, len);
        if (j != 0) continue;
        rowhash[r] = h;
        int slot = (int)(h >> 20) & (kBKindSlots - 1);
        bool placed = false;
        for (int p = 0; p < kBKindProbes && !placed; ++p, slot = (slot + 1) & (kBKindSlots - 1)) {
            unsigned long long k = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 0) {
                k = atomicCAS(&keys[slot], 0ull, h);
                if (k == 0) {
                    k = h;
                    if (atomicAdd(&ctrl[3], 1) >= kBKindMax) ctrl[0] = 1;
                }
            }
            if (k == h) {
                if (__hip_atomic_load(&rep[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > r) atomicMin(&rep[slot], r);
                placed = true;
            }
        }
        if (!placed) ctrl[0] = 1;
    }
    if (maxlen > 0 && j == 0) atomicMax(&ctrl[1], maxlen);
}

__global__ __launch_bounds__(kBlock) void bkind_number_kernel(const unsigned long long *__restrict__ keys, int *slot_kid, int *ctrl)
{
    __shared__ int cnt[kBlock];
    constexpr int per = kBKindSlots / kBlock;
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
        if (run > kBKindMax) ctrl[0] = 1;
    }
    __syncthreads();
    int run = cnt[threadIdx.x];
    for (int k = 0; k < per; ++k) {
        const int s = threadIdx.x * per + k;
        slot_kid[s] = keys[s] != 0 ? run++ : -1;
    }
}

__global__ __launch_bounds__(kBlock) void bkind_assign_kernel(int nb, const int *__restrict__ browptr, const int *__restrict__ bcol,
                                                              const double *__restrict__ bval,
                                                              const unsigned long long *__restrict__ keys,
                                                              const int *__restrict__ rep, const int *__restrict__ slot_kid,
                                                              const unsigned long long *__restrict__ rowhash,
                                                              unsigned short *kind, int *ctrl)
{
    if (ctrl[0]) return;
    const int j = threadIdx.x & 31, team = threadIdx.x >> 5;
    for (int r0 = blockIdx.x * (kBlock / 32); r0 < nb; r0 += gridDim.x * (kBlock / 32)) {
        const int r = r0 + team;
        bool ok = true;
        int slot = 0;
        if (r < nb) {
            const int bs = browptr[r], len = browptr[r + 1] - bs;
            const unsigned long long h = rowhash[r];
            slot = (int)(h >> 20) & (kBKindSlots - 1);
            bool found = false;
            for (int p = 0; p < kBKindProbes && !found; ++p) {
                if (keys[slot] == h) found = true;
                else slot = (slot + 1) & (kBKindSlots - 1);
            }
            ok = found;
            if (found) {
                // equal hashes are not taken for equal block rows: every offset and value, bit for bit (lane j: block j)
                const int q = rep[slot], qs = browptr[q];
                ok = browptr[q + 1] - qs == len;
                if (ok && j < len) {
                    ok = (bcol[bs + j] - r) == (bcol[qs + j] - q);
                    const double *a = bval + (size_t)9 * (bs + j), *b2 = bval + (size_t)9 * (qs + j);
                    for (int t = 0; t < 9; ++t) ok = ok && __double_as_longlong(a[t]) == __double_as_longlong(b2[t]);
                }
            }
        }
        // (the 32 lanes of a row agree: a wave holds two rows)
        const unsigned long long bad = __ballot(!ok);
        const unsigned long long mine = (bad >> (threadIdx.x & 32)) & 0xffffffffull;
        if (r < nb && j == 0) {
            if (mine == 0) kind[r] = (unsigned short)slot_kid[slot];
            else ctrl[0] = 1;
        }
    }
}

// A refactorize under a kept pattern (Newton): the rows that shared a kind before most likely still do.  One pass: every
// block row against the CURRENT values of its previous kind's representative, 32 lanes per row; all equal -> the kinds stand
// and only the dictionary's values are taken again (half of a full build: no hashing, no table).
// (round 6, advice: the block row's length and its blocks' column offsets are compared too -- 4 bytes per block beside the 72
// of its values -- so that a stale "same pattern" verdict cannot leave old offsets in use with new values)
__global__ __launch_bounds__(kBlock) void bkind_verify_kernel(int nb, const int *__restrict__ browptr, const int *__restrict__ bcol,
                                                              const double *__restrict__ bval,
                                                              const unsigned short *__restrict__ kind, const int *__restrict__ krep,
                                                              const int *__restrict__ klen, int nk, int *ctrl)
{
    const int j = threadIdx.x & 31, team = threadIdx.x >> 5;
    bool bad = false;
    for (int r0 = blockIdx.x * (kBlock / 32); r0 < nb; r0 += gridDim.x * (kBlock / 32)) {
        const int r = r0 + team;
        if (r >= nb) continue;
        const int bs = browptr[r], len = browptr[r + 1] - bs;
        const int kd = kind[r];
        if (kd >= nk) {
            bad = true;
            continue;
        }
        const int q = krep[kd];
        if (len != klen[kd]) { // (the representative's own row included: the dictionary's stored length)
            bad = true;
            continue;
        }
        if (q == r) continue;
        if (q < 0 || q >= nb || browptr[q + 1] - browptr[q] != len) {
            bad = true;
            continue;
        }
        if (j >= len) continue;
        bad = bad || (bcol[bs + j] - r) != (bcol[browptr[q] + j] - q);
        const double *a = bval + (size_t)9 * (bs + j), *b2 = bval + (size_t)9 * (browptr[q] + j);
        for (int t = 0; t < 9; ++t) bad = bad || __double_as_longlong(a[t]) != __double_as_longlong(b2[t]);
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) ctrl[0] = 1;
}

__global__ __launch_bounds__(kBlock) void bkind_redictionary_kernel(int nk, const int *__restrict__ browptr, const double *__restrict__ bval,
                                                                    const int *__restrict__ krep, const int *__restrict__ klen, int kml,
                                                                    double *kraw)
{
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < nk * kml * 9; t += gridDim.x * kBlock) {
        const int kid = t / (kml * 9), rem = t - kid * kml * 9, j = rem / 9, q9 = rem - 9 * j;
        kraw[t] = j < klen[kid] ? bval[(size_t)9 * (browptr[krep[kid]] + j) + q9] : 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void bkind_dictionary_kernel(const int *__restrict__ browptr, const int *__restrict__ bcol,
                                                                  const double *__restrict__ bval,
                                                                  const unsigned long long *__restrict__ keys,
                                                                  const int *__restrict__ rep, const int *__restrict__ slot_kid,
                                                                  int kml, int *koff, int *klen, double *kraw, int *krep)
{
    for (int s = blockIdx.x * kBlock + threadIdx.x; s < kBKindSlots; s += gridDim.x * kBlock) {
        if (keys[s] == 0) continue;
        const int q = rep[s], qs = browptr[q], len = browptr[q + 1] - qs, kid = slot_kid[s];
        klen[kid] = len;
        krep[kid] = q;
        for (int j = 0; j < kml; ++j) {
            koff[(size_t)kid * kml + j] = j < len ? bcol[qs + j] - q : 0;
            for (int t = 0; t < 9; ++t) kraw[((size_t)kid * kml + j) * 9 + t] = j < len ? bval[(size_t)9 * (qs + j) + t] : 0.0;
        }
    }
}
} // namespace

bool Bsr3Kinds::build(const Launch &L, const Bsr3Dev &B, bool same_pattern)
{
    const bool had = valid && same_pattern && built_nb == B.nb && view.nk > 0;
    const int nk_prev = view.nk, kml_prev = view.kml;
    reset();
    if (B.nb <= 0 || B.nnzb <= 0 || !B.val) return false;
    hipStream_t s = L.stream;
    const dim3 g(L.grid), blk(kBlock);
    int nk = 0, kml = 0;
    bool verified = false;
    if (had) {
        ctrl.ensure(8);
        host.ensure(8);
        PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(bkind_verify_kernel, g, blk, 0, s, B.nb, B.rowptr, B.col, B.val, kind.ptr, krep.ptr, klen.ptr, nk_prev, ctrl.ptr);
        hipLaunchKernelGGL(bkind_redictionary_kernel, dim3(std::max(1, (nk_prev * kml_prev * 9 + kBlock - 1) / kBlock)), blk, 0, s, nk_prev,
                           B.rowptr, B.val, krep.ptr, klen.ptr, kml_prev, kraw.ptr);
        PS_HIP_CHECK(hipGetLastError());
        PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        if (host.ptr[0] == 0) {
            verified = true;
            nk = nk_prev;
            kml = kml_prev;
        }
    }
    if (!verified) {
    keys.ensure(kBKindSlots);
    rep.ensure(kBKindSlots);
    slot_kid.ensure(kBKindSlots);
    ctrl.ensure(8);
    host.ensure(8);
    kind.ensure((size_t)B.nb + 8);
    PS_HIP_CHECK(hipMemsetAsync(keys.ptr, 0, kBKindSlots * sizeof(unsigned long long), s));
    PS_HIP_CHECK(hipMemsetAsync(rep.ptr, 0x7f, kBKindSlots * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
    rowhash.ensure((size_t)B.nb + 8);
    hipLaunchKernelGGL(bkind_insert_kernel, g, blk, 0, s, B.nb, B.rowptr, B.col, B.val, keys.ptr, rep.ptr, ctrl.ptr, rowhash.ptr);
    hipLaunchKernelGGL(bkind_number_kernel, dim3(1), blk, 0, s, keys.ptr, slot_kid.ptr, ctrl.ptr);
    hipLaunchKernelGGL(bkind_assign_kernel, g, blk, 0, s, B.nb, B.rowptr, B.col, B.val, keys.ptr, rep.ptr, slot_kid.ptr, rowhash.ptr,
                       kind.ptr, ctrl.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int failed = host.ptr[0], ml = host.ptr[1];
    nk = host.ptr[2];
    if (failed || nk <= 0 || nk > kBKindMax || ml <= 0 || ml > kBKindMaxLen) return false;
    if ((int64_t)nk * 8 > (int64_t)B.nb) return false; // (block rows that hardly repeat: the tables would be the matrix again)
    kml = ((ml + 2) / 3) * 3; // (the kernel takes three blocks at a time)
    koff.ensure((size_t)nk * kml + 8);
    klen.ensure((size_t)nk + 8);
    kraw.ensure((size_t)nk * kml * 9 + 8);
    krep.ensure((size_t)nk + 8);
    hipLaunchKernelGGL(bkind_dictionary_kernel, dim3(std::max(1, kBKindSlots / kBlock)), blk, 0, s, B.rowptr, B.col, B.val, keys.ptr,
                       rep.ptr, slot_kid.ptr, kml, koff.ptr, klen.ptr, kraw.ptr, krep.ptr);
    PS_HIP_CHECK(hipGetLastError());
    } // (!verified)
    // the distinct 3x3 blocks, on the host (the dictionary is a few hundred kilobytes at most)
    std::vector<double> hv((size_t)nk * kml * 9);
    std::vector<int> hl((size_t)nk);
    PS_HIP_CHECK(hipMemcpyAsync(hv.data(), kraw.ptr, hv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipMemcpyAsync(hl.data(), klen.ptr, hl.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<double> blk_vals(9, 0.0); // block 0: all zeros (the padding of short block rows)
    std::vector<unsigned short> ids((size_t)nk * kml, 0);
    auto same = [&](const double *a, const double *b) { return std::memcmp(a, b, 9 * sizeof(double)) == 0; };
    for (int k = 0; k < nk; ++k)
        for (int j = 0; j < hl[(size_t)k]; ++j) {
            const double *v = hv.data() + ((size_t)k * kml + j) * 9;
            int id = -1;
            const int nbk = (int)(blk_vals.size() / 9);
            for (int t = 1; t < nbk && id < 0; ++t)
                if (same(v, blk_vals.data() + (size_t)9 * t)) id = t;
            if (id < 0) {
                id = nbk;
                blk_vals.insert(blk_vals.end(), v, v + 9);
            }
            if (id > 1023) return false; // (the kernel packs offset and id into 32 bits)
            ids[(size_t)k * kml + j] = (unsigned short)id;
        }
    const int nblk = (int)(blk_vals.size() / 9);
    if ((size_t)nblk * 80 + (size_t)nk * kml * 4 + 64 > (size_t)kBsrKindLdsBytes) return false;
    {   // the kernel packs (block offset, block id) into 32 bits: offsets of at most 20 bits and a sign
        std::vector<int> ho((size_t)nk * kml);
        PS_HIP_CHECK(hipMemcpyAsync(ho.data(), koff.ptr, ho.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        for (int o : ho)
            if (o <= -(1 << 20) || o >= (1 << 20)) return false;
    }
    blocks.ensure(blk_vals.size() + 8);
    kblk.ensure(ids.size() + 8);
    PS_HIP_CHECK(hipMemcpyAsync(blocks.ptr, blk_vals.data(), blk_vals.size() * sizeof(double), hipMemcpyHostToDevice, s));
    PS_HIP_CHECK(hipMemcpyAsync(kblk.ptr, ids.data(), ids.size() * sizeof(unsigned short), hipMemcpyHostToDevice, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    view.kind = kind.ptr;
    view.koff = koff.ptr;
    view.kblk = kblk.ptr;
    view.blocks = blocks.ptr;
    view.krep = krep.ptr;
    view.nk = nk;
    view.kml = kml;
    view.nblk = nblk;
    valid = true;
    built_nb = B.nb;
    return true;
}

namespace {
// a refactorize under a kept pattern: every row against the CURRENT values of its previous kind's representative (one pass,
// no hashing, no table); all equal -> the kinds stand, only the dictionary's values are taken again
// (round 6, advice: "same pattern" is an equality of 64-bit hashes -- the structure is checked as well, nearly for free: the
// row's length and its pattern id, 2 bytes per row, against the representative's; a mismatch sends the caller to a full build)
__global__ __launch_bounds__(kBlock) void kind_verify_kernel(int n, const int *__restrict__ rowptr, const double *__restrict__ val,
                                                             const unsigned short *__restrict__ kind, const int *__restrict__ krep,
                                                             const int *__restrict__ klen, const unsigned short *__restrict__ pid, int nk,
                                                             int *ctrl)
{
    bool bad = false;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        const int rs = rowptr[r], len = rowptr[r + 1] - rs;
        const int kd = kind[r];
        if (kd >= nk) {
            bad = true;
            continue;
        }
        const int q = krep[kd];
        if (len != klen[kd]) { // (the representative's own row included: the dictionary's stored length)
            bad = true;
            continue;
        }
        if (q == r) continue;
        if (q < 0 || q >= n) {
            bad = true;
            continue;
        }
        const int qs = rowptr[q];
        if (rowptr[q + 1] - qs != len || pid[r] != pid[q]) { // another length or another column-offset pattern than its kind's
            bad = true;
            continue;
        }
        for (int j = 0; j < len; ++j) bad = bad || __double_as_longlong(val[rs + j]) != __double_as_longlong(val[qs + j]);
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) ctrl[0] = 1;
}

__global__ __launch_bounds__(kBlock) void kind_revalue_kernel(int nk, const int *__restrict__ rowptr, const double *__restrict__ val,
                                                              const int *__restrict__ krep, const int *__restrict__ klen, int kml,
                                                              double *kval)
{
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < nk * kml; t += gridDim.x * kBlock) {
        const int kid = t / kml, j = t - kid * kml;
        kval[t] = j < klen[kid] ? val[rowptr[krep[kid]] + j] : 0.0;
    }
}
} // namespace

bool PatMatrix::build_row_table(const Launch &L, int n, const double *v, DeviceBuffer<double> &table)
{
    if (!valid || !view.kind || view.nkind <= 0 || !v || n <= 0) return false;
    table.ensure((size_t)view.nkind + 8);
    ctrl.ensure(8);
    host.ensure(8);
    hipStream_t s = L.stream;
    PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(kind_table_fill_kernel, dim3(std::max(1, kKindSlots / kBlock)), dim3(kBlock), 0, s, vkeys.ptr, vrep.ptr,
                       vslot_kid.ptr, v, table.ptr);
    hipLaunchKernelGGL(kind_table_check_kernel, dim3(L.grid), dim3(kBlock), 0, s, n, kind.ptr, v, table.ptr, ctrl.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    return host.ptr[0] == 0;
}

bool PatMatrix::build_values(const Launch &L, const CsrDev &A, bool same_pattern)
{
    const bool had = same_pattern && view.kind != nullptr && kinds_n == A.n && view.nkind > 0;
    const int nk_prev = view.nkind, kml_prev = view.kml;
    drop_values();
    if (!valid || A.n <= 0 || A.nnz <= 0) return false;
    hipStream_t s = L.stream;
    const dim3 g(L.grid), blk(kBlock);
    int nk = 0;
    const int ml = view.ml, kml = (ml + 7) & ~7;
    bool verified = false;
    if (had && kml_prev == kml) {
        ctrl.ensure(8);
        host.ensure(8);
        PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(kind_verify_kernel, g, blk, 0, s, A.n, A.rowptr, A.val, kind.ptr, krep.ptr, klen.ptr, view.id, nk_prev, ctrl.ptr);
        hipLaunchKernelGGL(kind_revalue_kernel, dim3(std::max(1, (nk_prev * kml + kBlock - 1) / kBlock)), blk, 0, s, nk_prev, A.rowptr, A.val,
                           krep.ptr, klen.ptr, kml, kval.ptr);
        PS_HIP_CHECK(hipGetLastError());
        PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        if (host.ptr[0] == 0) {
            verified = true;
            nk = nk_prev;
        }
    }
    if (!verified) {
    vkeys.ensure(kKindSlots);
    vrep.ensure(kKindSlots);
    vslot_kid.ensure(kKindSlots);
    ctrl.ensure(8);
    host.ensure(8);
    kind.ensure((size_t)A.n + 8);
    PS_HIP_CHECK(hipMemsetAsync(vkeys.ptr, 0, kKindSlots * sizeof(unsigned long long), s));
    PS_HIP_CHECK(hipMemsetAsync(vrep.ptr, 0x7f, kKindSlots * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(ctrl.ptr, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(kind_insert_kernel, g, blk, 0, s, A.n, A.rowptr, A.val, id.ptr, vkeys.ptr, vrep.ptr, ctrl.ptr);
    hipLaunchKernelGGL(kind_number_kernel, dim3(1), blk, 0, s, vkeys.ptr, vslot_kid.ptr, ctrl.ptr);
    hipLaunchKernelGGL(kind_assign_kernel, g, blk, 0, s, A.n, A.rowptr, A.val, id.ptr, vkeys.ptr, vrep.ptr, vslot_kid.ptr,
                       kind.ptr, ctrl.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipMemcpyAsync(host.ptr, ctrl.ptr, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int failed = host.ptr[0];
    nk = host.ptr[2];
    if (failed || nk <= 0 || nk > kKindMax || (size_t)nk * (12 * (size_t)kml + 4) > (size_t)kKindMaxLdsBytes) return false;
    kval.ensure((size_t)nk * kml + 8);
    koff.ensure((size_t)nk * kml + 8);
    klen.ensure((size_t)nk + 8);
    krep.ensure((size_t)nk + 8);
    hipLaunchKernelGGL(kind_dictionary_kernel, dim3(std::max(1, kKindSlots / kBlock)), blk, 0, s, A.rowptr, A.val, id.ptr, off.ptr,
                       vkeys.ptr, vrep.ptr, vslot_kid.ptr, ml, kml, kval.ptr, koff.ptr, klen.ptr, krep.ptr);
    PS_HIP_CHECK(hipGetLastError());
    } // (!verified)
    kinds_n = A.n;
    view.kind = kind.ptr;
    view.kval = kval.ptr;
    view.koff = koff.ptr;
    view.klen = klen.ptr;
    view.nkind = nk;
    view.kml = kml;
    // the slot form (PatDev::scoef), on the host: the dictionary is a few kilobytes
    {
        std::vector<double> hv((size_t)nk * kml);
        std::vector<int> ho((size_t)nk * kml), hl((size_t)nk);
        PS_HIP_CHECK(hipMemcpyAsync(hv.data(), kval.ptr, hv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipMemcpyAsync(ho.data(), koff.ptr, ho.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipMemcpyAsync(hl.data(), klen.ptr, hl.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<int> slots;
        bool ok = true;
        for (int k = 0; k < nk && ok; ++k)
            for (int j = 0; j < hl[(size_t)k] && ok; ++j) {
                const int o = ho[(size_t)k * kml + j];
                if (j > 0 && o <= ho[(size_t)k * kml + j - 1]) ok = false; // entries not in ascending column order
                if (std::find(slots.begin(), slots.end(), o) == slots.end()) slots.push_back(o);
                if ((int)slots.size() > kSlotMax) ok = false;
            }
        if (ok && !slots.empty()) {
            std::sort(slots.begin(), slots.end());
            std::vector<double> hc((size_t)nk * kSlotMax, 0.0);
            std::vector<unsigned> hm((size_t)nk, 0u);
            for (int k = 0; k < nk; ++k)
                for (int j = 0; j < hl[(size_t)k]; ++j) {
                    const int sl = (int)(std::find(slots.begin(), slots.end(), ho[(size_t)k * kml + j]) - slots.begin());
                    hc[(size_t)k * kSlotMax + sl] = hv[(size_t)k * kml + j];
                    hm[(size_t)k] |= 1u << sl;
                }
            scoef.ensure(hc.size() + 8);
            smask.ensure(hm.size() + 8);
            PS_HIP_CHECK(hipMemcpyAsync(scoef.ptr, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice, s));
            PS_HIP_CHECK(hipMemcpyAsync(smask.ptr, hm.data(), hm.size() * sizeof(unsigned), hipMemcpyHostToDevice, s));
            PS_HIP_CHECK(hipStreamSynchronize(s)); // (the host vectors die with this scope)
            view.scoef = scoef.ptr;
            view.smask = smask.ptr;
            view.nslot = (int)slots.size();
            view.sdiag = -1;
            for (int t = 0; t < kSlotMax; ++t) {
                view.soff[t] = t < (int)slots.size() ? slots[(size_t)t] : 0;
                if (t < (int)slots.size() && slots[(size_t)t] == 0) view.sdiag = t;
            }
        }
    }
    return true;
}

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
