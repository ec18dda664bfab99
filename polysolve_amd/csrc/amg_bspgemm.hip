// amg_bspgemm.hip -- numeric Galerkin products on 3 x 3 blocks (round 4).
//
// Block value types (AMGCL_Block<3>, /root/reference/src/polysolve/linear/AMGCL.cpp:243-302) whose operator stores its
// node blocks in full: P, R = P^T, A P and R (A P) consist of full blocks, and their scalar CSR arrays are the block
// patterns expanded (expand_block_csr: block row i of `len` blocks -> scalar row 3 i + r starts at 9 ptr[i] + 3 r len,
// block k's columns at + 3 k).  The scalar product kernel (spgemm_numeric_lds_kernel) looks every scalar b_kj up in the
// parked output row -- 9 look-ups per pair of blocks; here a pair of blocks costs one look-up and 27 multiply-adds, the
// output row is parked as blocks (9 accumulators per slot), and the numbers land in the expanded scalar layout directly.
// For every entry of the output the terms are added in the order of the scalar loop (node k ascending, inside a node the
// three scalar columns in order; no contraction), so the product is the scalar kernel's, bit for bit
// (refresh of configs[2]: A P 11.9 ms, R (A P) 5.4 ms with the scalar kernel).
#include "amg_symbolic.hpp"

namespace psolve {

namespace {

// capacities per wave, by the template parameter CC: blocks of an output row parked at a time = blocks of B staged per
// segment = CC (48: 9.5 KiB of LDS per wave, 16 waves per CU; 32: 6.6 KiB, 24 waves per CU -- chosen by the average row of C),
// a column -> position table of 128 / 64 slots, up to 16 blocks of A per segment
constexpr int kAcap = 16;
// (9.5 KiB of LDS per wave: four workgroups = 16 waves per CU; at 13.6 KiB -- two workgroups -- A P of configs[2] took 13.6
// ms instead of 9.0 with three: the staging rounds are latency, hidden only by other waves)

#define PS_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

template <int CC>
struct BsWave { // one wave's slice of LDS
    static constexpr int kCcap = CC, kScap = CC, kCtab = CC <= 32 ? 64 : 128;
    int ccol[kCcap], ctab[kCtab];
    double cacc[kCcap * 9];
    int sq[kScap], sai[kScap], sslot[kScap];
    double sval[kScap * 9];
    double sa[kAcap * 9];
    int soff[kAcap + 1], sbb[kAcap];
};

// C = A B, one wave per block row of C.
//   A: block CSR (aptr, acol); values 9 per block in aval -- A_T: block p of A is the TRANSPOSE of block amap[p] of aval
//      (R = P^T read out of P's block values);
//   B: block CSR (bptr, bcol); values 9 per block (B_EXP false) or in the expanded scalar layout of its own pattern (true);
//   C: block pattern (cptr, ccol), values written in the expanded scalar layout.
// A first version walked the row of A block by block, every step a chain of dependent loads (column -> extent of the row
// of B -> its blocks): 2 us per step, 13.8 ms for A P of configs[2] (26 steps per row, a million rows) -- slower than the
// scalar kernel.  Now a SEGMENT of the row of A -- as many consecutive blocks as have at most kScap blocks of B between
// them -- is staged in LDS in two rounds of independent loads by all lanes (extents; then columns, values and the
// blocks of A), and the products run out of LDS in the order of the scalar loop: lane l takes block l / 9 of a chunk of
// seven blocks of one row of B, element l % 9 = (r, c).
template <bool A_T, bool B_EXP, int CC>
__global__ __launch_bounds__(kBlock) void bspgemm3_numeric_kernel(int nbr, const int *__restrict__ cptr,
                                                                   const int *__restrict__ ccol, double *__restrict__ cval,
                                                                   const int *__restrict__ aptr, const int *__restrict__ acol,
                                                                   const double *__restrict__ aval,
                                                                   const int *__restrict__ amap,
                                                                   const int *__restrict__ bptr, const int *__restrict__ bcol,
                                                                   const double *__restrict__ bval)
{
    constexpr int kCcap = BsWave<CC>::kCcap, kScap = BsWave<CC>::kScap, kCtab = BsWave<CC>::kCtab;
    __shared__ BsWave<CC> lds[kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = gridDim.x * (kBlock / 64);
    const int jj = lane / 9, e = lane - jj * 9, r = e / 3, c = e - r * 3;
    BsWave<CC> &W = lds[wave];
    auto add_block = [&](int j, double a0, double a1, double a2, double b0, double b1, double b2) {
        unsigned slot = ((unsigned)j * 2654435761u >> 12) & (kCtab - 1);
        int t = W.ctab[slot];
        while (t >= 0 && W.ccol[t] != j) {
            slot = (slot + 1) & (kCtab - 1);
            t = W.ctab[slot];
        }
        if (t >= 0) {
            double s = W.cacc[t * 9 + e];
            s += a0 * b0;
            s += a1 * b1;
            s += a2 * b2;
            W.cacc[t * 9 + e] = s;
        }
    };
    for (int i = blockIdx.x * (kBlock / 64) + wave; i < nbr; i += nwaves) {
        const int cb = cptr[i], ce = cptr[i + 1], clen = ce - cb;
        const int ab = aptr[i], ae = aptr[i + 1];
        for (int c0 = cb; c0 < ce; c0 += kCcap) { // (one pass unless the row of C has more than kCcap blocks)
            const int len = min(kCcap, ce - c0);
            for (int t = lane; t < kCtab; t += 64) W.ctab[t] = -1;
            PS_WAVE_SYNC();
            for (int t = lane; t < len; t += 64) {
                const int col = ccol[c0 + t];
                W.ccol[t] = col;
                unsigned slot = ((unsigned)col * 2654435761u >> 12) & (kCtab - 1);
                while (atomicCAS(&W.ctab[slot], -1, t) != -1) slot = (slot + 1) & (kCtab - 1);
            }
            for (int t = lane; t < len * 9; t += 64) W.cacc[t] = 0.0;
            PS_WAVE_SYNC();
            int ja = ab;
            while (ja < ae) {
                // round 1: extents of the rows of B the next blocks of A point at
                const int na = min(kAcap, ae - ja);
                int bb = 0, bl = 0;
                if (lane < na) {
                    const int k = acol[ja + lane];
                    bb = bptr[k];
                    bl = bptr[k + 1] - bb;
                }
                int incl = bl;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int v = __shfl_up(incl, d);
                    if (lane >= d) incl += v;
                }
                const int m = __popcll(__ballot(lane < na && incl <= kScap));
                if (m == 0) {
                    // one row of B longer than the stage: straight from memory, seven blocks at a time
                    const int src = A_T ? amap[ja] : ja;
                    const double *ap = aval + (size_t)src * 9;
                    const double a0 = A_T ? ap[0 * 3 + r] : ap[r * 3 + 0], a1 = A_T ? ap[1 * 3 + r] : ap[r * 3 + 1],
                                 a2 = A_T ? ap[2 * 3 + r] : ap[r * 3 + 2];
                    const int bb0 = __shfl(bb, 0), bl0 = __shfl(bl, 0);
                    for (int q0 = bb0; q0 < bb0 + bl0; q0 += 7) {
                        const int q = q0 + jj;
                        if (jj < 7 && q < bb0 + bl0) {
                            const double *bp = B_EXP ? bval + (size_t)bb0 * 9 + (size_t)(q - bb0) * 3 + c : bval + (size_t)q * 9 + c;
                            const size_t st = B_EXP ? (size_t)bl0 * 3 : 3;
                            add_block(bcol[q], a0, a1, a2, bp[0], bp[st], bp[2 * st]);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    ja += 1;
                    continue;
                }
                const int T = __shfl(incl, m - 1);
                if (lane < m) {
                    W.soff[lane] = incl - bl;
                    W.sbb[lane] = bb;
                }
                if (lane == 0) W.soff[m] = T;
                PS_WAVE_SYNC();
                // round 2: columns and values of the staged blocks of B, and the blocks of A ([r][t] layout)
                for (int w = lane; w < T; w += 64) {
                    int lo = 0, hi = m; // the last a with soff[a] <= w
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (W.soff[mid] <= w) lo = mid; else hi = mid;
                    }
                    const int q = W.sbb[lo] + (w - W.soff[lo]);
                    W.sai[w] = lo;
                    W.sq[w] = q;
                    const int j = bcol[q];
                    // where block j sits in the parked row (-1: not in this pass), looked up here by all lanes at once
                    // rather than inside the sequential products
                    unsigned slot = ((unsigned)j * 2654435761u >> 12) & (kCtab - 1);
                    int t = W.ctab[slot];
                    while (t >= 0 && W.ccol[t] != j) {
                        slot = (slot + 1) & (kCtab - 1);
                        t = W.ctab[slot];
                    }
                    W.sslot[w] = t;
                }
                for (int v = lane; v < m * 9; v += 64) {
                    const int a = v / 9, e2 = v - a * 9;
                    const int src = A_T ? amap[ja + a] : ja + a;
                    W.sa[v] = A_T ? aval[(size_t)src * 9 + (e2 % 3) * 3 + e2 / 3] : aval[(size_t)src * 9 + e2];
                }
                PS_WAVE_SYNC();
                for (int v = lane; v < T * 9; v += 64) {
                    const int w = v / 9, e2 = v - w * 9;
                    const int q = W.sq[w];
                    if (B_EXP) {
                        const int a = W.sai[w], b0 = W.sbb[a], blen = W.soff[a + 1] - W.soff[a];
                        W.sval[v] = bval[(size_t)b0 * 9 + (size_t)(e2 / 3) * blen * 3 + (size_t)(q - b0) * 3 + e2 % 3];
                    } else {
                        W.sval[v] = bval[(size_t)q * 9 + e2];
                    }
                }
                PS_WAVE_SYNC();
                // the products, in the order of the scalar loop
                for (int a = 0; a < m; ++a) {
                    const double a0 = W.sa[a * 9 + r * 3 + 0], a1 = W.sa[a * 9 + r * 3 + 1], a2 = W.sa[a * 9 + r * 3 + 2];
                    const int w1 = W.soff[a + 1];
                    for (int w0 = W.soff[a]; w0 < w1; w0 += 7) {
                        const int w = w0 + jj;
                        if (jj < 7 && w < w1) {
                            const int t = W.sslot[w];
                            const double b0 = W.sval[w * 9 + c], b1 = W.sval[w * 9 + 3 + c], b2 = W.sval[w * 9 + 6 + c];
                            if (t >= 0) {
                                double sm = W.cacc[t * 9 + e];
                                sm += a0 * b0;
                                sm += a1 * b1;
                                sm += a2 * b2;
                                W.cacc[t * 9 + e] = sm;
                            }
                        }
                        __builtin_amdgcn_wave_barrier(); // (the next chunk / block of A may meet the same slot from another lane)
                    }
                }
                PS_WAVE_SYNC(); // the stage is reused
                ja += m;
            }
            PS_WAVE_SYNC();
            for (int idx = lane; idx < len * 9; idx += 64) {
                const int t = idx / 9, e2 = idx - t * 9, r2 = e2 / 3, c2 = e2 - r2 * 3;
                cval[(size_t)cb * 9 + (size_t)r2 * clen * 3 + (size_t)(c0 - cb + t) * 3 + c2] = W.cacc[idx];
            }
            PS_WAVE_SYNC();
        }
    }
}

} // namespace

void launch_bspgemm3_numeric(const Launch &L, int nbr, const int *cptr, const int *ccol, double *cval_expanded, const int *aptr,
                             const int *acol, const double *aval, const int *amap_transposed, const int *bptr,
                             const int *bcol, const double *bval, bool b_expanded, double avg_c_blocks)
{
    if (nbr <= 0) return;
    // short output rows (A P: a dozen blocks): the smaller slice, six workgroups per CU; else four
    const bool small = avg_c_blocks > 0 && avg_c_blocks <= 20.0;
    const int grid = std::max(1, std::min((small ? 6 : 4) * L.num_cus, (nbr + kBlock / 64 - 1) / (kBlock / 64)));
#define PS_BSPGEMM(AT, BE, CC)                                                                                          \
    hipLaunchKernelGGL((bspgemm3_numeric_kernel<AT, BE, CC>), dim3(grid), dim3(kBlock), 0, L.stream, nbr, cptr, ccol, cval_expanded, \
                       aptr, acol, aval, amap_transposed, bptr, bcol, bval)
#define PS_BSPGEMM2(AT, BE)       \
    do {                          \
        if (small) PS_BSPGEMM(AT, BE, 32); \
        else PS_BSPGEMM(AT, BE, 48);       \
    } while (0)
    if (amap_transposed) {
        if (b_expanded) PS_BSPGEMM2(true, true);
        else PS_BSPGEMM2(true, false);
    } else {
        if (b_expanded) PS_BSPGEMM2(false, true);
        else PS_BSPGEMM2(false, false);
    }
#undef PS_BSPGEMM2
#undef PS_BSPGEMM
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
