// amg_plan.hip -- product plans for the numeric Galerkin products of an AMG refresh (round 5).
//
// A numeric refresh (factorize of a matrix whose sparsity pattern the hierarchy already holds: Newton refactorizes every
// iteration, /root/reference/src/polysolve/nonlinear/descent_strategies/Newton.cpp:189-193; the reference's `pre_factor`
// test, tests/test_linear_solver.cpp:241-307) recomputes A P and R (A P) on patterns that do not change.  The row-wise
// kernels (kernels.hip: spgemm_numeric_lds_kernel, amg_bspgemm.hip) find the destination of every term again at every
// refresh -- output row parked in LDS, a hash table, chains of dependent loads -- and ran at 0.07-0.2 of the HBM stream
// (profiles/r04_refresh.md; the PMC counters of profiles/r05_base_pmc_refresh_*.json: 76 % of the wave cycles parked on
// memory, twice the algorithmic bytes fetched, 0.2-1.4 LDS bank conflicts per LDS cycle).
//
// The PLAN is what the symbolic phase knows and the numeric phase then only has to follow: for every entry e of C the
// list of its terms (ja, jb) -- entry ja of A times entry jb of B -- in the order in which the sequential Gustavson
// product of the host (amg_setup.cpp, the oracle's csr_product: row i of A in stored order, for each its row of B) adds
// them.  A refresh is then a streamed segmented multiply-add
//       C.val[e] = sum_{t in [tp[e], tp[e + 1])} A.val[pa[t]] * B.val[pb[t]]        (one thread per entry, terms in order)
// with no search, no LDS and no dependent loads beyond the two index streams: every output entry gets the same
// multiplications and additions in the same order as the host product, so the refreshed hierarchy stays bit-equal
// (tests/test_gpu_amg.py: test_amg_numeric_refresh_on_same_pattern, test_device_setup_equals_host_hierarchy,
// test_refresh_*).  3 x 3-block hierarchies (AMGCL_Block<3>, /root/reference/src/polysolve/linear/AMGCL.cpp:243-302)
// plan on the BLOCK patterns: a term is a pair of blocks, nine lanes own the nine entries of an output block and add, per
// term, a[r][0] b[0][c], a[r][1] b[1][c], a[r][2] b[2][c] in that order -- the scalar loop's order (node k ascending, inside
// a node its three scalar columns in order).
//
// MEASURED (profiles/r05_refresh.md): slower than the row-wise kernels it was meant to replace -- 256^3 refresh 28.0 -> 35.5 ms,
// configs[2] 33.1 -> 36.2 ms.  One thread per output entry makes every term two scattered 8-byte gathers (A's value, B's value):
// 1.55 G terms at 256^3 = 3.1 G gathers against the ~260 G gathers/s the L2 serves (`box.probe`), where the row-wise walk reads a
// row of B as one contiguous segment.  Kept as an option ("amg.product_plan" 1 / 2, default 0), bit-equal and tested.
//
// The plan is built on the device, once per pattern, at the FIRST refresh ("amg.product_plan" 1; 2: already at
// the first factorize; 0, the default: never) by the Gustavson walk itself: the parked output row hands out, per slot, the next free
// position of that entry's term list -- the k's of a row come in stored order and a row of B holds a column once, so the
// positions are the sequential order.  Cost: 8 bytes per term (256^3 Poisson, level 0: 425 M + 548 M terms = 7.8 GB; configs[2]:
// 118 M + 55 M block terms = 1.4 GB); a level whose plan would not fit a quarter of the free device memory, or whose term count
// passes 2^31, keeps the row-wise kernels.
#include "amg_symbolic.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace psolve {

namespace {

#define PS_WAVE_SYNC_P()                                       \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// One pass of the Gustavson walk over the patterns.  LPR lanes share a row of C: they park its sorted columns in LDS behind
// a small open-addressing table, walk row i of A in stored order and, for entry ja = (i, k), stride row k of B; the term
// (ja, jb) belongs to the slot of column bcol[jb].  FILL false: count the terms of every slot (cnt[e]); FILL true: the slot's
// running position starts at tp[e] and every term is written there.  Rows of C longer than the LDS slot take one lane per
// output entry and a bisection of every row of B (the order is the same: ja ascending).
template <int LPR, bool FILL>
__global__ __launch_bounds__(kBlock) void plan_rows_kernel(int n, const int *__restrict__ cptr, const int *__restrict__ ccol,
                                                            const int *__restrict__ aptr, const int *__restrict__ acol,
                                                            const int *__restrict__ amap, const int *__restrict__ bptr,
                                                            const int *__restrict__ bcol, int *__restrict__ cnt_or_tp,
                                                            int *__restrict__ pa, int *__restrict__ pb)
{
    constexpr int CAP = 8 * LPR, GROUPS = kBlock / LPR, TS = 16 * LPR;
    __shared__ int lcol[GROUPS][CAP];
    __shared__ int lpos[GROUPS][CAP];
    __shared__ int ltab[GROUPS][TS];
    const int lane = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int rows_per_pass = (gridDim.x * kBlock) / LPR;
    int *mycol = lcol[grp], *mypos = lpos[grp], *mytab = ltab[grp];
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / LPR; i < n; i += rows_per_pass) {
        const int cb = cptr[i], ce = cptr[i + 1], len = ce - cb;
        const int ab = aptr[i], ae = aptr[i + 1];
        if (len > CAP) { // (uniform over the row's lanes)
            for (int pos = cb + lane; pos < ce; pos += LPR) {
                const int c = ccol[pos];
                int at = FILL ? cnt_or_tp[pos] : 0;
                for (int ja = ab; ja < ae; ++ja) {
                    const int ca = acol[ja];
                    int lo = bptr[ca], hi = bptr[ca + 1];
                    const int end = hi;
                    while (lo < hi) {
                        const int mid = lo + ((hi - lo) >> 1);
                        if (bcol[mid] < c) lo = mid + 1; else hi = mid;
                    }
                    if (lo < end && bcol[lo] == c) {
                        if (FILL) {
                            pa[at] = amap ? amap[ja] : ja;
                            pb[at] = lo;
                        }
                        ++at;
                    }
                }
                if (!FILL) cnt_or_tp[pos] = at;
            }
            continue;
        }
        for (int t = lane; t < TS; t += LPR) mytab[t] = -1;
        PS_WAVE_SYNC_P();
        for (int t = lane; t < len; t += LPR) {
            const int c = ccol[cb + t];
            mycol[t] = c;
            mypos[t] = FILL ? cnt_or_tp[cb + t] : 0;
            unsigned slot = ((unsigned)c * 2654435761u >> 12) & (TS - 1);
            while (atomicCAS(&mytab[slot], -1, t) != -1) slot = (slot + 1) & (TS - 1);
        }
        PS_WAVE_SYNC_P();
        const int gbase = (threadIdx.x & 63) / LPR * LPR;
        for (int base = ab; base < ae; base += LPR) {
            const int jl = base + lane;
            int bb_l = 0, be_l = 0;
            if (jl < ae) {
                const int ca = acol[jl];
                bb_l = bptr[ca];
                be_l = bptr[ca + 1];
            }
            const int cnt = min(LPR, ae - base);
            for (int t = 0; t < cnt; ++t) {
                const int bb = __shfl(bb_l, gbase + t), be = __shfl(be_l, gbase + t);
                const int ja = base + t;
                for (int jb = bb + lane; jb < be; jb += LPR) {
                    const int j = bcol[jb];
                    unsigned slot = ((unsigned)j * 2654435761u >> 12) & (TS - 1);
                    int s = mytab[slot];
                    while (s >= 0 && mycol[s] != j) {
                        slot = (slot + 1) & (TS - 1);
                        s = mytab[slot];
                    }
                    if (s >= 0) { // (a row of B holds a column once: the lanes of one step touch distinct slots)
                        const int at = mypos[s];
                        mypos[s] = at + 1;
                        if (FILL) {
                            pa[at] = amap ? amap[ja] : ja;
                            pb[at] = jb;
                        }
                    }
                }
                PS_WAVE_SYNC_P(); // the next entry of A's row comes after this one in every slot's list
            }
        }
        if (!FILL)
            for (int t = lane; t < len; t += LPR) cnt_or_tp[cb + t] = mypos[t];
        PS_WAVE_SYNC_P();
    }
}

// C.val[e] = the terms of e, in order.  One thread per output entry; four terms' indices, then their eight values, are in
// flight together; the sum itself is sequential (0 + t0 + t1 + ...: the host product's `c += a * b` from c = 0).
__global__ __launch_bounds__(kBlock) void plan_numeric_kernel(int64_t nc, const int *__restrict__ tp,
                                                               const int *__restrict__ pa, const int *__restrict__ pb,
                                                               const double *__restrict__ aval,
                                                               const double *__restrict__ bval, double *__restrict__ cval)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nc; e += stride) {
        const int t0 = tp[e], t1 = tp[e + 1];
        double s = 0.0;
        int t = t0;
        for (; t + 4 <= t1; t += 4) {
            int ia[4], ib[4];
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ia[q] = __builtin_nontemporal_load(pa + t + q);
                ib[q] = __builtin_nontemporal_load(pb + t + q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = aval[ia[q]];
                b[q] = bval[ib[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s += a[q] * b[q];
        }
        {
            int ia[3], ib[3];
            double a[3], b[3];
            const int rem = t1 - t;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ia[q] = q < rem ? __builtin_nontemporal_load(pa + t + q) : 0;
                ib[q] = q < rem ? __builtin_nontemporal_load(pb + t + q) : 0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a[q] = q < rem ? aval[ia[q]] : 0.0;
                b[q] = q < rem ? bval[ib[q]] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < rem) s += a[q] * b[q];
        }
        __builtin_nontemporal_store(s, cval + e);
    }
}

// 3 x 3 blocks: nine lanes per output block, lane (r, c) owns entry (r, c); 7 output blocks per wave (lane 63 idles).
//   A_T: the A operand of a term is the TRANSPOSE of block pa[t] of aval (R = P^T read out of P's block values);
//   C_EXP: C's values go to the expanded scalar layout -- element (r, c) of output block e at co[e] + r cs[e] + c
//   (expand_block_csr: block row i of `len` blocks -> scalar row 3 i + r starts at 9 ptr[i] + 3 r len) -- else 9 per block.
// Both operands are read 9 per block, row-major.
template <bool A_T, bool C_EXP>
__global__ __launch_bounds__(kBlock) void plan_numeric_block3_kernel(int64_t ncb, const int *__restrict__ tp,
                                                                      const int *__restrict__ pa, const int *__restrict__ pb,
                                                                      const double *__restrict__ aval,
                                                                      const double *__restrict__ bval,
                                                                      double *__restrict__ cval, const int *__restrict__ co,
                                                                      const int *__restrict__ cs)
{
    const int wlane = threadIdx.x & 63, sub = wlane / 9, el = wlane - sub * 9, r = el / 3, c = el - r * 3;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t e0 = wave * 7; e0 < ncb; e0 += nwaves * 7) {
        const int64_t e = e0 + sub;
        if (sub >= 7 || e >= ncb) continue;
        const int t0 = tp[e], t1 = tp[e + 1];
        double s = 0.0;
        int t = t0;
        for (; t + 2 <= t1; t += 2) {
            const int ia0 = pa[t], ib0 = pb[t], ia1 = pa[t + 1], ib1 = pb[t + 1];
            const double *a0 = aval + (int64_t)ia0 * 9, *b0 = bval + (int64_t)ib0 * 9;
            const double *a1 = aval + (int64_t)ia1 * 9, *b1 = bval + (int64_t)ib1 * 9;
            double x0, x1, x2, y0, y1, y2, u0, u1, u2, v0, v1, v2;
            if (A_T) {
                x0 = a0[r], x1 = a0[3 + r], x2 = a0[6 + r];
                u0 = a1[r], u1 = a1[3 + r], u2 = a1[6 + r];
            } else {
                x0 = a0[3 * r], x1 = a0[3 * r + 1], x2 = a0[3 * r + 2];
                u0 = a1[3 * r], u1 = a1[3 * r + 1], u2 = a1[3 * r + 2];
            }
            y0 = b0[c], y1 = b0[3 + c], y2 = b0[6 + c];
            v0 = b1[c], v1 = b1[3 + c], v2 = b1[6 + c];
            s += x0 * y0;
            s += x1 * y1;
            s += x2 * y2;
            s += u0 * v0;
            s += u1 * v1;
            s += u2 * v2;
        }
        if (t < t1) {
            const int ia0 = pa[t], ib0 = pb[t];
            const double *a0 = aval + (int64_t)ia0 * 9, *b0 = bval + (int64_t)ib0 * 9;
            double x0, x1, x2;
            if (A_T) x0 = a0[r], x1 = a0[3 + r], x2 = a0[6 + r];
            else x0 = a0[3 * r], x1 = a0[3 * r + 1], x2 = a0[3 * r + 2];
            const double y0 = b0[c], y1 = b0[3 + c], y2 = b0[6 + c];
            s += x0 * y0;
            s += x1 * y1;
            s += x2 * y2;
        }
        if (C_EXP) cval[(int64_t)co[e] + (int64_t)r * cs[e] + c] = s;
        else cval[e * 9 + el] = s;
    }
}

// offsets of the output blocks in the expanded scalar layout of their pattern
__global__ __launch_bounds__(kBlock) void plan_expanded_offsets_kernel(int nbr, const int *__restrict__ cptr, int *__restrict__ co,
                                                                        int *__restrict__ cs)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
    for (int i = wave; i < nbr; i += nwaves) {
        const int b = cptr[i], len = cptr[i + 1] - b;
        for (int k = lane; k < len; k += 64) {
            co[b + k] = 9 * b + 3 * k;
            cs[b + k] = 3 * len;
        }
    }
}

__global__ __launch_bounds__(kBlock) void plan_compare_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                               unsigned long long *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const unsigned long long x = (unsigned long long)__double_as_longlong(a[i]), y = (unsigned long long)__double_as_longlong(b[i]);
        if (x != y) {
            atomicAdd(&out[0], 1ull);
            atomicMin(&out[1], (unsigned long long)i);
        }
    }
}

} // namespace

// debugging aid ("lab.plan_verbose" 2): how many entries of two value arrays differ bitwise, and the first such index
void plan_compare(const Launch &L, int64_t n, const double *a, const double *b, const char *what)
{
    DeviceBuffer<unsigned long long> out;
    out.ensure(2);
    const unsigned long long init[2] = {0ull, ~0ull};
    PS_HIP_CHECK(hipMemcpyAsync(out.ptr, init, sizeof(init), hipMemcpyHostToDevice, L.stream));
    hipLaunchKernelGGL(plan_compare_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, a, b, out.ptr);
    unsigned long long h[2];
    PS_HIP_CHECK(hipMemcpyAsync(h, out.ptr, sizeof(h), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    double va = 0, vb = 0;
    if (h[0]) {
        PS_HIP_CHECK(hipMemcpy(&va, a + h[1], 8, hipMemcpyDeviceToHost));
        PS_HIP_CHECK(hipMemcpy(&vb, b + h[1], 8, hipMemcpyDeviceToHost));
    }
    std::fprintf(stderr, "[psolve plan check] %s: %llu of %lld entries differ; first at %lld: %.17g vs %.17g\n", what, h[0], (long long)n,
                 h[0] ? (long long)h[1] : -1ll, va, vb);
}

void ProductPlan::reset()
{
    tp.release();
    pa.release();
    pb.release();
    co.release();
    cs.release();
    nc = nterms = 0;
    valid = false;
    tried = false;
}

int g_plan_verbose = 0;

bool device_product_plan(const Launch &L, int n, const int *cptr, const int *ccol, int64_t cnnz, const int *aptr,
                         const int *acol, const int *amap, int64_t annz, const int *bptr, const int *bcol, int nrows_b,
                         int64_t bnnz, bool expanded_offsets, ProductPlan &plan, SymbolicScratch &S)
{
    plan.reset();
    plan.tried = true;
    if (n <= 0 || cnnz <= 0 || annz <= 0 || bnnz <= 0) return false;
    // a bound on the terms before anything is allocated: sum over A's entries of a row of B <= nnz(A) * (longest row of B)
    // is too coarse; the count pass is cheap next to what it decides, so it simply runs -- but its own array (4 (cnnz + 1)
    // bytes) and the plan must fit a quarter of what the device has left
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    if ((size_t)(cnnz + 1) * 4 > free_b / 8) return false;
    const double avg_b_row = nrows_b > 0 ? (double)bnnz / (double)nrows_b : 1.0;
    const double avg_c_row = (double)cnnz / (double)n;
    int lpr = 4;
    while (lpr < 64 && ((double)lpr < avg_b_row || 8.0 * lpr < 1.5 * avg_c_row)) lpr *= 2;
    plan.tp.ensure((size_t)cnnz + 2);
    dim3 g(L.grid), blk(kBlock);
#define PS_PLAN(LPR, FILL, CT)                                                                                              \
    hipLaunchKernelGGL((plan_rows_kernel<LPR, FILL>), g, blk, 0, L.stream, n, cptr, ccol, aptr, acol, amap, bptr, bcol, CT, \
                       plan.pa.ptr, plan.pb.ptr)
#define PS_PLAN_ALL(FILL, CT)             \
    do {                                  \
        if (lpr == 4) PS_PLAN(4, FILL, CT);        \
        else if (lpr == 8) PS_PLAN(8, FILL, CT);   \
        else if (lpr == 16) PS_PLAN(16, FILL, CT); \
        else if (lpr == 32) PS_PLAN(32, FILL, CT); \
        else PS_PLAN(64, FILL, CT);                \
    } while (0)
    PS_PLAN_ALL(false, plan.tp.ptr);
    PS_HIP_CHECK(hipGetLastError());
    int64_t nterms = -1;
    try {
        nterms = device_exclusive_scan(L, plan.tp.ptr, cnnz, S);
    } catch (const Error &e) {
        if (e.code != PSOLVE_HIP_ERANGE) throw;
    }
    if (nterms <= 0) { // more than 2^31 terms: the row-wise kernels stay
        plan.reset();
        plan.tried = true;
        return false;
    }
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)nterms * 8 > free_b / 4) {
        if (g_plan_verbose)
            std::fprintf(stderr, "[psolve] product plan of %lld terms (%.1f MiB) does not fit a quarter of the free device memory: kept off\n",
                         (long long)nterms, (double)nterms * 8 / 1048576.0);
        plan.reset();
        plan.tried = true;
        return false;
    }
    plan.pa.ensure((size_t)nterms + 8);
    plan.pb.ensure((size_t)nterms + 8);
    PS_PLAN_ALL(true, plan.tp.ptr);
    PS_HIP_CHECK(hipGetLastError());
#undef PS_PLAN_ALL
#undef PS_PLAN
    if (expanded_offsets) {
        plan.co.ensure((size_t)cnnz + 2);
        plan.cs.ensure((size_t)cnnz + 2);
        hipLaunchKernelGGL(plan_expanded_offsets_kernel, g, blk, 0, L.stream, n, cptr, plan.co.ptr, plan.cs.ptr);
        PS_HIP_CHECK(hipGetLastError());
    }
    plan.nc = cnnz;
    plan.nterms = nterms;
    plan.valid = true;
    return true;
}

void launch_plan_numeric(const Launch &L, const ProductPlan &plan, const double *aval, const double *bval, double *cval)
{
    if (!plan.valid || plan.nc <= 0) return;
    const int64_t want = (plan.nc + kBlock - 1) / kBlock;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)L.num_cus * 32));
    hipLaunchKernelGGL(plan_numeric_kernel, dim3(grid), dim3(kBlock), 0, L.stream, plan.nc, plan.tp.ptr, plan.pa.ptr, plan.pb.ptr,
                       aval, bval, cval);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_plan_numeric_block3(const Launch &L, const ProductPlan &plan, const double *aval, bool a_transposed,
                                const double *bval, double *cval, bool c_expanded)
{
    if (!plan.valid || plan.nc <= 0) return;
    PS_REQUIRE(!c_expanded || (plan.co.ptr && plan.cs.ptr), PSOLVE_HIP_EINVAL, "product plan: no expanded offsets");
    const int64_t waves = (plan.nc + 6) / 7, want = (waves + 3) / 4;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)L.num_cus * 32));
    dim3 g(grid), blk(kBlock);
#define PS_BPLAN(AT, CE)                                                                                                    \
    hipLaunchKernelGGL((plan_numeric_block3_kernel<AT, CE>), g, blk, 0, L.stream, plan.nc, plan.tp.ptr, plan.pa.ptr, plan.pb.ptr, \
                       aval, bval, cval, plan.co.ptr, plan.cs.ptr)
    if (a_transposed) {
        if (c_expanded) PS_BPLAN(true, true);
        else PS_BPLAN(true, false);
    } else {
        if (c_expanded) PS_BPLAN(false, true);
        else PS_BPLAN(false, false);
    }
#undef PS_BPLAN
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
