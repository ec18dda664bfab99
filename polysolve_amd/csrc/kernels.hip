// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the PCG hot path.
//
// All of them are HBM-bandwidth-bound (fp64 CSR SpMV: 2 flop per 12 B), so there is no MFMA here;
// what matters is 16-byte coalesced streaming of the matrix, keeping the gathered vector in the
// XCD's L2, deterministic reductions without atomics and never synchronising with the host inside
// the CG loop.
//
// Launch geometry shared by every kernel: a PERSISTENT grid of `grid` workgroups (multiple of 8,
// default 8 per CU) x 256 threads.  Workgroup g is observed to run on XCD g % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch"); SpMV uses that only for speed: XCD c sweeps the
// contiguous row range [c, c+1) * ceil(n/8), so the three x-planes a 7-point row block touches stay
// in that XCD's 4 MiB L2 instead of being fetched by all eight.
#include "kernels.hpp"
#include <hip/hip_ext.h>

#include <type_traits>

#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "common.hpp"

// a product / fused vector kernel launch that can carry the caller's timing events (Launch::ev_start / ev_stop); `L` is the
// Launch in scope, the argument list is hipLaunchKernelGGL's
#define PS_TIMED_LAUNCH(kern, grid, block, lds, strm, ...)                                                              \
    do {                                                                                                                \
        if (L.ev_start) hipExtLaunchKernelGGL(kern, grid, block, lds, strm, L.ev_start, L.ev_stop, 0, __VA_ARGS__);      \
        else hipLaunchKernelGGL(kern, grid, block, lds, strm, __VA_ARGS__);                                             \
    } while (0)

namespace psolve {

// The product instantiation launch_spmv chose, as rocprofv3 prints it (the bench reports the kernel of its
// roofline line from HERE instead of composing a name of its own).  Recorded only while a caller asks for it
// (tl_spmv_kernel_record, set by Context around the first product of a solve): a snprintf per launch otherwise.
thread_local char tl_spmv_kernel_name[160] = "";
thread_local int tl_spmv_kernel_record = 0;
thread_local char tl_vec_kernel_name[2][96] = {"", ""}; // ... and of PCG's two vector kernels (update_r, update_xp)
#define PS_NOTE_KERNEL(...)                                                                          \
    do {                                                                                             \
        if (tl_spmv_kernel_record) std::snprintf(tl_spmv_kernel_name, sizeof(tl_spmv_kernel_name), __VA_ARGS__); \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTile = 1856; // nnz products staged through LDS per chunk: 2 x 14.5 KiB tiles => 5 workgroups per CU

// ---------------------------------------------------------------------------------------------
// wave64 / workgroup reductions (deterministic: fixed butterfly + fixed wave order)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
    // butterfly over the 64 lanes; each step is a pair of ds_bpermute_b32 on the two halves of v
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(v));
        int hi = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// every thread returns the workgroup total; sh must hold kBlock/64 doubles
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads(); // sh may still be read from a previous call
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// every workgroup folds the same `np` partials in the same order => bitwise identical scalars
__device__ __forceinline__ double fold_partials(const double *part, int np, double *sh)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += kBlock) s += part[i];
    return block_sum(s, sh);
}

// ---------------------------------------------------------------------------------------------
// CSR SpMV: LDS-staged, software-pipelined row-block stream
// ---------------------------------------------------------------------------------------------
// A row-block is R consecutive rows (R = 256, 128, ... 8, chosen at factorize so that an average
// row-block's nonzeros fit one kTile-entry LDS tile); T = 256 / R threads share a row.
//
// Per row-block, one pass of the loop:
//   A  the (col, val) stream of THIS block -- loaded into registers one iteration ago with fully
//      coalesced 16-byte loads, 4 consecutive nonzeros per thread -- is gathered against x and the
//      products are parked in LDS tile `buf`;
//   -- one workgroup barrier --
//   B  the stream loads of the NEXT block are issued (their HBM latency hides under C);
//   C  every row adds up its slice of the tile.  With T == 1 the adds run in column order with the
//      products already rounded (no FMA across the LDS), i.e. exactly the oracle's scalar CSR loop:
//      y is bit-identical to it.  With T > 1 the T partial sums are combined by a wave butterfly.
// The LDS tile is double-buffered, so one barrier per row-block is enough.  A row-block whose
// nonzeros exceed the tile (rows much longer than the average) streams its remaining chunks through
// the same tile, two barriers per extra chunk.
//
// Why this shape on MI355X (measured, profiles/r01_spmv_lab.md): a pure 16-B read stream of the
// matrix runs at 6.4-6.9 TB/s, but mixing in the y stores costs ~37 % (HBM read/write turnaround; a
// dedicated store wave, nt/sc1 stores or store bursts do not change it) and the x gathers another
// ~10 %; every variant that streams coalesced and keeps >= 4 workgroups per CU resident lands within
// 3 % of the same floor, so the kernel goes for the fewest barriers and the longest load-to-use
// distance.  Non-temporal loads on the matrix stream are NOT used: the two half-line 16-B loads of a
// value pair then miss L1 twice.
// VT = double, or float: the matrix VALUES stored in single precision (8 B instead of 12 B per nonzero),
// products and sums in double all the same -- used for the operators inside the AMG cycle on request.
template <int R, int MODE, typename VT>
__global__ __launch_bounds__(kBlock) void spmv_csr_pipe(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col,
                                                         const VT *__restrict__ val,
                                                         const double *__restrict__ x,
                                                         const double *__restrict__ b, double *__restrict__ y,
                                                         double *__restrict__ partials,
                                                         const int *__restrict__ done_flag, int nrb,
                                                         int rb_per_xcd, int xcd_map, SpmvExtra ex, int np_total)
{
    constexpr int T = kBlock / R;           // threads per row
    constexpr int ROUNDS = (kTile + kBlock * 4 - 1) / (kBlock * 4);
    __shared__ double prod[2][kTile];
    __shared__ double ybuf[R];
    __shared__ double red[kBlock / 64];
    if (done_flag && *done_flag) return;

    const int tid = threadIdx.x;
    const int row_l = tid / T, sub = tid % T;
    // workgroup -> row-block schedule (the XCD workgroup b lands on is observed to be b % 8; used for
    // speed only).  xcd_map 0: row-blocks round-robin over all workgroups.  1: XCD c sweeps the
    // contiguous eighth [c, c+1) * rb_per_xcd.  2 (default): chunks of `chunk` consecutive row-blocks
    // (8192 rows) are dealt round-robin to the XCDs and swept by that XCD's workgroups -- neighbouring
    // rows share an XCD's L2 (x is fetched ~once: HBM reads 1.66 GB vs 1.96 GB round-robin at 256^3)
    // while all XCDs still stream one compact window of the matrix (as fast as round-robin; the
    // contiguous eighths are 7 % slower).  Measured in profiles/r01_spmv_lab.md.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int *__restrict__ rb_list = ex.rb_list;
    if (rb_list) xcd_map = 0; // an explicit row-block list is always walked round-robin
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = rb_list ? ex.n_list
                              : (xcd_map == 1 ? rb_per_xcd
                                              : (xcd_map == 2 ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb));
    const int base = xcd_map == 1 ? xcd * rb_per_xcd : 0;
    int lrb = xcd_map ? slot : (int)blockIdx.x;
    auto rb_of = [&](int l) {
        if (rb_list) return rb_list[l];
        if (xcd_map == 2) return ((l / chunk) * 8 + xcd) * chunk + (l % chunk);
        return base + l;
    };
    double dacc = 0.0, dacc2 = 0.0;

    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs = 0, re = 0, lo = 0, hi = 0;

    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * R, r = row0 + row_l;
        rs_ = 0;
        re_ = 0;
        if (r < n) {
            rs_ = rowptr[r];
            re_ = rowptr[r + 1];
        }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + R, n)];
    };
    // first chunk of a row-block -> registers (entries before lo / after hi are never used)
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * kBlock * 4;
            c[k] = (v4i){0, 0, 0, 0};
            va[k] = (v2d){0.0, 0.0};
            vb[k] = (v2d){0.0, 0.0};
            if (i < hi_ && i - c0 < kTile) {
                if ((int64_t)i + 3 < nnz) {
                    c[k] = *(const v4i *)(col + i);
                    if constexpr (sizeof(VT) == 8) {
                        va[k] = *(const v2d *)(val + i);
                        vb[k] = *(const v2d *)(val + i + 2);
                    } else {
                        const v4f vv = *(const v4f *)(val + i);
                        va[k] = (v2d){(double)vv.x, (double)vv.y};
                        vb[k] = (v2d){(double)vv.z, (double)vv.w};
                    }
                } else { // last few entries of the whole matrix
                    if ((int64_t)i + 0 < nnz) { c[k].x = col[i]; va[k].x = val[i]; }
                    if ((int64_t)i + 1 < nnz) { c[k].y = col[i + 1]; va[k].y = val[i + 1]; }
                    if ((int64_t)i + 2 < nnz) { c[k].z = col[i + 2]; vb[k].x = val[i + 2]; }
                }
            }
        }
    };
    auto row_sum = [&](const double *P, int a, int e, double acc) {
        if (T == 1) {
            int j = a;
            for (; j + 8 <= e; j += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            if (j < e) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = (j + q < e) ? P[j + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j + q < e) acc += t[q];
            }
        } else {
            for (int j = a + sub; j < e; j += T) acc += P[j];
        }
        return acc;
    };

    auto valid = [&](int l) { return l < nloop && (rb_list != nullptr || rb_of(l) < nrb); };
    if (valid(lrb)) {
        load_ptr(rb_of(lrb), rs, re, lo, hi);
        load_stream(lo, hi);
    }
    int buf = 0;
    while (valid(lrb)) {
        const int rb = rb_of(lrb);
        const int c0 = lo & ~3;
        const int lnext = lrb + step;
        const bool has_next = valid(lnext);
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rb_of(lnext), rs_n, re_n, lo_n, hi_n);
        // A: gather + products of the first chunk
        double *P = prod[buf];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int o = tid * 4 + k * kBlock * 4;
            if (o < kTile && c0 + o < hi) {
                v2d p0, p1;
                p0.x = va[k].x * x[c[k].x];
                p0.y = va[k].y * x[c[k].y];
                p1.x = vb[k].x * x[c[k].z];
                p1.y = vb[k].y * x[c[k].w];
                *(v2d *)(P + o) = p0;
                *(v2d *)(P + o + 2) = p1;
            }
        }
        __syncthreads();
        // B: next block's stream
        if (has_next) load_stream(lo_n, hi_n);
        // C: my row's slice of the tile
        double acc = row_sum(P, max(rs, c0) - c0, min(re, c0 + kTile) - c0, 0.0);
        // rows longer than the tile: remaining chunks, the slow way (uniform branch)
        for (int c1 = c0 + kTile; c1 < hi; c1 += kTile) {
            __syncthreads();
            const int cend = min(c1 + kTile, hi);
            for (int i = c1 + tid * 4; i < cend; i += kBlock * 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((int64_t)i + q < nnz) P[i - c1 + q] = (double)val[i + q] * x[col[i + q]];
            }
            __syncthreads();
            acc = row_sum(P, max(rs, c1) - c1, min(re, c1 + kTile) - c1, acc);
        }
        // epilogue
        int r;
        bool mine;
        if (T == 1) {
            r = rb * R + tid;
            mine = r < n;
        } else {
#pragma unroll
            for (int off = T >> 1; off > 0; off >>= 1) {
                int lo32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(acc));
                int hi32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(acc));
                acc += __hiloint2double(hi32, lo32);
            }
            if (sub == 0) ybuf[row_l] = acc;
            __syncthreads();
            r = rb * R + tid;
            mine = tid < R && r < n;
            if (mine) acc = ybuf[tid];
        }
        if (mine) {
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += x[r] * acc;
            } else if (MODE == SPMV_ADD) {
                acc = y[r] + acc;
            } else if (MODE == SPMV_CHEB) {
                // amgcl/relaxation/chebyshev.hpp solve(): residual, scale, axpby(alpha, r, beta, p), x += p
                const double res = ex.dinv[r] * (b[r] - acc);
                const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[r] : ex.alpha * res;
                ex.p[r] = pn;
                acc = x[r] + pn;
            } else if (MODE == SPMV_POWER) {
                acc = ex.dinv[r] * acc;
                dacc += acc * acc;
                dacc2 += fabs(acc * x[r]);
            }
            y[r] = acc;
        }
        lrb = lnext;
        rs = rs_n;
        re = re_n;
        lo = lo_n;
        hi = hi_n;
        buf ^= 1;
    }
    // consumers fold np_total partials (the grid the Launch advertises); this kernel's LDS admits only 5
    // workgroups per CU, so it may run on a smaller grid: clear the slots it does not own
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) {
            ex.partials2[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) ex.partials2[k] = 0.0;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// CSR SpMV, round 2: LDS-DMA staged stream, non-temporal on both sides
// ---------------------------------------------------------------------------------------------
// What round 1 could not get past (profiles/r01_spmv_lab.md): every coalesced variant sat at ~0.345 ms for the
// 256^3 operator, because mixing the 8 % of y stores into the matrix stream cost four times their own time.
// profiles/r02_spmv_lab.md has the way out: the penalty falls from 0.08 ms to 0.03 ms when BOTH the matrix
// stream is loaded non-temporally AND y is stored non-temporally (either one alone changes nothing) -- provided
// an nt load instruction covers whole 128-byte lines, or the second touch of a line misses L1 again (round 1's
// nt attempt).  LDS-DMA gives exactly that access shape: `global_load_lds_dwordx4` moves 16 B per lane,
// 1 KiB contiguous per wave instruction, straight into LDS, with no staging registers at all.
//
// Per row-block (R rows, T = 256 / R threads per row), per chunk of kDmaTile stored entries:
//   1. the chunk's columns and values go global -> LDS by DMA (nt), 6 wave instructions per wave;
//   2. barrier (hipcc drains vmcnt before it);
//   3. thread (row, sub) walks its entries out of LDS -- consecutive lanes read consecutive rows, so the
//      x gathers of one instruction fall on a few lines -- multiplies, and adds in column order: with T == 1
//      exactly the oracle's scalar loop, bit for bit; with T > 1 the same partial sums and butterfly as
//      spmv_csr_pipe;
//   4. epilogue, y stored non-temporally; barrier before the tile is reused.
// Single-buffered on purpose: LDS is what limits the bytes in flight per CU (6 workgroups x 24 KiB), and a second
// tile per workgroup would halve the residency; the overlap comes from the six resident workgroups.
// Round 3: the tile is sized to the operator (dynamic LDS, `tile` entries).  A row-block step takes ~4.6 us whatever
// it moves, so the rate is (bytes in flight per CU) / 4.6 us: wide-row operators, whose row-blocks of 32 rows filled
// only half of a 2048-entry tile (level 1 of the 256^3 hierarchy: 12 KiB of 24), ran at half the residency they could
// have; with a tile of 1280 entries eight workgroups fit a CU instead of six (dma_tile / dma_wg_per_cu below).
constexpr int kDmaTile = 2048; // largest tile: 8 KiB of columns + 16 KiB of values

// tile (entries) for row-blocks of R rows of an operator with `avg` stored entries per row: 25 % head-room over the
// average row-block (12 % left the restriction of the 256^3 hierarchy, whose rows vary between 20 and 40 entries, with
// too many two-chunk row-blocks: 206 -> 240 us), a multiple of 256 (whole DMA wave instructions), at most kDmaTile --
// fuller row-blocks take the multi-chunk path
// ("lab.dma_tile_max", a knob of the handle: the largest tile, for the tests that force multi-pass row-blocks)
constexpr int kRowBlockFill = 2304; // stored entries an average row-block may hold when the row-block height is chosen
static int dma_tile(const LabKnobs &lab, int R, double avg)
{
    const int want = (int)(R * avg * 1.25) + 8;
    return std::max(512, std::min(lab.dma_tile_max, (want + 255) & ~255));
}

int spmv_dma_tile(const LabKnobs &lab, int R, double avg_nnz_per_row) { return dma_tile(lab, R, avg_nnz_per_row); }

void pack_row_blocks(int n, const int *rowptr, int R, int tile_entries, std::vector<int> &starts)
{
    starts.clear();
    const int cap = std::max(16, tile_entries - 4); // (a pass starts at a multiple of four entries)
    int r0 = 0;
    while (r0 < n) {
        int r1 = r0;
        while (r1 < n && r1 - r0 < R && rowptr[r1 + 1] - rowptr[r0] <= cap) ++r1;
        if (r1 == r0) r1 = r0 + 1; // a row longer than the tile: alone, in several passes
        starts.push_back(r0);
        r0 = r1;
    }
    starts.push_back(n);
}

// workgroups of spmv_csr_dma a CU holds: LDS (160 KiB; tile x 12 bytes + ~1 KiB per workgroup), at most 8 (32 waves)
static int dma_wg_per_cu(int tile, int value_bytes)
{
    return std::max(1, std::min(8, (160 * 1024) / (tile * (4 + value_bytes) + 1024)));
}

__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base, bool nt)
{
    // lane l's 16 bytes land at lds_wave_base + 16 l (wave-uniform base + lane * size); aux 2 = nt
    if (nt) __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// streaming accesses of the vector kernels: non-temporal on systems too large for the Infinity Cache (NT), plain
// otherwise; a workgroup step covers 4 KiB contiguous, so every instruction touches whole lines
template <bool NT>
__device__ __forceinline__ v2d load_stream2(const double *p)
{
    if constexpr (NT) return __builtin_nontemporal_load((const v2d *)p);
    else return *(const v2d *)p;
}
template <bool NT>
__device__ __forceinline__ void store_stream2(double *p, v2d v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, (v2d *)p);
    else *(v2d *)p = v;
}

// (a template parameter, not a run-time flag: with `if (nt) nt-store else store` in one function body LLVM merges
// the two stores into one plain store before the flag is ever known)
template <bool NT>
__device__ __forceinline__ void store_stream(double *p, double v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// C16: the column stream is CsrDev::col16 (`col` then points at 16-bit entries, the array padded to a multiple of eight),
// decoded through the row-block's eight window bases: 10 instead of 12 bytes per entry, the same columns in the same order.
// STNT: the stores of the results (y, and p of the Chebyshev step) non-temporal too.  Not the same decision as NT: a level
// operator of 765 MB over vectors of 16 MB streams the matrix past the cache and keeps the vectors in it.
template <int R, int MODE, typename VT, bool NT, bool C16 = false, bool STNT = NT>
__global__ __launch_bounds__(kBlock) void spmv_csr_dma(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, const VT *__restrict__ val,
                                                        const double *__restrict__ x, const double *__restrict__ b,
                                                        double *__restrict__ y, double *__restrict__ partials,
                                                        const int *__restrict__ done_flag, int nrb, int rb_per_xcd,
                                                        int xcd_map, SpmvExtra ex, int tile,
                                                        const int *__restrict__ rb_base = nullptr,
                                                        const int *__restrict__ rb_start = nullptr)
{
    constexpr int T = kBlock / R;
    constexpr int VPL = 16 / (int)sizeof(VT);   // values per lane per DMA instruction (2 doubles / 4 floats)
    constexpr int VPI = 64 * VPL;               // values per wave instruction
    using CT = std::conditional_t<C16, unsigned short, int>;
    extern __shared__ __attribute__((aligned(16))) unsigned char dma_smem[]; // tile columns, then tile values
    CT *lcol = reinterpret_cast<CT *>(dma_smem);
    VT *lval = reinterpret_cast<VT *>(dma_smem + (size_t)tile * sizeof(CT));
    __shared__ double ybuf[T > 1 ? R : 1];
    __shared__ double red[kBlock / 64];
    __shared__ int lbase[8];
    if (done_flag && *done_flag) return;
    auto colof = [&](int j) -> int {
        if constexpr (C16) {
            const unsigned v = lcol[j];
            return lbase[v >> 13] + (int)(v & 8191u);
        } else {
            return lcol[j];
        }
    };

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row_l = tid / T, sub = tid % T;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int *__restrict__ rb_list = ex.rb_list;
    if (rb_list) xcd_map = 0;
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = rb_list ? ex.n_list
                              : (xcd_map == 1 ? rb_per_xcd
                                              : (xcd_map == 2 ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb));
    const int base = xcd_map == 1 ? xcd * rb_per_xcd : 0;
    double dacc = 0.0, dacc2 = 0.0;
    for (int l = xcd_map ? slot : (int)blockIdx.x; l < nloop; l += step) {
        const int rbf = rb_list ? rb_list[l] : (xcd_map == 2 ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : base + l);
        if (rbf >= nrb) continue; // (uniform)
        const int rb = (ex.reverse && !rb_list) ? nrb - 1 - rbf : rbf; // (the same schedule, swept from the last row-block)
        const int row0 = rb_start ? rb_start[rb] : rb * R;
        const int nrows = rb_start ? rb_start[rb + 1] - row0 : min(R, n - row0); // (variable height: CsrDev::rb_start)
        const int lo = rowptr[row0], hi = rowptr[row0 + nrows];
        int rs = 0, re = 0;
        if (row_l < nrows) {
            rs = rowptr[row0 + row_l];
            re = rowptr[row0 + row_l + 1];
        }
        double acc = 0.0, xdiag = 0.0;
        bool have_diag = false;
        if (C16 && tid < 8) lbase[tid] = rb_base[8 * rb + tid]; // (read after the barrier behind the first DMA)
        for (int c1 = lo & ~(C16 ? 7 : 3); c1 < hi; c1 += tile) { // one pass unless rows are much longer than average
            const int cnt = min(hi - c1, tile);
            if constexpr (C16) {
                // columns: 8 per lane, 512 per wave instruction (the tile is a multiple of 512, the array padded)
                const unsigned short *c16 = reinterpret_cast<const unsigned short *>(col);
                for (int e = wave * 512; e < cnt; e += 2048) {
                    const int64_t i = (int64_t)c1 + e + lane * 8;
                    if (i < nnz) dma16(c16 + i, lcol + e, NT);
                }
            } else {
            // columns: 4 per lane, 256 per wave instruction (tile is a multiple of 256)
            for (int e = wave * 256; e < cnt; e += 1024) {
                const int64_t i = (int64_t)c1 + e + lane * 4;
                if (i + 3 < nnz) dma16(col + i, lcol + e, NT);
            }
            }
            // values: VPL per lane
            for (int e = wave * VPI; e < cnt; e += 4 * VPI) {
                const int64_t i = (int64_t)c1 + e + lane * VPL;
                if (i + VPL - 1 < nnz) dma16(val + i, lval + e, NT);
            }
            if ((int64_t)c1 + cnt + 3 >= nnz && tid < 4) { // the last few entries of the whole matrix, by hand
                const int64_t i = (nnz & ~(int64_t)3) + tid;
                if (i < nnz && i >= c1 && i - c1 < tile) {
                    if constexpr (!C16) lcol[i - c1] = col[i];
                    lval[i - c1] = val[i];
                }
            }
            __syncthreads();
            const int a = max(rs, c1) - c1, e_ = min(re, c1 + tile) - c1;
            if (T == 1) {
                // four entries at a time: their gathers are in flight together, the adds stay in column order
                // (the p.q epilogue needs x[row]: it comes by with the diagonal entry's gather)
                const int rme = row0 + tid;
                int j = a;
                for (; j + 4 <= e_; j += 4) {
                    const int c0_ = colof(j), c1_ = colof(j + 1), c2_ = colof(j + 2), c3_ = colof(j + 3);
                    const double v0 = (double)lval[j], v1 = (double)lval[j + 1], v2 = (double)lval[j + 2],
                                 v3 = (double)lval[j + 3];
                    const double x0 = x[c0_], x1 = x[c1_], x2 = x[c2_], x3 = x[c3_];
                    acc += v0 * x0;
                    acc += v1 * x1;
                    acc += v2 * x2;
                    acc += v3 * x3;
                    if (MODE == SPMV_DOT) {
                        if (c0_ == rme) { xdiag = x0; have_diag = true; }
                        if (c1_ == rme) { xdiag = x1; have_diag = true; }
                        if (c2_ == rme) { xdiag = x2; have_diag = true; }
                        if (c3_ == rme) { xdiag = x3; have_diag = true; }
                    }
                }
                for (; j < e_; ++j) {
                    const int cj = colof(j);
                    const double xj = x[cj];
                    acc += (double)lval[j] * xj;
                    if (MODE == SPMV_DOT && cj == rme) { xdiag = xj; have_diag = true; }
                }
            } else {
                // several threads per row: thread `sub` owns entries a + sub, a + sub + T, ...  Four of them per step,
                // all four tile reads first, then all four gathers (in flight together), then the adds in the same
                // order as the one-at-a-time loop -- the same bits, without the dependent chain LDS read -> gather ->
                // add per entry that bounded the wide-row products (profiles/r02_spmv_lab.md section 5)
                // (a thread's entries are those at positions sub, sub + T, ... OF THE ROW, wherever the passes of the tile
                // cut it: the partial sums do not depend on the tile size or on the alignment of its passes)
                const int g_ = max(rs, c1), js = rs + sub + (((g_ - rs - sub + T - 1) & ~(T - 1))) - c1;
                if (!ex.gather4) {
                    for (int j = js; j < e_; j += T) acc += (double)lval[j] * x[colof(j)];
                } else
                for (int j = js; j < e_; j += 4 * T) {
                    const bool k1 = j + T < e_, k2 = j + 2 * T < e_, k3 = j + 3 * T < e_;
                    const int c0_ = colof(j), c1_ = k1 ? colof(j + T) : 0, c2_ = k2 ? colof(j + 2 * T) : 0,
                              c3_ = k3 ? colof(j + 3 * T) : 0;
                    const double v0 = (double)lval[j], v1 = k1 ? (double)lval[j + T] : 0.0,
                                 v2 = k2 ? (double)lval[j + 2 * T] : 0.0, v3 = k3 ? (double)lval[j + 3 * T] : 0.0;
                    const double x0 = x[c0_], x1 = k1 ? x[c1_] : 0.0, x2 = k2 ? x[c2_] : 0.0, x3 = k3 ? x[c3_] : 0.0;
                    acc += v0 * x0;
                    if (k1) acc += v1 * x1;
                    if (k2) acc += v2 * x2;
                    if (k3) acc += v3 * x3;
                }
            }
            __syncthreads(); // the tile is reused by the next chunk / row-block
        }
        int r;
        bool mine;
        if (T == 1) {
            r = row0 + tid;
            mine = tid < nrows;
        } else {
#pragma unroll
            for (int off = T >> 1; off > 0; off >>= 1) {
                int lo32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(acc));
                int hi32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(acc));
                acc += __hiloint2double(hi32, lo32);
            }
            if (sub == 0) ybuf[row_l] = acc;
            __syncthreads();
            r = row0 + tid;
            mine = tid < nrows;
            if (mine) acc = ybuf[tid];
        }
        if (mine) {
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += ((T == 1 && have_diag) ? xdiag : x[r]) * acc;
            } else if (MODE == SPMV_ADD) {
                acc = y[r] + acc;
            } else if (MODE == SPMV_CHEB) {
                const double res = ex.dinv[r] * (b[r] - acc);
                const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[r] : ex.alpha * res;
                store_stream<STNT>(ex.p + r, pn);
                acc = x[r] + pn;
            } else if (MODE == SPMV_POWER) {
                acc = ex.dinv[r] * acc;
                dacc += acc * acc;
                dacc2 += fabs(acc * x[r]);
            }
            store_stream<STNT>(y + r, acc);
        }
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) partials[blockIdx.x] = t;
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) ex.partials2[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// CSR SpMV without the column stream (pattern dictionary, see PatDev)
// ---------------------------------------------------------------------------------------------
// spmv_csr_dma<256> with the 8 KiB column tile gone: the values are staged by LDS-DMA as before, thread t owns row
// row0 + t, and entry j of the row multiplies x[row + off[pattern][j]], the offsets read from a copy of the
// dictionary in LDS (a 7-point grid: 27 patterns, 756 bytes) -- the same number of LDS reads as the column tile
// cost.  16 KiB + the dictionary of LDS per workgroup.
template <int R, int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_csr_pat(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                        const double *__restrict__ val, PatDev P,
                                                        const double *__restrict__ x, const double *__restrict__ b,
                                                        double *__restrict__ y, double *__restrict__ partials,
                                                        const int *__restrict__ done_flag, int nrb, int rb_per_xcd,
                                                        int xcd_map, SpmvExtra ex, int np_total)
{
    constexpr int T = kBlock / R; // lanes per row (R = 256: one thread per row, the scalar loop's sums)
    __shared__ __attribute__((aligned(16))) double lval[kDmaTile];
    __shared__ double ybuf[T > 1 ? R : 1];
    __shared__ double red[kBlock / 64];
    extern __shared__ int ldict[]; // [npat * ml]
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ml = P.ml;
    for (int t = tid; t < P.npat * ml; t += kBlock) ldict[t] = P.off[t];
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int *__restrict__ rb_list = ex.rb_list; // shards: interior / boundary row-blocks
    if (rb_list) xcd_map = 0;
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = rb_list ? ex.n_list
                              : (xcd_map == 1 ? rb_per_xcd
                                              : (xcd_map == 2 ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb));
    const int base = xcd_map == 1 ? xcd * rb_per_xcd : 0;
    double dacc = 0.0, dacc2 = 0.0;
    for (int l = xcd_map ? slot : (int)blockIdx.x; l < nloop; l += step) {
        const int rbf = rb_list ? rb_list[l] : (xcd_map == 2 ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : base + l);
        if (rbf >= nrb) continue; // (uniform)
        const int rb = (ex.reverse && !rb_list) ? nrb - 1 - rbf : rbf; // (the same schedule, swept from the last row-block)
        const int row0 = rb * R, row_l = tid / T, sub = tid % T;
        const int rr = row0 + row_l; // the row this lane works on (T lanes share it)
        const int lo = rowptr[row0], hi = rowptr[min(row0 + R, n)];
        int rs = 0, re = 0, pid = 0;
        if (rr < n) {
            rs = rowptr[rr];
            re = rowptr[rr + 1];
            pid = P.id[rr];
        }
        const int *mo = ldict + pid * ml;
        double acc = 0.0, xdiag = 0.0;
        bool have_diag = false;
        for (int c1 = lo & ~1; c1 < hi; c1 += kDmaTile) { // one pass for rows of up to 8 entries
            const int cnt = min(hi - c1, kDmaTile);
#pragma unroll
            for (int k = 0; k < kDmaTile / 512; ++k) { // values: 2 per lane, 128 per wave instruction
                const int e = (k * 4 + wave) * 128;
                if (e < cnt) {
                    const int64_t i = (int64_t)c1 + e + lane * 2;
                    if (i + 1 < nnz) dma16(val + i, lval + e, NT);
                }
            }
            if ((int64_t)c1 + cnt + 1 >= nnz && tid < 2) { // the last entry or two of the whole matrix, by hand
                const int64_t i = (nnz & ~(int64_t)1) + tid;
                if (i < nnz && i >= c1 && i - c1 < kDmaTile) lval[i - c1] = val[i];
            }
            __syncthreads();
            const int a = max(rs, c1) - c1, e_ = min(re, c1 + kDmaTile) - c1;
            int jj = max(rs, c1) - rs; // position inside the row of the first entry of this pass
            int j = a;
            if (T == 1) {
                const int r = rr;
                // four entries at a time: their gathers are in flight together, the adds stay in column order
                for (; j + 4 <= e_; j += 4, jj += 4) {
                    const int c0_ = r + mo[jj], c1_ = r + mo[jj + 1], c2_ = r + mo[jj + 2], c3_ = r + mo[jj + 3];
                    const double v0 = lval[j], v1 = lval[j + 1], v2 = lval[j + 2], v3 = lval[j + 3];
                    const double x0 = x[c0_], x1 = x[c1_], x2 = x[c2_], x3 = x[c3_];
                    acc += v0 * x0;
                    acc += v1 * x1;
                    acc += v2 * x2;
                    acc += v3 * x3;
                    if (MODE == SPMV_DOT) {
                        if (c0_ == r) { xdiag = x0; have_diag = true; }
                        if (c1_ == r) { xdiag = x1; have_diag = true; }
                        if (c2_ == r) { xdiag = x2; have_diag = true; }
                        if (c3_ == r) { xdiag = x3; have_diag = true; }
                    }
                }
                for (; j < e_; ++j, ++jj) {
                    const int cj = r + mo[jj];
                    const double xj = x[cj];
                    acc += lval[j] * xj;
                    if (MODE == SPMV_DOT && cj == r) { xdiag = xj; have_diag = true; }
                }
            } else {
                // T lanes stride the row; the partial sums meet in the butterfly below (spmv_csr_dma's association)
                for (j = a + sub, jj += sub; j < e_; j += T, jj += T) acc += lval[j] * x[rr + mo[jj]];
            }
            __syncthreads(); // the tile is reused by the next pass / row-block
        }
        int r;
        bool mine;
        if (T == 1) {
            r = rr;
            mine = r < n;
        } else {
#pragma unroll
            for (int off = T >> 1; off > 0; off >>= 1) {
                int lo32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(acc));
                int hi32 = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(acc));
                acc += __hiloint2double(hi32, lo32);
            }
            if (sub == 0) ybuf[row_l] = acc;
            __syncthreads();
            r = row0 + tid;
            mine = tid < R && r < n;
            if (mine) acc = ybuf[tid];
        }
        if (mine) {
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += ((T == 1 && have_diag) ? xdiag : x[r]) * acc;
            } else if (MODE == SPMV_ADD) {
                acc = y[r] + acc;
            } else if (MODE == SPMV_CHEB) {
                const double res = ex.dinv[r] * (b[r] - acc);
                const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[r] : ex.alpha * res;
                store_stream<NT>(ex.p + r, pn);
                acc = x[r] + pn;
            } else if (MODE == SPMV_POWER) {
                acc = ex.dinv[r] * acc;
                dacc += acc * acc;
                dacc2 += fabs(acc * x[r]);
            }
            store_stream<NT>(y + r, acc);
        }
    }
    // the callers fold np_total partial sums (the Launch's SpMV grid): this grid may be a different one
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            if ((int)blockIdx.x < np_total) partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) {
            if ((int)blockIdx.x < np_total) ex.partials2[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) ex.partials2[k] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CSR SpMV of an operator whose rows repeat pattern AND values (row kinds, see PatDev): no matrix stream
// ---------------------------------------------------------------------------------------------
// Thread t owns row row0 + t; its kind's offsets and values are read from LDS (the lanes of a wave mostly share one kind:
// broadcast reads), the entries of x through the caches as in spmv_csr_pat<256>.  No tile, no barrier inside the loop.
// Entry j of the row is kval[kind][j] = val[rowptr[r] + j] bit for bit, times x[r + off[j]], added in row order: the sums
// of spmv_csr_pat<256> / the scalar loop.  HBM bytes: 2 n (kinds) + 8 n (x, once) + 8 n (y) + the mode's vectors.
template <int MODE, bool NT, int U>
__global__ __launch_bounds__(kBlock) void spmv_csr_kind(int n, PatDev P, const double *__restrict__ x,
                                                         const double *__restrict__ b, double *__restrict__ y,
                                                         double *__restrict__ partials, const int *__restrict__ done_flag,
                                                         int nrb, int rb_per_xcd, int xcd_map, SpmvExtra ex, int np_total, int sched, int probe)
{
    // U rows per thread: U independent chains kind -> offsets -> gathers per lane keep U times the loads in flight
    constexpr int R = kBlock;
    __shared__ double red[kBlock / 64];
    extern __shared__ double lkind[]; // [nkind * ml] values | [nkind * ml] offsets | [nkind] lengths
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x;
    const int ml = P.kml, nkm = P.nkind * P.kml; // (the padded stride)
    double *lv = lkind;
    int *lo = reinterpret_cast<int *>(lkind + nkm);
    int *ll = lo + nkm;
    for (int t = tid; t < nkm; t += kBlock) {
        lv[t] = P.kval[t];
        lo[t] = P.koff[t];
    }
    for (int t = tid; t < P.nkind; t += kBlock) ll[t] = P.klen[t];
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int *__restrict__ rb_list = ex.rb_list; // shards: interior / boundary row-blocks (U == 1)
    if (rb_list) xcd_map = 0;
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = rb_list ? ex.n_list
                              : (xcd_map == 1 ? rb_per_xcd
                                              : (xcd_map == 2 ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb));
    const int base = xcd_map == 1 ? xcd * rb_per_xcd : 0;
    // sched 1 (no row-block list): XCD c owns the c-th eighth of the row-blocks and every workgroup of it a contiguous run of
    // that eighth, swept in order -- the entries of x a row-block gathers from the row-blocks next to it (a grid's y
    // neighbours) are then in this CU's L1 from the turn before instead of coming from L2 again
    const bool runs = sched == 1 && !rb_list;
    const int per = (rb_per_xcd + slots - 1) / max(slots, 1);
    const int run0 = xcd * rb_per_xcd + slot * per, run1 = min(min(run0 + per, (xcd + 1) * rb_per_xcd), nrb);
    double dacc = 0.0, dacc2 = 0.0;
    for (int l0 = runs ? 0 : (xcd_map ? slot : (int)blockIdx.x); l0 < (runs ? per : nloop); l0 += U * (runs ? 1 : step)) {
        // U consecutive turns of this workgroup's own schedule together (NOT U neighbouring row-blocks): every lane meets
        // the rows it would meet one at a time, in the same order -- its share of the fused dot product adds up the same
        int r[U], kd[U];
        bool uni = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u * (runs ? 1 : step);
            int rbf = nrb;
            if (runs) rbf = run0 + l < run1 ? run0 + l : nrb;
            else if (l < nloop) rbf = rb_list ? rb_list[l] : (xcd_map == 2 ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : base + l);
            const int rb = (ex.reverse && !rb_list) ? nrb - 1 - rbf : rbf;
            r[u] = (rbf < nrb && rb * R + tid < n) ? rb * R + tid : -1;
            kd[u] = r[u] >= 0 ? (int)P.kind[r[u]] : -1;
            uni = uni && __ballot(kd[u] != __builtin_amdgcn_readfirstlane(kd[u])) == 0;
        }
        double acc[U], xdiag[U];
        bool have_diag[U];
        int ra[U]; // (a lane without a row gathers x[0] and keeps nothing)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u] = 0.0;
            xdiag[u] = 0.0;
            have_diag[u] = false;
            ra[u] = max(r[u], 0);
        }
        // Eight entries of each of the U rows at a time, every gather issued before the first product needs one.  The
        // dictionary rows are padded to a multiple of eight entries with offset 0 / value 0: a padded entry's gather is
        // x[row] (always there) and its product is DROPPED by a select, not added -- the sums of the stored entries only,
        // in row order.
        if (uni) {
            // every wave of an interior row-block: one kind per row-block turn -- offsets and values by scalar loads (no
            // LDS traffic, no per-lane address arithmetic on the dictionary)
            int kd0[U], len0[U], maxlen = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                kd0[u] = __builtin_amdgcn_readfirstlane(kd[u]);
                len0[u] = kd0[u] >= 0 ? P.klen[kd0[u]] : 0;
                kd0[u] = max(kd0[u], 0);
                maxlen = max(maxlen, len0[u]);
            }
            for (int j0 = 0; j0 < maxlen; j0 += 8) {
                double xv[U][8];
                int cj[U][8];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int *__restrict__ so = P.koff + kd0[u] * ml + j0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        cj[u][k] = r[u] >= 0 ? r[u] + ((probe & 1) ? 0 : so[k]) : 0;
                        xv[u][k] = x[cj[u][k]];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double *__restrict__ sv = P.kval + kd0[u] * ml + j0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool on = j0 + k < len0[u];
                        const double t = acc[u] + sv[k] * xv[u][k];
                        acc[u] = on ? t : acc[u];
                        if (MODE == SPMV_DOT) {
                            const bool d = on && cj[u][k] == ra[u];
                            xdiag[u] = d ? xv[u][k] : xdiag[u];
                            have_diag[u] = have_diag[u] || d;
                        }
                    }
                }
            }
        } else {
            int len[U], maxlen = 0;
            const double *mv[U];
            const int *mo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = max(kd[u], 0);
                len[u] = kd[u] >= 0 ? ll[k] : 0;
                mv[u] = lv + k * ml;
                mo[u] = lo + k * ml;
                maxlen = max(maxlen, len[u]);
            }
            for (int j0 = 0; j0 < maxlen; j0 += 8) {
                double xv[U][8];
                int cj[U][8];
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        cj[u][k] = r[u] >= 0 ? r[u] + mo[u][j0 + k] : 0;
                        xv[u][k] = x[cj[u][k]];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool on = j0 + k < len[u];
                        const double t = acc[u] + mv[u][j0 + k] * xv[u][k];
                        acc[u] = on ? t : acc[u];
                        if (MODE == SPMV_DOT) {
                            const bool d = on && cj[u][k] == ra[u];
                            xdiag[u] = d ? xv[u][k] : xdiag[u];
                            have_diag[u] = have_diag[u] || d;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ru = r[u];
            if (ru < 0) continue;
            double a = acc[u];
            if (MODE == SPMV_RESIDUAL) {
                a = b[ru] - a;
                dacc += a * a;
            } else if (MODE == SPMV_DOT) {
                dacc += (have_diag[u] ? xdiag[u] : x[ru]) * a;
            } else if (MODE == SPMV_ADD) {
                a = y[ru] + a;
            } else if (MODE == SPMV_CHEB) {
                const double res = ex.dinv[ru] * (b[ru] - a);
                const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[ru] : ex.alpha * res;
                store_stream<NT>(ex.p + ru, pn);
                a = x[ru] + pn;
            } else if (MODE == SPMV_POWER) {
                a = ex.dinv[ru] * a;
                dacc += a * a;
                dacc2 += fabs(a * x[ru]);
            }
            if (!(probe & 2) || a == 12345.678) store_stream<NT>(y + ru, a);
        }
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            if ((int)blockIdx.x < np_total) partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) {
            if ((int)blockIdx.x < np_total) ex.partials2[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) ex.partials2[k] = 0.0;
        }
    }
}

// ("lab.kind_sched" -1: what was measured best -- 0 for spmv_csr_kind, 118 against 135 us at 256^3, 1 for spmv_csr_slots)

template <int U>
static void launch_spmv_kind_u(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b, double *y,
                               double *partials, const int *done_flag, SpmvExtra ex)
{
    constexpr int R = kBlock;
    const PatDev &P = *A.pat;
    const int nrb = (A.n + R - 1) / R;
    const int rb_per_xcd = (nrb + 7) / 8;
    ex.chunk = std::max(1, L.spmv_chunk_rows / R);
    const int xcd_map = (L.spmv_xcd_map == 2 && (int64_t)nrb < 256ll * ex.chunk) ? 0 : L.spmv_xcd_map;
    // the cache policy goes by what this kernel streams: the kinds and the two vectors
    const int64_t bytes = 18ll * A.n;
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && bytes > L.spmv_nt_bytes);
    dim3 grid(L.spmv_grid), block(kBlock);
    const size_t lds = (size_t)P.nkind * P.kml * 12 + (size_t)P.nkind * 4 + 8;
    PS_NOTE_KERNEL("spmv_csr_kind<%d, %s, %d>", (int)mode, nt ? "true" : "false", U);
#define PS_KIND_CASE(M)                                                                                                  \
    case M:                                                                                                              \
        if (nt)                                                                                                          \
            PS_TIMED_LAUNCH((spmv_csr_kind<M, true, U>), grid, block, lds, L.stream, A.n, P, x, b, y, partials, done_flag,  \
                               nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid, L.lab.kind_sched < 0 ? 0 : L.lab.kind_sched, L.lab.kind_probe);     \
        else                                                                                                             \
            PS_TIMED_LAUNCH((spmv_csr_kind<M, false, U>), grid, block, lds, L.stream, A.n, P, x, b, y, partials, done_flag, \
                               nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid, L.lab.kind_sched < 0 ? 0 : L.lab.kind_sched, L.lab.kind_probe);     \
        break;
    switch (mode) {
        PS_KIND_CASE(SPMV_PLAIN)
        PS_KIND_CASE(SPMV_DOT)
        PS_KIND_CASE(SPMV_RESIDUAL)
        PS_KIND_CASE(SPMV_ADD)
        PS_KIND_CASE(SPMV_CHEB)
        PS_KIND_CASE(SPMV_POWER)
    }
#undef PS_KIND_CASE
}

static void launch_spmv_kind(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b, double *y,
                             double *partials, const int *done_flag, const SpmvExtra &ex)
{
    const int u = L.lab.kind_unroll;
    if (u >= 4) launch_spmv_kind_u<4>(L, A, mode, x, b, y, partials, done_flag, ex);
    else if (u >= 2) launch_spmv_kind_u<2>(L, A, mode, x, b, y, partials, done_flag, ex);
    else launch_spmv_kind_u<1>(L, A, mode, x, b, y, partials, done_flag, ex);
}

// ---------------------------------------------------------------------------------------------
// ... and in the slot form (PatDev::scoef): two rows per lane, 16-byte gathers that do not wait for the kind
// ---------------------------------------------------------------------------------------------
// What bounds spmv_csr_kind is neither bytes (2.5 TB/s of its 18 n) nor the latency of kind -> offsets -> gathers, nor the
// arithmetic: it is the NUMBER of vector-memory instructions (profiles/r05_kind.md: 10 per wave-row -- kind, 8 gathers,
// store; switching the gathers off halves the time, each instruction taken away gives ~8 us back whatever its size).  So:
//   * a lane owns TWO consecutive rows (2 t, 2 t + 1 of a 256-row block handled by 128 lanes): a gather is one 16-byte
//     load for both rows (x[r + o], x[r + 1 + o] are neighbours), the kinds of both rows one 4-byte load, the results one
//     16-byte store -- 4.5 instead of 10 instructions per row;
//   * every lane gathers at ALL of the operator's (at most 8) distinct offsets whatever its kinds, so no load waits for
//     the kinds, and the kinds only pick the coefficients from LDS (slot order = the row's ascending column order; a slot
//     a kind does not have is skipped by a select, not added as 0.0 x): the same sums bit for bit;
//   * the gathers are buffer loads (32-bit byte offset from the vector's descriptor: one VALU add per address).
// Row-blocks within the largest |offset| of an end of the vector take the EDGE path: 8-byte loads, each checked against
// the descriptor's range on its own (a boundary row's missing neighbour may lie outside the vector: it reads 0 there and
// is not used), plain 8-byte loads and stores for the other vectors.
typedef unsigned slot_u2 __attribute__((ext_vector_type(2)));
typedef unsigned slot_u4 __attribute__((ext_vector_type(4)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));

struct SlotTurn {
    double x0[kSlotMax], x1[kSlotMax]; // the gathers of row r and of row r + 1
    double e0[2], e1[2], e2[2];        // the mode's other vectors at r, r + 1
    unsigned kk;                       // kind of r | kind of r + 1 << 16
};

template <int MODE>
__device__ __forceinline__ void slot_issue(SlotTurn &t, int r, __amdgpu_buffer_rsrc_t xrs, const PatDev &P,
                                           const double *__restrict__ b, const double *__restrict__ y, const SpmvExtra &ex,
                                           int probe)
{
    // r: the lane's first row (even) of a row-block away from the vector's ends, or -1 (then row 0's data, unused)
    const int ra = max(r, 0);
    const unsigned r8 = (unsigned)ra << 3;
    t.kk = (probe & 4) ? 0x000d000du : *reinterpret_cast<const unsigned *>(P.kind + ra); // (the array is padded)
#pragma unroll
    for (int s = 0; s < kSlotMax; ++s) {
        if (s < P.nslot) { // (uniform)
            const unsigned o = r8 + ((unsigned)P.soff[s] << 3);
            const slot_u4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)((probe & 1) ? r8 : o), 0, 0);
            t.x0[s] = __hiloint2double((int)v.y, (int)v.x);
            t.x1[s] = __hiloint2double((int)v.w, (int)v.z);
        } else {
            t.x0[s] = t.x1[s] = 0.0;
        }
    }
    t.e0[0] = t.e0[1] = t.e1[0] = t.e1[1] = t.e2[0] = t.e2[1] = 0.0;
    auto pair = [&](const double *__restrict__ v, double *o) {
        const v2d_a8 q = *reinterpret_cast<const v2d_a8 *>(v + ra);
        o[0] = q.x;
        o[1] = q.y;
    };
    if (MODE == SPMV_RESIDUAL || MODE == SPMV_CHEB) pair(b, t.e0);
    if (MODE == SPMV_ADD) pair(y, t.e0);
    if (MODE == SPMV_CHEB || MODE == SPMV_POWER) pair(ex.dinv, t.e1);
    if (MODE == SPMV_CHEB && ex.beta != 0.0) pair(ex.p, t.e2);
}

template <int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_csr_slots(int n, int nx, PatDev P, const double *__restrict__ x,
                                                          const double *__restrict__ b, double *__restrict__ y,
                                                          double *__restrict__ partials, const int *__restrict__ done_flag,
                                                          int nrb, int rb_per_xcd, int xcd_map, SpmvExtra ex, int np_total, int sched,
                                                          int probe)
{
    constexpr int R = kBlock;
    __shared__ double red[kBlock / 64];
    extern __shared__ double lslot[]; // [nkind * kSlotMax] coefficients | [nkind] masks
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x, half = tid >> 7, t2 = (tid & 127) * 2;
    const int nkc = P.nkind * kSlotMax;
    unsigned *lm = reinterpret_cast<unsigned *>(lslot + nkc);
    for (int t = tid; t < nkc; t += kBlock) lslot[t] = P.scoef[t];
    for (int t = tid; t < P.nkind; t += kBlock) lm[t] = P.smask[t];
    __syncthreads();
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int *__restrict__ rb_list = ex.rb_list;
    if (rb_list) xcd_map = 0;
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = rb_list ? ex.n_list
                              : (xcd_map == 1 ? rb_per_xcd
                                              : (xcd_map == 2 ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb));
    const int base = xcd_map == 1 ? xcd * rb_per_xcd : 0;
    const bool runs = sched == 1 && !rb_list; // (spmv_csr_kind's schedules)
    const int per = (rb_per_xcd + slots - 1) / max(slots, 1);
    const int run0 = xcd * rb_per_xcd + slot * per, run1 = min(min(run0 + per, (xcd + 1) * rb_per_xcd), nrb);
    const int l_first = runs ? 0 : (xcd_map ? slot : (int)blockIdx.x), l_end = runs ? per : nloop, l_step = runs ? 1 : step;
    const int omin = min(P.soff[0], 0), omax = max(P.soff[max(P.nslot - 1, 0)], 0); // (ascending)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, (int)((unsigned)nx * 8u), 0x00020000);
    // the first row of schedule entry l's row-block, or -1 (uniform per half-workgroup: two waves share an entry)
    auto row0_of = [&](int l) -> int {
        if (l >= l_end) return -1;
        int rbf;
        if (runs) rbf = run0 + l < run1 ? run0 + l : nrb;
        else rbf = rb_list ? rb_list[l] : (xcd_map == 2 ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : base + l);
        if (rbf >= nrb) return -1;
        const int rb = (ex.reverse && !rb_list) ? nrb - 1 - rbf : rbf;
        return rb * R;
    };
    // EDGE row-blocks -- within the largest |offset| of an end of the vector, or the last, partial one -- are left out of
    // the pipelined loop (row0_fast gives -1 for them) and done afterwards, one row per lane, by checked 8-byte loads
    auto is_edge = [&](int row0) -> bool { return row0 + omin < 0 || row0 + R + omax + 1 > nx || row0 + R > n; };
    auto row0_fast = [&](int l) -> int {
        const int row0 = row0_of(l);
        return (row0 >= 0 && !is_edge(row0)) ? row0 : -1;
    };
    auto issue = [&](SlotTurn &t, int row0, int r) { slot_issue<MODE>(t, r, xrs, P, b, y, ex, probe); };
    double dacc = 0.0, dacc2 = 0.0;
    auto sums = [&](const SlotTurn &cur, int row0, int r) {
        if (row0 < 0) return; // (uniform per wave; every row of the block exists: row0 + R <= n)
        const int k0 = (int)(cur.kk & 0xffffu), k1 = (int)(cur.kk >> 16);
        const unsigned m0 = lm[k0], m1 = lm[k1];
        const double *c0 = lslot + k0 * kSlotMax, *c1 = lslot + k1 * kSlotMax;
        double a0 = 0.0, a1 = 0.0, xr0 = 0.0, xr1 = 0.0;
        const unsigned mu = __builtin_amdgcn_readfirstlane(m0);
        if (__ballot(m0 != mu || m1 != mu) == 0) {
            // every row of the wave has the same slots (an interior wave): no selects, uniform skips
#pragma unroll
            for (int s = 0; s < kSlotMax; ++s)
                if ((mu >> s) & 1u) {
                    a0 += c0[s] * cur.x0[s];
                    a1 += c1[s] * cur.x1[s];
                }
        } else {
#pragma unroll
            for (int s = 0; s < kSlotMax; ++s) {
                const double u0 = a0 + c0[s] * cur.x0[s], u1 = a1 + c1[s] * cur.x1[s];
                a0 = ((m0 >> s) & 1u) ? u0 : a0;
                a1 = ((m1 >> s) & 1u) ? u1 : a1;
            }
        }
        if (MODE == SPMV_DOT || MODE == SPMV_CHEB || MODE == SPMV_POWER) {
            if (P.sdiag >= 0) { // x[r]: the gather of the slot of offset 0 (made whether or not the row stores a diagonal)
#pragma unroll
                for (int s = 0; s < kSlotMax; ++s)
                    if (s == P.sdiag) {
                        xr0 = cur.x0[s];
                        xr1 = cur.x1[s];
                    }
            } else {
                xr0 = x[r];
                xr1 = x[r + 1];
            }
        }
        if (MODE == SPMV_RESIDUAL) {
            a0 = cur.e0[0] - a0;
            a1 = cur.e0[1] - a1;
            dacc += a0 * a0;
            dacc += a1 * a1;
        } else if (MODE == SPMV_DOT) {
            dacc += xr0 * a0;
            dacc += xr1 * a1;
        } else if (MODE == SPMV_ADD) {
            a0 = cur.e0[0] + a0;
            a1 = cur.e0[1] + a1;
        } else if (MODE == SPMV_CHEB) {
            const double res0 = cur.e1[0] * (cur.e0[0] - a0), res1 = cur.e1[1] * (cur.e0[1] - a1);
            const double p0 = (ex.beta != 0.0) ? ex.alpha * res0 + ex.beta * cur.e2[0] : ex.alpha * res0;
            const double p1 = (ex.beta != 0.0) ? ex.alpha * res1 + ex.beta * cur.e2[1] : ex.alpha * res1;
            const v2d pv = {p0, p1};
            store_stream2<NT>(ex.p + r, pv);
            a0 = xr0 + p0;
            a1 = xr1 + p1;
        } else if (MODE == SPMV_POWER) {
            a0 = cur.e1[0] * a0;
            a1 = cur.e1[1] * a1;
            dacc += a0 * a0;
            dacc2 += fabs(a0 * xr0);
            dacc += a1 * a1;
            dacc2 += fabs(a1 * xr1);
        }
        if ((probe & 2) && a0 != 12345.678) return;
        const v2d yv = {a0, a1};
        store_stream2<NT>(y + r, yv);
    };
    // A trip takes two schedule entries per register set -- one per half-workgroup (128 lanes x 2 rows = a 256-row block) --
    // and two register sets take turns (a copy `cur = next` at the end of a trip would make the compiler wait for the loads
    // it has just issued): while one set's products are added, the other's loads are under way.
    int l = l_first + half * l_step;
    int row0a = row0_fast(l);
    int ra = row0a >= 0 ? row0a + t2 : -1;
    SlotTurn A, B;
    issue(A, row0a, ra);
    while (l - half * l_step < l_end) {
        const int row0b = row0_fast(l + 2 * l_step);
        const int rb_ = row0b >= 0 ? row0b + t2 : -1;
        issue(B, row0b, rb_); // (an entry that is none, or an edge's: row 0's data, unused)
        sums(A, row0a, ra);
        l += 4 * l_step;
        row0a = row0_fast(l);
        ra = row0a >= 0 ? row0a + t2 : -1;
        issue(A, row0a, ra);
        sums(B, row0b, rb_);
    }
    // The edge row-blocks: one row per lane, every index checked.  They sit at the two ends of the row range (a grid's
    // first and last planes) -- left to the workgroups whose schedule they fall into, a few workgroups would each do dozens
    // of these slow turns after everybody else has finished; dealt round-robin over ALL workgroups it is one turn each.
    // (A row-block list -- a shard's interior / boundary rows -- is walked as listed.)
    const int e_lo = min(nrb, (-omin + R - 1) / R);                                  // row-blocks [0, e_lo)
    const int hi_num = nx - omax - 1 - R;
    const int first_hi = max(e_lo, min(nrb, hi_num < 0 ? 0 : hi_num / R + 1));      // ... and [e_hi0, nrb): at least a partial last one
    const int e_hi0 = max(e_lo, min(first_hi, (n % R) ? nrb - 1 : nrb));
    const int n_edge = rb_list ? 0 : e_lo + (nrb - e_hi0);
    for (int le = rb_list ? l_first : (int)blockIdx.x; le < (rb_list ? l_end : n_edge); le += rb_list ? l_step : (int)gridDim.x) {
        int row0;
        if (rb_list) {
            row0 = row0_of(le);
            if (row0 < 0 || !is_edge(row0)) continue; // (uniform)
        } else {
            row0 = (le < e_lo ? le : e_hi0 + (le - e_lo)) * R;
            if (!is_edge(row0)) continue; // (cannot happen: the two ranges are the edge row-blocks)
        }
        const int r = row0 + tid;
        if (r >= n) continue;
        const int kd = (int)P.kind[r];
        const unsigned m = lm[kd];
        const double *co = lslot + kd * kSlotMax;
        double acc = 0.0;
        for (int sl = 0; sl < P.nslot; ++sl) {
            const int i = r + P.soff[sl];
            if (((m >> sl) & 1u) && i >= 0 && i < nx) acc += co[sl] * x[i];
        }
        if (MODE == SPMV_RESIDUAL) {
            acc = b[r] - acc;
            dacc += acc * acc;
        } else if (MODE == SPMV_DOT) {
            dacc += x[r] * acc;
        } else if (MODE == SPMV_ADD) {
            acc = y[r] + acc;
        } else if (MODE == SPMV_CHEB) {
            const double res = ex.dinv[r] * (b[r] - acc);
            const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[r] : ex.alpha * res;
            store_stream<NT>(ex.p + r, pn);
            acc = x[r] + pn;
        } else if (MODE == SPMV_POWER) {
            acc = ex.dinv[r] * acc;
            dacc += acc * acc;
            dacc2 += fabs(acc * x[r]);
        }
        store_stream<NT>(y + r, acc);
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            if ((int)blockIdx.x < np_total) partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) {
            if ((int)blockIdx.x < np_total) ex.partials2[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) ex.partials2[k] = 0.0;
        }
    }
}


static void launch_spmv_slots(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b, double *y,
                              double *partials, const int *done_flag, SpmvExtra ex)
{
    constexpr int R = kBlock;
    const PatDev &P = *A.pat;
    const int nrb = (A.n + R - 1) / R;
    const int rb_per_xcd = (nrb + 7) / 8;
    ex.chunk = std::max(1, L.spmv_chunk_rows / R);
    const int xcd_map = (L.spmv_xcd_map == 2 && (int64_t)nrb < 256ll * ex.chunk) ? 0 : L.spmv_xcd_map;
    const int64_t bytes = 18ll * A.n;
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && bytes > L.spmv_nt_bytes);
    dim3 grid(L.spmv_grid), block(kBlock);
    const size_t lds = (size_t)P.nkind * kSlotMax * 8 + (size_t)P.nkind * 4 + 8;
    const int nx = std::max(A.n_ext, A.n);
    PS_NOTE_KERNEL("spmv_csr_slots<%d, %s>", (int)mode, nt ? "true" : "false");
#define PS_SLOT_CASE(M)                                                                                                  \
    case M:                                                                                                              \
        if (nt)                                                                                                          \
            PS_TIMED_LAUNCH((spmv_csr_slots<M, true>), grid, block, lds, L.stream, A.n, nx, P, x, b, y, partials, done_flag, \
                               nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid, L.lab.kind_sched < 0 ? 1 : L.lab.kind_sched, L.lab.kind_probe); \
        else                                                                                                             \
            PS_TIMED_LAUNCH((spmv_csr_slots<M, false>), grid, block, lds, L.stream, A.n, nx, P, x, b, y, partials, done_flag, \
                               nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid, L.lab.kind_sched < 0 ? 1 : L.lab.kind_sched, L.lab.kind_probe); \
        break;
    switch (mode) {
        PS_SLOT_CASE(SPMV_PLAIN)
        PS_SLOT_CASE(SPMV_DOT)
        PS_SLOT_CASE(SPMV_RESIDUAL)
        PS_SLOT_CASE(SPMV_ADD)
        PS_SLOT_CASE(SPMV_CHEB)
        PS_SLOT_CASE(SPMV_POWER)
    }
#undef PS_SLOT_CASE
}

// (round 5's spmv_csr_ring -- the near gathers served from a ring of x in LDS -- measured no faster than spmv_csr_slots:
// 110 us against 90 us per product at 256^3, profiles/r05_kind.md section 4; removed from the library in round 6)

template <int R>
static void launch_spmv_pat_r(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b,
                              double *y, double *partials, const int *done_flag, SpmvExtra ex)
{
    const int nrb = (A.n + R - 1) / R;
    const int rb_per_xcd = (nrb + 7) / 8;
    ex.chunk = std::max(1, L.spmv_chunk_rows / R);
    const int xcd_map = (L.spmv_xcd_map == 2 && (int64_t)nrb < 256ll * ex.chunk) ? 0 : L.spmv_xcd_map;
    // the cache policy goes by what this kernel streams (8 nnz + 22 n), the rule is launch_spmv_r's
    const int64_t bytes = A.nnz * 8ll + 22ll * A.n;
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && bytes > L.spmv_nt_bytes);
    dim3 grid(L.spmv_grid), block(kBlock);
    const size_t dict_bytes = (size_t)A.pat->npat * A.pat->ml * sizeof(int);
    PS_NOTE_KERNEL("spmv_csr_pat<%d, %d, %s>", R, (int)mode, nt ? "true" : "false");
#define PS_PAT_CASE(M)                                                                                             \
    case M:                                                                                                        \
        if (nt)                                                                                                    \
            PS_TIMED_LAUNCH((spmv_csr_pat<R, M, true>), grid, block, dict_bytes, L.stream, A.n, A.nnz, A.rowptr, A.val, \
                               *A.pat, x, b, y, partials, done_flag, nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid);   \
        else                                                                                                       \
            PS_TIMED_LAUNCH((spmv_csr_pat<R, M, false>), grid, block, dict_bytes, L.stream, A.n, A.nnz, A.rowptr, A.val, \
                               *A.pat, x, b, y, partials, done_flag, nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid);   \
        break;
    switch (mode) {
        PS_PAT_CASE(SPMV_PLAIN)
        PS_PAT_CASE(SPMV_DOT)
        PS_PAT_CASE(SPMV_RESIDUAL)
        PS_PAT_CASE(SPMV_ADD)
        PS_PAT_CASE(SPMV_CHEB)
        PS_PAT_CASE(SPMV_POWER)
    }
#undef PS_PAT_CASE
}

static void launch_spmv_pat(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b,
                            double *y, double *partials, const int *done_flag, const SpmvExtra &ex)
{
    // (a row-block list -- a shard's interior / boundary rows -- speaks of blocks of A.rows_per_block rows: the kind kernels'
    // own 256-row blocks serve it only where the two agree)
    if (A.pat->kind && (A.rows_per_block == kBlock || !ex.rb_list) && L.spmv_kernel != 3) { // ("spmv_kernel" 3: the dictionary kernel, by name)
        const bool slots_ok = A.pat->nslot > 0 && L.lab.kind_slots && (int64_t)std::max(A.n_ext, A.n) < (1ll << 29);
        if (slots_ok) launch_spmv_slots(L, A, mode, x, b, y, partials, done_flag, ex);
        else launch_spmv_kind(L, A, mode, x, b, y, partials, done_flag, ex);
        return;
    }
    switch (A.rows_per_block) {
    case 256: launch_spmv_pat_r<256>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    case 128: launch_spmv_pat_r<128>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    default: launch_spmv_pat_r<64>(L, A, mode, x, b, y, partials, done_flag, ex); break; // (rows of up to 32 entries)
    }
}

// ---------------------------------------------------------------------------------------------
// SELL-64-sigma SpMV (wide rows: coarse AMG levels, elasticity as CSR)
// ---------------------------------------------------------------------------------------------
// With 30-80 entries per row the row-block kernels above are latency-bound: a step is load tile -> barrier ->
// (LDS read -> gather -> add) per entry -> barrier, and the phases of a workgroup do not overlap
// (profiles/r02_spmv_lab.md section 5: 82 % of wave cycles waiting at 50 % of the HBM rate).  Here a wave owns a
// slice of 64 rows stored column-major: per step of 8 positions it issues 16 whole-line loads (columns, values),
// then 8 gathers, then 8 adds in column order -- no LDS, no barrier, and every lane's sum is the scalar loop's.
template <int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_sell_kernel(int n, SellDev S, const double *__restrict__ x,
                                                            const double *__restrict__ b, double *__restrict__ y,
                                                            double *__restrict__ partials,
                                                            const int *__restrict__ done_flag, int xcd_map, SpmvExtra ex)
{
    __shared__ double red[kBlock / 64];
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int ngroups = (S.nslices + 3) >> 2; // a workgroup step: 4 slices = 256 rows
    const int chunk = ex.chunk > 0 ? ex.chunk : 1;
    const int step = xcd_map ? slots : (int)gridDim.x;
    const int nloop = xcd_map ? (((ngroups + chunk - 1) / chunk + 7) / 8) * chunk : ngroups;
    double dacc = 0.0, dacc2 = 0.0;
    for (int l = xcd_map ? slot : (int)blockIdx.x; l < nloop; l += step) {
        const int g = xcd_map ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : l;
        const int s = g * 4 + wave;
        if (s >= S.nslices) continue; // (wave-uniform)
        const int base = S.slice_ptr[s], w = (S.slice_ptr[s + 1] - base) >> 6;
        const int2 me = S.slot[s * 64 + lane];
        const int r = me.x, len = me.y;
        const int *__restrict__ cp = S.col + base + lane;
        const double *__restrict__ vp = S.val + base + lane;
        double acc = 0.0;
        // software pipeline: the stream loads of step j + 1 are issued before the gathers of step j
        int c[8], cn[8];
        double v[8], vn[8];
        auto load8 = [&](int j, int *cc, double *vv) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (j + k < w) { // (wave-uniform)
                    if constexpr (NT) {
                        cc[k] = __builtin_nontemporal_load(cp + (j + k) * 64);
                        vv[k] = __builtin_nontemporal_load(vp + (j + k) * 64);
                    } else {
                        cc[k] = cp[(j + k) * 64];
                        vv[k] = vp[(j + k) * 64];
                    }
                } else {
                    cc[k] = 0;
                    vv[k] = 0.0;
                }
            }
        };
        load8(0, c, v);
        for (int j = 0; j < w; j += 8) {
            if (j + 8 < w) load8(j + 8, cn, vn);
            double xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = (j + k < len) ? x[c[k]] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (j + k < len) acc += v[k] * xv[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c[k] = cn[k];
                v[k] = vn[k];
            }
        }
        if (r >= 0) {
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += x[r] * acc;
            } else if (MODE == SPMV_ADD) {
                acc = y[r] + acc;
            } else if (MODE == SPMV_CHEB) {
                const double res = ex.dinv[r] * (b[r] - acc);
                const double pn = (ex.beta != 0.0) ? ex.alpha * res + ex.beta * ex.p[r] : ex.alpha * res;
                ex.p[r] = pn;
                acc = x[r] + pn;
            } else if (MODE == SPMV_POWER) {
                acc = ex.dinv[r] * acc;
                dacc += acc * acc;
                dacc2 += fabs(acc * x[r]);
            }
            y[r] = acc;
        }
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL || MODE == SPMV_POWER) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) partials[blockIdx.x] = t;
    }
    if (MODE == SPMV_POWER) {
        const double t = block_sum(dacc2, red);
        if (tid == 0) ex.partials2[blockIdx.x] = t;
    }
}

static void launch_spmv_sell(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b,
                             double *y, double *partials, const int *done_flag, SpmvExtra ex)
{
    const SellDev &S = *A.sell;
    const int ngroups = (S.nslices + 3) / 4;
    ex.chunk = std::max(1, L.spmv_chunk_rows / 256);
    const int xcd_map = (L.spmv_xcd_map == 2 && (int64_t)ngroups >= 256ll * ex.chunk) ? 2 : 0;
    const int64_t bytes = A.nnz * 12ll + 20ll * A.n;
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && bytes > L.spmv_nt_bytes);
    // persistent grid where per-workgroup partial sums are folded later (their count is the Launch's); one
    // workgroup per step otherwise: with only ~5 steps per resident workgroup a persistent grid loses the
    // remainder round (5.16 steps of work take 6), the dispatcher's refill does not
    const bool reduces = mode == SPMV_DOT || mode == SPMV_RESIDUAL || mode == SPMV_POWER;
    const int nsteps = xcd_map ? (((ngroups + ex.chunk - 1) / ex.chunk + 7) / 8) * ex.chunk * 8 : ngroups;
    dim3 grid(reduces ? L.spmv_grid : std::max(8, nsteps)), block(kBlock);
    PS_NOTE_KERNEL("spmv_sell_kernel<%d, %s>", (int)mode, nt ? "true" : "false");
#define PS_SELL_CASE(M)                                                                                            \
    case M:                                                                                                        \
        if (nt)                                                                                                    \
            PS_TIMED_LAUNCH((spmv_sell_kernel<M, true>), grid, block, 0, L.stream, A.n, S, x, b, y, partials,   \
                               done_flag, xcd_map, ex);                                                            \
        else                                                                                                       \
            PS_TIMED_LAUNCH((spmv_sell_kernel<M, false>), grid, block, 0, L.stream, A.n, S, x, b, y, partials,  \
                               done_flag, xcd_map, ex);                                                            \
        break;
    switch (mode) {
        PS_SELL_CASE(SPMV_PLAIN)
        PS_SELL_CASE(SPMV_DOT)
        PS_SELL_CASE(SPMV_RESIDUAL)
        PS_SELL_CASE(SPMV_ADD)
        PS_SELL_CASE(SPMV_CHEB)
        PS_SELL_CASE(SPMV_POWER)
    }
#undef PS_SELL_CASE
}

// ---------------------------------------------------------------------------------------------
// BSR-3 SpMV (block_size 3: elasticity-type systems, AMGCL_Block<3>'s storage)
// ---------------------------------------------------------------------------------------------
// A workgroup step covers G consecutive block rows (3G rows), whose blocks are streamed in chunks of
// 256: the chunk's values (72 B per block) are loaded with coalesced 16-byte loads -- one chunk ahead,
// in registers -- parked raw in LDS, then thread t multiplies block t by the 3 gathered x entries of
// its block column (one 24-byte gather instead of nine 8-byte ones) and parks the 3 row contributions;
// after the second barrier thread t < 3G adds up row t's contributions in block-column order.  The
// products are the scalar loop's; only the association differs (three products are summed per block
// first), i.e. a few ulp of the row's absolute sum.  24 KiB of LDS: 6 workgroups per CU.
constexpr int kBsrChunk = 256; // blocks per chunk = threads per workgroup

template <int MODE, typename VT, bool PD>
__global__ __launch_bounds__(kBlock) void spmv_bsr3_kernel(int nb, int64_t nnzb, const int *__restrict__ browptr,
                                                            const int *__restrict__ bcol,
                                                            const VT *__restrict__ bval,
                                                            const double *__restrict__ x,
                                                            const double *__restrict__ b, double *__restrict__ y,
                                                            double *__restrict__ partials,
                                                            const int *__restrict__ done_flag, int G, int ngroups,
                                                            int chunk_groups, int np_total)
{
    // single-buffered: A(write raw) |B1| C(read raw, write part) |B2| D(read part); every thread passes
    // D before it can reach the next chunk's B1, so neither tile is overwritten while still being read
    // VT = float: the block values are stored in single precision (40 B per block instead of 76 B), products
    // and sums in double; chunks then start at a multiple of 4 blocks (16-byte aligned 16-B loads)
    constexpr bool F32 = sizeof(VT) == 4;
    constexpr int ALIGN = F32 ? 4 : 2;
    __shared__ VT raw[kBsrChunk * 9];
    __shared__ double part[kBsrChunk * 3];
    __shared__ double red[kBlock / 64];
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    // groups are dealt to the XCDs in chunks of `chunk_groups` consecutive groups (same idea as the CSR schedule)
    const int nloop = (((ngroups + chunk_groups - 1) / chunk_groups + 7) / 8) * chunk_groups;
    auto group_of = [&](int l) { return ((l / chunk_groups) * 8 + xcd) * chunk_groups + (l % chunk_groups); };
    double dacc = 0.0;
    v2d pre[F32 ? 1 : 5];
    v4f pre4[F32 ? 3 : 1];
    int pre_col = 0;
    // loads of the chunk [k0, kend) of the value stream (k0 a multiple of ALIGN => 16-byte aligned)
    auto load_chunk = [&](int k0, int kend) {
        const int nd = 9 * (kend - k0); // values in the chunk
        if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int d = 4 * (j * kBlock + tid);
                pre4[j] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (d < nd) {
                    const int64_t g = (int64_t)9 * k0 + d;
                    if (g + 3 < (int64_t)9 * nnzb) {
                        pre4[j] = *(const v4f *)(bval + g);
                    } else {
                        if (g < (int64_t)9 * nnzb) pre4[j].x = bval[g];
                        if (g + 1 < (int64_t)9 * nnzb) pre4[j].y = bval[g + 1];
                        if (g + 2 < (int64_t)9 * nnzb) pre4[j].z = bval[g + 2];
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int d = 2 * (j * kBlock + tid);
                pre[j] = (v2d){0.0, 0.0};
                if (d < nd) {
                    const int64_t g = (int64_t)9 * k0 + d;
                    if (g + 1 < (int64_t)9 * nnzb) pre[j] = *(const v2d *)(bval + g);
                    else if (g < (int64_t)9 * nnzb) pre[j].x = bval[g];
                }
            }
        }
        pre_col = (k0 + tid < kend) ? bcol[k0 + tid] : 0;
    };
    int l = slot;
    int lo = 0, hi = 0;
    bool have = l < nloop && group_of(l) < ngroups;
    if (have) {
        const int brow0 = group_of(l) * G;
        lo = browptr[brow0];
        hi = browptr[min(brow0 + G, nb)];
        load_chunk(lo & ~(ALIGN - 1), min((lo & ~(ALIGN - 1)) + kBsrChunk, hi));
    }
    while (have) {
        const int g = group_of(l);
        const int brow0 = g * G;
        // PD (groups of up to 10 block rows, i.e. long block rows): eight lanes per (block row, component) share the
        // sum of the row's contributions -- one row thread alone adds ~27 LDS values one after the other while 230
        // threads wait at the barrier (Q1 elasticity M = 100: 0.424 -> 0.373 ms per product, AMG-PCG 158 -> 139 ms;
        // non-temporal loads of the block stream on top: 0.403 ms, worse)
        const int rt = PD ? tid >> 3 : tid, sub = PD ? tid & 7 : 0;
        const int br = brow0 + rt / 3, comp = rt % 3;
        const bool row_thread = rt < 3 * G && br < nb;
        int bs = 0, be = 0;
        if (row_thread) {
            bs = browptr[br];
            be = browptr[br + 1];
        }
        // next group's extent (for the prefetch at the end of this group's last chunk)
        const int ln = l + slots;
        const bool have_next = ln < nloop && group_of(ln) < ngroups;
        int lo_n = 0, hi_n = 0;
        if (have_next) {
            const int brow0n = group_of(ln) * G;
            lo_n = browptr[brow0n];
            hi_n = browptr[min(brow0n + G, nb)];
        }
        double acc = 0.0;
        for (int k0 = lo & ~(ALIGN - 1); k0 < hi; k0 += kBsrChunk) {
            const int kend = min(k0 + kBsrChunk, hi);
            // A: prefetched registers -> raw LDS
            VT *R = raw;
            double *P = part;
            const int nd = 9 * (kend - k0);
            if constexpr (F32) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int d = 4 * (j * kBlock + tid);
                    if (d < nd) *(v4f *)(R + d) = pre4[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int d = 2 * (j * kBlock + tid);
                    if (d < nd) *(v2d *)(R + d) = pre[j];
                }
            }
            const int myk = k0 + tid, mycol = pre_col;
            __syncthreads();
            // B: the next chunk (of this group, or the first of the next group) goes out now
            if (k0 + kBsrChunk < hi) load_chunk(k0 + kBsrChunk, min(k0 + 2 * kBsrChunk, hi));
            else if (have_next) load_chunk(lo_n & ~(ALIGN - 1), min((lo_n & ~(ALIGN - 1)) + kBsrChunk, hi_n));
            // C: block products
            if (myk >= lo && myk < kend) {
                const double x0 = x[3 * mycol], x1 = x[3 * mycol + 1], x2 = x[3 * mycol + 2];
                const VT *v = R + 9 * tid;
                double s0 = (double)v[0] * x0, s1 = (double)v[3] * x0, s2 = (double)v[6] * x0;
                s0 += (double)v[1] * x1; s1 += (double)v[4] * x1; s2 += (double)v[7] * x1;
                s0 += (double)v[2] * x2; s1 += (double)v[5] * x2; s2 += (double)v[8] * x2;
                P[3 * tid] = s0;
                P[3 * tid + 1] = s1;
                P[3 * tid + 2] = s2;
            }
            __syncthreads();
            // D: rows of this group add up their slice of the chunk
            if (row_thread) {
                const int a = max(bs, k0), e = min(be, kend);
                if constexpr (PD) {
                    for (int k = a + sub; k < e; k += 8) acc += P[3 * (k - k0) + comp];
                } else {
                    for (int k = a; k < e; ++k) acc += P[3 * (k - k0) + comp];
                }
            }
        }
        // a group of empty block rows never entered the chunk loop, so nobody prefetched for the next group
        if (have_next && !((lo & ~(ALIGN - 1)) < hi))
            load_chunk(lo_n & ~(ALIGN - 1), min((lo_n & ~(ALIGN - 1)) + kBsrChunk, hi_n));
        if constexpr (PD) {
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        }
        if (row_thread && sub == 0) {
            const int r = 3 * br + comp;
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += x[r] * acc;
            }
            y[r] = acc;
        }
        l = ln;
        have = have_next;
        lo = lo_n;
        hi = hi_n;
    }
    if (MODE != SPMV_PLAIN) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            partials[blockIdx.x] = t;
            // consumers fold np_total (= the CSR SpMV grid) partials: clear the slots this smaller grid does not own
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
}

// The same product with the block stream staged by LDS-DMA (no staging registers: 60-odd VGPRs instead of 93, six
// resident workgroups per CU instead of five) -- the scheme of spmv_csr_dma on 72-byte blocks.  Single-buffered:
// DMA of the chunk's values and block columns |B1| block products -> part |B2| row sums.
// Round 4: every block operator of a block-3 hierarchy runs here (level operators, prolongations with ~4 blocks per
// block row, restrictions with ~120), so the shape is a launch parameter: G block rows per group (any number, 3 G <= 256)
// and LPR = 2^lpr_log2 lanes per (block row, component) for the row sums (3 G LPR <= 256; a group of long block rows
// spans several chunks and carries its sums across them), and the epilogues the cycle needs are fused behind the row
// sums: SPMV_ADD (prolongation) and SPMV_CHEB -- the Chebyshev step with BLOCK scaling, res = D_i^-1 (b - A x)_i needs
// the three residuals of a node, which the row threads exchange through LDS (one barrier per group) instead of a
// residual vector in HBM and a second launch (block_cheb_update_kernel).
constexpr int kBsrChebRows = 96; // SPMV_CHEB: at most 32 block rows per group (the exchange buffer is double-buffered)

// LPRLOG >= 0: 2^LPRLOG lanes per row sum known at compile time (8 for the ~27 blocks per block row of a 3-D elasticity
// operator: the instantiation PCG's product runs on); -1: taken from `lpr_log2`.
// PRE: a thread keeps its block column in a register and issues its three gathers BEFORE the barrier, next to the DMA,
// instead of behind it.  Measured at M = 100 (one box, interleaved runs, profiles/r04_amg.md section 1): the plain
// epilogues lose 2-4 % with it (their chain is not what bounds them), the fused Chebyshev step -- whose epilogue adds a
// barrier and six operand loads per row to the chain of a group -- gains 15 %; the next group's row pointers fetched one
// group ahead gained nothing in either (dropped).
// Round 6, VT = float: "amg.matrix_fp32" -- the cycle's copy of a block operator holds single-precision values (40 instead
// of 76 bytes per block; PCG's own product, the vectors and every sum stay double).  The same kernel: the chunk's values are
// staged as 4-byte elements (four per DMA lane, chunks start on a multiple of four blocks = 144 bytes), widened when read.
template <int MODE, int LPRLOG, bool PRE, typename VT = double>
__global__ __launch_bounds__(kBlock) void spmv_bsr3_dma(int nb, int64_t nnzb, const int *__restrict__ browptr,
                                                         const int *__restrict__ bcol,
                                                         const VT *__restrict__ bval,
                                                         const double *__restrict__ x, const double *__restrict__ b,
                                                         double *__restrict__ y, double *__restrict__ partials,
                                                         const int *__restrict__ done_flag, int G, int ngroups,
                                                         int chunk_groups, int np_total, int lpr_log2,
                                                         const double *__restrict__ dinv_blk, double *__restrict__ pvec,
                                                         double alpha, double beta, int reverse)
{
    constexpr bool kPreGather = PRE;
    constexpr int EPL = 16 / (int)sizeof(VT);     // elements per DMA lane (16 bytes)
    constexpr int kAlign = sizeof(VT) == 8 ? 1 : 3; // chunks start on a block whose values start on a 16-byte boundary
    __shared__ __attribute__((aligned(16))) VT raw[kBsrChunk * 9];
    __shared__ __attribute__((aligned(16))) int lcol[kPreGather ? 1 : kBsrChunk];
    __shared__ double part[kBsrChunk * 3];
    __shared__ double red[kBlock / 64];
    __shared__ double nres[MODE == SPMV_CHEB ? 2 * kBsrChebRows : 1];
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = (((ngroups + chunk_groups - 1) / chunk_groups + 7) / 8) * chunk_groups;
    const int64_t nval = (int64_t)9 * nnzb;
    const int lg = LPRLOG >= 0 ? LPRLOG : lpr_log2;
    const int lpr = 1 << lg;
    const int sub = tid & (lpr - 1); // `lpr` lanes per (block row, component)
    int rt = tid >> lg;
    int rt_brow = rt / 3, comp = rt - 3 * rt_brow;
    bool rt_ok = rt < 3 * G;
    // round 5, the fused block Chebyshev step with eight lanes per row sum and at most eight block rows per group (the
    // 27-blocks-per-row operators it is the dominant kernel of): the three row sums of a node sit in ONE wave -- wave w owns
    // the nodes 2 w and 2 w + 1 of the group, eight-lane teams 0..2 and 3..5, teams 6 and 7 idle -- so the node's residuals
    // meet by shuffles instead of an LDS round trip behind a third barrier per group.  Which lanes add a row's partial
    // products changes, the order in which they are added does not (lane `sub` still takes the blocks sub, sub + 8, ...).
    const bool wave_nodes = MODE == SPMV_CHEB && LPRLOG == 3 && G <= 8;
    if (wave_nodes) {
        const int team = lane >> 3, node_l = team / 3;
        comp = team - 3 * node_l;
        rt_brow = 2 * wave + node_l;
        rt_ok = team < 6 && rt_brow < G;
        rt = 3 * rt_brow + comp;
    }
    double dacc = 0.0;
    int par = 0;
    for (int l = slot; l < nloop; l += slots) {
        const int gf = ((l / chunk_groups) * 8 + xcd) * chunk_groups + (l % chunk_groups);
        if (gf >= ngroups) continue; // (uniform)
        const int g = reverse ? ngroups - 1 - gf : gf;
        const int brow0 = g * G;
        const int lo = browptr[brow0], hi = browptr[min(brow0 + G, nb)];
        const int br = brow0 + rt_brow;
        const bool row_thread = rt_ok && br < nb;
        const int r = 3 * br + comp;
        const bool mine = row_thread && sub == 0;
        int bs = 0, be = 0;
        if (row_thread) {
            bs = browptr[br];
            be = browptr[br + 1];
        }
        // operands of the fused epilogues, requested at the top of the group: behind the row sums they add their latency
        // to the single-buffered workgroup's serial chain (the fused Chebyshev step at M = 100: 642 -> 412 us)
        double e_b = 0.0, e_p = 0.0, e_x = 0.0, e_d0 = 0.0, e_d1 = 0.0, e_d2 = 0.0;
        if (MODE == SPMV_ADD || MODE == SPMV_CHEB) {
            if (mine) {
                e_x = MODE == SPMV_ADD ? y[r] : x[r];
                if (MODE == SPMV_CHEB) {
                    e_b = b[r];
                    if (beta != 0.0) e_p = pvec[r];
                    const double *D = dinv_blk + (size_t)9 * br + 3 * comp;
                    e_d0 = D[0];
                    e_d1 = D[1];
                    e_d2 = D[2];
                }
            }
        }
        double acc = 0.0;
        for (int k0 = lo & ~kAlign; k0 < hi; k0 += kBsrChunk) { // (an even block -- fp32: every fourth -- starts on a 16-byte boundary)
            const int kend = min(k0 + kBsrChunk, hi);
            const int nd = 9 * (kend - k0);
#pragma unroll
            for (int k = 0; k < (sizeof(VT) == 8 ? 5 : 3); ++k) { // 2 doubles (4 floats) per lane, 128 (256) per wave instruction
                const int e = (k * 4 + wave) * 64 * EPL;
                if (e < nd) {
                    const int64_t i = (int64_t)9 * k0 + e + lane * EPL;
                    if (i + EPL - 1 < nval) dma16(bval + i, raw + e, false); // (non-temporal: 0.366 ms against 0.338 ms at M = 100)
                }
            }
            if ((int64_t)9 * kend + EPL - 1 >= nval && tid < EPL - 1 && (nval & (EPL - 1))) { // the last values of the whole array (no full DMA lane), by hand
                const int64_t i = (nval & ~(int64_t)(EPL - 1)) + tid;
                if (i < nval && i >= (int64_t)9 * k0 && i - (int64_t)9 * k0 < kBsrChunk * 9) raw[i - (int64_t)9 * k0] = bval[i];
            }
            const int myk = k0 + tid;
            const bool has_block = myk >= lo && myk < kend;
            double x0 = 0.0, x1 = 0.0, x2 = 0.0;
            if constexpr (kPreGather) {
                if (has_block) {
                    const int c = bcol[myk];
                    x0 = x[3 * c];
                    x1 = x[3 * c + 1];
                    x2 = x[3 * c + 2];
                }
            } else {
                if (myk < kend) lcol[tid] = bcol[myk];
            }
            __syncthreads();
            if (has_block) {
                if constexpr (!kPreGather) {
                    const int c = lcol[tid];
                    x0 = x[3 * c];
                    x1 = x[3 * c + 1];
                    x2 = x[3 * c + 2];
                }
                const VT *v = raw + 9 * tid;
                double s0 = (double)v[0] * x0, s1 = (double)v[3] * x0, s2 = (double)v[6] * x0;
                s0 += (double)v[1] * x1; s1 += (double)v[4] * x1; s2 += (double)v[7] * x1;
                s0 += (double)v[2] * x2; s1 += (double)v[5] * x2; s2 += (double)v[8] * x2;
                part[3 * tid] = s0;
                part[3 * tid + 1] = s1;
                part[3 * tid + 2] = s2;
            }
            __syncthreads();
            if (row_thread) {
                const int a = max(bs, k0), e = min(be, kend);
                for (int k = a + sub; k < e; k += lpr) acc += part[3 * (k - k0) + comp];
            }
            // (the next chunk's DMA writes raw and lcol, which nobody reads after B2; part is rewritten only after
            // the next B1, which every thread reaches after its row sums)
        }
        if constexpr (LPRLOG >= 0) {
#pragma unroll
            for (int off = (1 << (LPRLOG >= 0 ? LPRLOG : 0)) >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        } else {
            for (int off = lpr >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off); // (lpr <= 64: inside the wave)
        }
        if constexpr (MODE == SPMV_CHEB) {
            // block-Jacobi-scaled Chebyshev step (amgcl::relaxation::chebyshev with a block value type): the node's
            // three residuals meet in LDS; same operation order as block_cheb_update_kernel
            if (wave_nodes) {
                const double rv = e_b - acc;
                const int base = (lane >> 3) / 3 * 24; // first lane of this node's three teams
                const double t0 = __shfl(rv, base), t1 = __shfl(rv, base + 8), t2 = __shfl(rv, base + 16);
                if (mine) {
                    double res = 0.0;
                    res += e_d0 * t0;
                    res += e_d1 * t1;
                    res += e_d2 * t2;
                    const double pn = (beta != 0.0) ? alpha * res + beta * e_p : alpha * res;
                    pvec[r] = pn;
                    y[r] = e_x + pn;
                }
            } else {
            double *ex = nres + par * kBsrChebRows;
            if (mine) ex[rt] = e_b - acc;
            __syncthreads();
            if (mine) {
                const double *t = ex + 3 * rt_brow;
                double res = 0.0;
                res += e_d0 * t[0];
                res += e_d1 * t[1];
                res += e_d2 * t[2];
                const double pn = (beta != 0.0) ? alpha * res + beta * e_p : alpha * res;
                pvec[r] = pn;
                y[r] = e_x + pn;
            }
            par ^= 1; // (the buffer written two groups later: every thread has passed the next group's barrier by then)
            }
        } else if (mine) {
            if (MODE == SPMV_RESIDUAL) {
                acc = b[r] - acc;
                dacc += acc * acc;
            } else if (MODE == SPMV_DOT) {
                dacc += x[r] * acc;
            } else if (MODE == SPMV_ADD) {
                acc = e_x + acc;
            }
            y[r] = acc;
        }
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            if ((int)blockIdx.x < np_total) partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
}

// block rows per workgroup step.  Up to ~113 blocks per block row: as many block rows as fill one 256-block chunk with 12 %
// head-room (any number, not only powers of two: prolongations with 4.4 blocks per block row take 51, not 32 -- 55 % of a
// chunk); longer block rows (restrictions: ~120 blocks): the 1..4 block rows whose blocks fill whole chunks best, the group
// then spans several chunks.
int bsr3_brows_per_group(double avg_blocks_per_brow)
{
    const double avg = std::max(1.0, avg_blocks_per_brow);
    const int single = (int)((double)(kBsrChunk - 2) / (avg * 1.12));
    if (single >= 2) return std::min(single, 85); // 3 G <= 256 row threads
    int best = 1;
    double best_fill = 0.0;
    for (int G = 1; G <= 4; ++G) {
        const double blocks = G * avg;
        const double chunks = std::ceil(blocks * 1.08 / (double)kBsrChunk);
        const double fill = blocks / (chunks * kBsrChunk);
        if (fill > best_fill + 1e-9) {
            best_fill = fill;
            best = G;
        }
    }
    return best;
}

// lanes per (block row, component) of the row sums: the largest power of two with 3 G lanes <= 256
static int bsr3_lanes_log2(int G)
{
    int lg = 0;
    while (lg < 6 && 3 * G * (2 << lg) <= kBlock) ++lg;
    return lg;
}

bool bsr3_serves(const Bsr3Dev &B, SpmvMode mode, const Launch &L, const SpmvExtra &ex)
{
    if (ex.rb_list) return false;
    if (mode == SPMV_PLAIN || mode == SPMV_DOT || mode == SPMV_RESIDUAL) return true;
    // the fused epilogues live in the LDS-DMA kernel
    if (L.spmv_kernel == 0) return false;
    if (mode == SPMV_ADD) return true;
    if (mode == SPMV_CHEB) return ex.dinv_blk != nullptr && 3 * B.brows_per_group <= kBsrChebRows;
    return false;
}

// ---------------------------------------------------------------------------------------------
// 3x3-block product from block-row kinds (Bsr3KindDev): no matrix stream
// ---------------------------------------------------------------------------------------------
// One lane per node (block row): its kind's (offset, block id) list and the distinct 3x3 blocks are read from LDS, x of the
// neighbour nodes by buffer loads (three at a time in flight), the three row sums in column order -- the scalar CSR loop's
// order.  A short block row is padded with (offset 0, the all-zero block): adding 0.0 x[own node] leaves a sum that started
// at +0.0 as it is.  The 3x3 epilogue of the fused Chebyshev step is lane-local (no LDS round trip, no barrier).

template <int MODE, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_bsr3_kind(int nb, Bsr3KindDev K, const double *__restrict__ x,
                                                          const double *__restrict__ b, double *__restrict__ y,
                                                          double *__restrict__ partials, const int *__restrict__ done_flag,
                                                          int np_total, const double *__restrict__ dinv_blk,
                                                          double *__restrict__ pvec, double alpha, double beta)
{
    __shared__ double red[kBlock / 64];
    // [nblk * 10] the blocks, padded to 80 bytes (16-byte reads) | [nk * kml] entries: block offset << 10 | block id
    extern __shared__ __attribute__((aligned(16))) double lbk[];
    if (done_flag && *done_flag) return;
    const int tid = threadIdx.x;
    const int nt = K.nk * K.kml, kml = K.kml;
    int *lent = reinterpret_cast<int *>(lbk + 10 * K.nblk);
    for (int t = tid; t < 10 * K.nblk; t += kBlock) {
        const int bid = t / 10, q = t - 10 * bid;
        lbk[t] = q < 9 ? K.blocks[9 * bid + q] : 0.0;
    }
    for (int t = tid; t < nt; t += kBlock) lent[t] = K.koff[t] * 1024 + (int)K.kblk[t]; // (the id is below 1024; offsets fit 21 bits)
    __syncthreads();
    const int nrb = (nb + kBlock - 1) / kBlock, rb_per_xcd = (nrb + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, (int)((unsigned)nb * 24u), 0x00020000);
    auto ld = [&](unsigned off) -> double {
        const slot_u2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)off, 0, 0);
        return __hiloint2double((int)v.y, (int)v.x);
    };
    double dacc = 0.0;
    for (int k = slot; k < rb_per_xcd; k += slots) {
        const int rb = xcd * rb_per_xcd + k;
        const int node = rb * kBlock + tid;
        if (rb >= nrb || node >= nb) continue; // (no barrier below)
        const int tb = (int)K.kind[node] * kml;
        const unsigned base = (unsigned)node * 24u;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int j0 = 0; j0 < kml; j0 += 3) {
            double xv[3][3];
            const double *bv[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = lent[tb + j0 + u];
                const unsigned o = base + (unsigned)((e >> 10) * 24); // (arithmetic shift: the signed block offset)
                xv[u][0] = ld(o);
                xv[u][1] = ld(o + 8u);
                xv[u][2] = ld(o + 16u);
                bv[u] = lbk + 10 * (e & 1023);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const v2d *q = reinterpret_cast<const v2d *>(bv[u]);
                const v2d q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                const double v8 = bv[u][8];
                a0 += q0.x * xv[u][0];
                a0 += q0.y * xv[u][1];
                a0 += q1.x * xv[u][2];
                a1 += q1.y * xv[u][0];
                a1 += q2.x * xv[u][1];
                a1 += q2.y * xv[u][2];
                a2 += q3.x * xv[u][0];
                a2 += q3.y * xv[u][1];
                a2 += v8 * xv[u][2];
            }
        }
        const size_t r = (size_t)3 * node;
        if (MODE == SPMV_RESIDUAL) {
            a0 = b[r] - a0;
            a1 = b[r + 1] - a1;
            a2 = b[r + 2] - a2;
            dacc += a0 * a0;
            dacc += a1 * a1;
            dacc += a2 * a2;
        } else if (MODE == SPMV_DOT) {
            dacc += ld(base) * a0;
            dacc += ld(base + 8u) * a1;
            dacc += ld(base + 16u) * a2;
        } else if (MODE == SPMV_ADD) {
            a0 = y[r] + a0;
            a1 = y[r + 1] + a1;
            a2 = y[r + 2] + a2;
        } else if (MODE == SPMV_CHEB) {
            const double t0 = b[r] - a0, t1 = b[r + 1] - a1, t2 = b[r + 2] - a2;
            const double *D = dinv_blk + (size_t)9 * node;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            s0 += D[0] * t0;
            s0 += D[1] * t1;
            s0 += D[2] * t2;
            s1 += D[3] * t0;
            s1 += D[4] * t1;
            s1 += D[5] * t2;
            s2 += D[6] * t0;
            s2 += D[7] * t1;
            s2 += D[8] * t2;
            double p0 = alpha * s0, p1 = alpha * s1, p2 = alpha * s2;
            if (beta != 0.0) {
                p0 = p0 + beta * pvec[r];
                p1 = p1 + beta * pvec[r + 1];
                p2 = p2 + beta * pvec[r + 2];
            }
            store_stream<NT>(pvec + r, p0);
            store_stream<NT>(pvec + r + 1, p1);
            store_stream<NT>(pvec + r + 2, p2);
            a0 = ld(base) + p0;
            a1 = ld(base + 8u) + p1;
            a2 = ld(base + 16u) + p2;
        }
        store_stream<NT>(y + r, a0);
        store_stream<NT>(y + r + 1, a1);
        store_stream<NT>(y + r + 2, a2);
    }
    if (MODE == SPMV_DOT || MODE == SPMV_RESIDUAL) {
        const double t = block_sum(dacc, red);
        if (tid == 0 && partials) {
            if ((int)blockIdx.x < np_total) partials[blockIdx.x] = t;
            for (int k = blockIdx.x + gridDim.x; k < np_total; k += gridDim.x) partials[k] = 0.0;
        }
    }
}

static void launch_spmv_bsr3_kind(const Launch &L, const Bsr3Dev &B, SpmvMode mode, const double *x, const double *b, double *y,
                                  double *partials, const int *done_flag, const SpmvExtra &ex)
{
    const Bsr3KindDev &K = *B.kinds;
    const int nrb = (B.nb + kBlock - 1) / kBlock;
    const size_t lds = (size_t)K.nblk * 80 + (size_t)K.nk * K.kml * 4 + 16;
    // (the tables take up to 40 KiB of LDS: three or four workgroups per CU)
    const int per_cu = std::max(1, std::min(8, (int)((150 * 1024) / (lds + 1024))));
    const int grid = std::max(8, std::min(std::min(L.spmv_grid, (L.num_cus * per_cu + 7) & ~7), (nrb + 7) & ~7));
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && 50ll * B.nb > L.spmv_nt_bytes);
    PS_NOTE_KERNEL("spmv_bsr3_kind<%d, %s>", (int)mode, nt ? "true" : "false");
#define PS_BK_CASE(M)                                                                                                    \
    case M:                                                                                                              \
        if (nt)                                                                                                          \
            PS_TIMED_LAUNCH((spmv_bsr3_kind<M, true>), dim3(grid), dim3(kBlock), lds, L.stream, B.nb, K, x, b, y, partials, \
                               done_flag, L.spmv_grid, ex.dinv_blk, ex.p, ex.alpha, ex.beta);                            \
        else                                                                                                             \
            PS_TIMED_LAUNCH((spmv_bsr3_kind<M, false>), dim3(grid), dim3(kBlock), lds, L.stream, B.nb, K, x, b, y, partials, \
                               done_flag, L.spmv_grid, ex.dinv_blk, ex.p, ex.alpha, ex.beta);                            \
        break;
    switch (mode) {
        PS_BK_CASE(SPMV_PLAIN)
        PS_BK_CASE(SPMV_DOT)
        PS_BK_CASE(SPMV_RESIDUAL)
        PS_BK_CASE(SPMV_ADD)
        PS_BK_CASE(SPMV_CHEB)
    default: break;
    }
#undef PS_BK_CASE
}

static void launch_spmv_bsr3(const Launch &L, const Bsr3Dev &B, SpmvMode mode, const double *x, const double *b,
                             double *y, double *partials, const int *done_flag, const SpmvExtra &ex)
{
    if (B.kinds && L.lab.bsr3_kinds && L.spmv_kernel != 0 && (int64_t)B.nb * 24 < (1ll << 32) &&
        (mode != SPMV_CHEB || ex.dinv_blk)) {
        launch_spmv_bsr3_kind(L, B, mode, x, b, y, partials, done_flag, ex);
        return;
    }
    const int G = B.brows_per_group;
    const int ngroups = (B.nb + G - 1) / G;
    // chunks dealt to the XCDs; fewer than 32 chunks per XCD would leave XCDs idle: shrink towards round-robin
    const int chunk_groups = std::max(1, std::min(L.spmv_chunk_rows / (3 * G), ngroups / 256));
    // 24 KiB of LDS per workgroup would admit 6 workgroups per CU, but 5 is the measured optimum (M = 100 elasticity,
    // block-3 AMG-PCG: 162 ms at 5 per CU, 215 ms at 6)
    // (93 VGPRs: five waves per SIMD is what is resident anyway; forced down to 80 for six, the kernel spills and
    // runs at 0.50 ms instead of 0.42 ms)
    const int grid = std::max(8, std::min(L.spmv_grid, (L.num_cus * 5 + 7) & ~7));
    dim3 g(grid), blk(kBlock);
    const bool pd = G <= 10; // 3 G row sums x 8 lanes fit the workgroup
    // double values: the LDS-DMA staged kernel at six workgroups per CU (Q1 elasticity M = 100: 0.338 ms
    // against 0.376 ms, M = 64: 0.092 against 0.107 ms); "spmv_kernel" 0 keeps the register-staged one
    if (L.spmv_kernel != 0) {
        // a small operator (coarse levels, their transfers): no more workgroups than groups
        const int gd = std::max(8, std::min(std::min(L.spmv_grid, (L.num_cus * 6 + 7) & ~7), (ngroups + 7) & ~7));
        const int lg = bsr3_lanes_log2(G);
        {
            const bool pre_n = L.bsr3_variant >= 0 ? (L.bsr3_variant & 1) != 0 : mode == SPMV_CHEB;
            if (B.val32) PS_NOTE_KERNEL("spmv_bsr3_dma<%d, %d, %s, float>", (int)mode, lg == 3 ? 3 : -1, pre_n ? "true" : "false");
            else PS_NOTE_KERNEL("spmv_bsr3_dma<%d, %d, %s>", (int)mode, lg == 3 ? 3 : -1, pre_n ? "true" : "false");
        }
#define PS_BSRD_LAUNCH(M, LG, PRE)                                                                                \
    do {                                                                                                          \
        if (B.val32)                                                                                              \
            PS_TIMED_LAUNCH((spmv_bsr3_dma<M, LG, PRE, float>), dim3(gd), blk, 0, L.stream, B.nb, B.nnzb, B.rowptr, B.col, B.val32, x, \
                               b, y, partials, done_flag, G, ngroups, chunk_groups, L.spmv_grid, lg, ex.dinv_blk, ex.p, ex.alpha, \
                               ex.beta, ex.reverse);                                                              \
        else                                                                                                      \
            PS_TIMED_LAUNCH((spmv_bsr3_dma<M, LG, PRE, double>), dim3(gd), blk, 0, L.stream, B.nb, B.nnzb, B.rowptr, B.col, B.val, x, \
                               b, y, partials, done_flag, G, ngroups, chunk_groups, L.spmv_grid, lg, ex.dinv_blk, ex.p, ex.alpha, \
                               ex.beta, ex.reverse);                                                              \
    } while (0)
#define PS_BSRD_CASE(M)                                                                                           \
    case M: {                                                                                                     \
        const bool pre = L.bsr3_variant >= 0 ? (L.bsr3_variant & 1) != 0 : M == SPMV_CHEB;                        \
        const bool lg3 = lg == 3 && !(L.bsr3_variant >= 0 && (L.bsr3_variant & 4)); /* (4: the lane count at run time, A/B) */ \
        if (lg3 && pre) PS_BSRD_LAUNCH(M, 3, true);                                                               \
        else if (lg3) PS_BSRD_LAUNCH(M, 3, false);                                                                \
        else if (pre) PS_BSRD_LAUNCH(M, -1, true);                                                                \
        else PS_BSRD_LAUNCH(M, -1, false);                                                                        \
    } break;
        switch (mode) {
            PS_BSRD_CASE(SPMV_PLAIN)
            PS_BSRD_CASE(SPMV_DOT)
            PS_BSRD_CASE(SPMV_RESIDUAL)
            PS_BSRD_CASE(SPMV_ADD)
            PS_BSRD_CASE(SPMV_CHEB)
        default: break;
        }
#undef PS_BSRD_CASE
#undef PS_BSRD_LAUNCH
        return;
    }
    PS_NOTE_KERNEL("spmv_bsr3_kernel<%d, %s, %s>", (int)mode, B.val32 ? "float" : "double", pd ? "true" : "false");
#define PS_BSR_LAUNCH(M, VT, V, PDF)                                                                              \
    PS_TIMED_LAUNCH((spmv_bsr3_kernel<M, VT, PDF>), g, blk, 0, L.stream, B.nb, B.nnzb, B.rowptr, B.col, V, x, b, y, \
                       partials, done_flag, G, ngroups, chunk_groups, L.spmv_grid)
#define PS_BSR_CASE(M)                                                                                            \
    case M:                                                                                                       \
        if (B.val32 && pd) PS_BSR_LAUNCH(M, float, B.val32, true);                                                \
        else if (B.val32) PS_BSR_LAUNCH(M, float, B.val32, false);                                                \
        else if (pd) PS_BSR_LAUNCH(M, double, B.val, true);                                                       \
        else PS_BSR_LAUNCH(M, double, B.val, false);                                                              \
        break;
    switch (mode) {
        PS_BSR_CASE(SPMV_PLAIN)
        PS_BSR_CASE(SPMV_DOT)
        PS_BSR_CASE(SPMV_RESIDUAL)
    default: break;
    }
#undef PS_BSR_CASE
#undef PS_BSR_LAUNCH
}

Launch fit_launch(const Launch &max_cfg, int n, int rows_per_block, double avg_nnz_per_row)
{
    Launch L = max_cfg;
    auto round8 = [](int64_t v) { return (int)((v + 7) & ~(int64_t)7); };
    const int64_t vec_blocks = ((int64_t)n + 1023) / 1024; // >= 4 elements per thread
    L.grid = std::max(8, std::min(max_cfg.grid, round8(vec_blocks)));
    const int64_t nrb = ((int64_t)n + rows_per_block - 1) / rows_per_block;
    int cap = max_cfg.spmv_grid;
    if (avg_nnz_per_row > 0 && rows_per_block < kBlock && max_cfg.spmv_kernel != 0) {
        // several threads per row: the LDS-DMA kernel with a tile sized to the row-blocks; more of those fit a CU
        const int wg = dma_wg_per_cu(dma_tile(max_cfg.lab, rows_per_block, avg_nnz_per_row), 8);
        cap = std::max(cap, std::min(kMaxPartials, round8((int64_t)wg * max_cfg.num_cus)));
    }
    L.spmv_grid = std::max(8, std::min(cap, round8((nrb + 1) / 2)));
    return L;
}

// fit_launch for the SETUP kernels of an operator (row-set patterns, prolongation values, Galerkin products, block copies):
// they give a row to a group of up to 64 lanes, so the persistent grid is sized by rows x lanes, not by n / 1024 as for the
// vector kernels -- level 2 of the 216^3 hierarchy (25 613 rows of 500 entries) ran them on 32 workgroups, ~1 ms each (round 4)
Launch fit_setup_launch(const Launch &max_cfg, int n, int64_t nnz, int rows_per_block)
{
    Launch L = fit_launch(max_cfg, n, rows_per_block);
    const double avg = n > 0 ? (double)nnz / (double)n : 1.0;
    const int64_t want = ((int64_t)n * (int64_t)std::min(64.0, std::max(1.0, avg)) + kBlock - 1) / kBlock;
    L.grid = std::max(L.grid, (int)std::min<int64_t>(max_cfg.grid, (want + 7) & ~(int64_t)7));
    return L;
}

// rows per row-block for a matrix with `avg` nonzeros per row.  Narrow rows (one thread per row): 256 where the average
// row-block fits spmv_csr_pipe's tile with 3 % head-room.  Wide rows (several threads per row, spmv_csr_dma): the
// largest power of two <= 128 whose average row-block holds at most kRowBlockFill (2304) entries.  Round 4: a row-block
// step of the LDS-DMA kernel is latency (stream in -> barrier -> dependent LDS read / gather / add chains -> barrier),
// so its rate is the bytes it keeps in flight per CU: level 1 of the 256^3 hierarchy (31.4 entries per row) with 32-row
// blocks = 8 workgroups x 12 KiB per CU ran its Chebyshev step in 257 us, with 64-row blocks = 6 x 24 KiB in 188 us (half
// of the row-blocks spill a few entries into a second pass of the 2048-entry tile; a tile of 2560 / 3072 entries that
// holds them all leaves 5 / 4 workgroups per CU: 236 / 234 us; 128-row blocks, always two passes: 192-216 us) --
// profiles/r04_level1.md
int spmv_rows_per_block(double avg_nnz_per_row)
{
    if (256 * avg_nnz_per_row * 1.03 <= (double)(kTile - 4)) return 256;
    int R = 128;
    while (R > 8 && R * avg_nnz_per_row * 1.03 > (double)kRowBlockFill) R >>= 1;
    return R;
}

// ---------------------------------------------------------------------------------------------
// 16-bit column copy (CsrDev::col16)
// ---------------------------------------------------------------------------------------------
// one workgroup per row-block: the distinct 8192-column windows its entries fall into (at most eight, else the operator
// keeps its 32-bit columns), sorted, and every entry as (window, offset)
__global__ __launch_bounds__(kBlock) void col16_build_kernel(int n, int R, const int *__restrict__ rowptr,
                                                             const int *__restrict__ col, unsigned short *__restrict__ c16,
                                                             int *__restrict__ rb_base, int *fail)
{
    __shared__ int win[8];
    const int nrb = (n + R - 1) / R;
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        if (threadIdx.x < 8) win[threadIdx.x] = -1;
        __syncthreads();
        const int lo = rowptr[rb * R], hi = rowptr[min(rb * R + R, n)];
        bool bad = false;
        for (int k = lo + threadIdx.x; k < hi; k += kBlock) {
            const int w = col[k] >> 13;
            bool placed = false;
            for (int s = 0; s < 8 && !placed; ++s) {
                int cur = __hip_atomic_load(&win[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == -1) cur = atomicCAS(&win[s], -1, w) == -1 ? w : win[s];
                placed = cur == w;
            }
            bad = bad || !placed;
        }
        if (bad) atomicExch(fail, 1);
        __syncthreads();
        if (threadIdx.x == 0) { // ascending, empty slots last: the layout does not depend on who came first
            for (int a = 1; a < 8; ++a) {
                const int v = win[a];
                int q = a;
                while (q > 0 && (win[q - 1] == -1 || (v != -1 && win[q - 1] > v))) {
                    win[q] = win[q - 1];
                    --q;
                }
                win[q] = v;
            }
            for (int a = 0; a < 8; ++a) rb_base[8 * rb + a] = win[a] < 0 ? 0 : win[a] << 13;
        }
        __syncthreads();
        for (int k = lo + threadIdx.x; k < hi; k += kBlock) {
            const int c = col[k], w = c >> 13;
            int sel = 0;
            for (int s = 0; s < 8; ++s)
                if (win[s] == w) sel = s;
            c16[k] = (unsigned short)((sel << 13) | (c & 8191));
        }
        __syncthreads();
    }
}

bool Col16::build(const Launch &L, const CsrDev &A)
{
    valid = false;
    if (A.n <= 0 || A.nnz <= 0) return false;
    const int R = A.rows_per_block, nrb = (A.n + R - 1) / R;
    col.ensure((size_t)A.nnz + 16);
    base.ensure((size_t)nrb * 8 + 8);
    flag.ensure(4);
    PS_HIP_CHECK(hipMemsetAsync(flag.ptr, 0, 4 * sizeof(int), L.stream));
    PS_HIP_CHECK(hipMemsetAsync(col.ptr + (A.nnz & ~(int64_t)7), 0, (size_t)(A.nnz + 16 - (A.nnz & ~(int64_t)7)) * sizeof(unsigned short),
                                L.stream));
    const int grid = std::max(1, std::min(nrb, 8 * L.num_cus));
    hipLaunchKernelGGL(col16_build_kernel, dim3(grid), dim3(kBlock), 0, L.stream, A.n, R, A.rowptr, A.col, col.ptr, base.ptr,
                       flag.ptr);
    PS_HIP_CHECK(hipGetLastError());
    int bad = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&bad, flag.ptr, sizeof(int), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    valid = bad == 0;
    return valid;
}

template <int R>
static void launch_spmv_r(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b,
                          double *y, double *partials, const int *done_flag, const SpmvExtra &ex)
{
    // (variable-height row-blocks where the operator has them for this height and no explicit row-block list / 16-bit columns;
    // not for the launches that reduce -- p.q, |r|^2, the power iteration: which rows a workgroup sums decides the order of
    // its partial sum, and those keep the fixed partition so that the scalars do not depend on the packing)
    // (round-4 advice: only spmv_csr_dma reads the row-block list -- a launch that falls through to the register-staged
    // kernel, "spmv_kernel" 0 or "amg.stream_nt" 0 on a level, must keep the fixed partition: with the packed count as nrb
    // its row-blocks beyond ceil(n / R) would read row pointers past the end)
    const int64_t bytes = A.nnz * (int64_t)(A.val32 ? 8 : 12) + 20ll * A.n;
    const bool nt = L.spmv_nt == 1 || (L.spmv_nt < 0 && bytes > L.spmv_nt_bytes);
    const bool by_operator = L.spmv_kernel < 0 || L.spmv_kernel >= 2;
    const bool use_dma = L.spmv_kernel == 1 || (by_operator && (nt || R < 256));
    const bool vrb = use_dma && A.rb_start && A.rb_R == R && A.rb_count > 0 && !ex.rb_list && !A.col16 && !A.val32 && R < 256 &&
                     !partials && !ex.partials2;
    const int nrb = vrb ? A.rb_count : (A.n + R - 1) / R;
    const int rb_per_xcd = (nrb + 7) / 8;
    // chunks dealt to XCDs pay off on big operators only (x stays in one L2); an operator with fewer than
    // 32 chunks per XCD (coarse AMG levels, transfer operators) would leave XCDs idle: round-robin there
    const int xcd_map = (L.spmv_xcd_map == 2 && (int64_t)nrb < 256ll * ex.chunk) ? 0 : L.spmv_xcd_map;
    dim3 grid(L.spmv_grid), block(kBlock);
    // Which kernel, which cache policy (profiles/r02_spmv_lab.md):
    //  * non-temporal stream + non-temporal y stores when the operator cannot live in the 256 MiB Infinity Cache
    //    (operator > spmv_nt_bytes = 384 MiB): at 256^3 the stores of y no longer cost four times their own time in the
    //    middle of the read stream (0.341 -> 0.302 ms).  Until round 4 the rule also asked for vectors too large for the
    //    cache (8 n >= 96 MiB), on a round-2 measurement (216^3 AMG-PCG 39 -> 43 ms) that no longer holds: with the
    //    kernels as they are now the non-temporal stream wins from ~400 MiB of operator on, whatever the vectors' size
    //    (Jacobi-PCG 176^3 / 192^3 / 216^3: -3 / -3 / -6 %, AMG-PCG -3 / -3 / -2.5 %; 128^3 ... 160^3 even, 112^3 +6 %:
    //    profiles/r04_nt_crossover.txt) -- what the stream would leave in the cache is its own tail, which the next sweep,
    //    starting at the other end, evicts before it gets there;
    //  * otherwise plain accesses (operators that fit the cache, coarse AMG levels);
    //  * the LDS-DMA kernel for the non-temporal case, round 1's register-staged pipeline for the rest (it
    //    overlaps more inside a workgroup and is the faster one out of the cache: 128^3 0.038 vs 0.041 ms).
    // the results are stored non-temporally where the vectors cannot stay in the cache anyway (8 n >= 64 MiB, the rule of the
    // fused vector kernels); the 16 MB vectors of a 765 MB level operator stay (level-1 Chebyshev step of the 256^3 hierarchy:
    // 185 us with non-temporal stores, the next step reading p and x back from HBM)
    const bool st_nt = L.spmv_nt == 1 || (nt && 8ll * A.n >= (64ll << 20)) || (L.lab.alternate & 2);
    // wide rows (R < 256: several threads per row) are latency-bound per row-block, not cache-bound: the DMA kernel
    // wins there with or without nt (level 1 of the 256^3 hierarchy, 31 nnz/row: 0.197 vs 0.223 ms; Q1 elasticity
    // as CSR, 81 nnz/row: 0.170 vs 0.185 ms)
    // ("spmv_kernel" 2 / 3 ask for a SELL copy / a pattern dictionary: an operator that has neither is served as with -1)
    if (use_dma) {
        // the tile follows the operator's row-blocks; a smaller tile admits more workgroups per CU.  The grid may only
        // grow where nobody reads per-workgroup partial sums afterwards (their count is the Launch's spmv_grid)
        // four gathers of a thread in flight pay where the gathered vector is about as long as the rows (level operators:
        // 274 -> 243 us on level 1 of the 256^3 hierarchy) and cost where it is much longer (its restriction, gathering
        // from the 134 MB fine vector: 197 -> 243 us) -- A/B in profiles/r03_amg.md
        SpmvExtra ex2 = ex;
        ex2.gather4 = (int64_t)A.n_ext <= 2ll * A.n ? 1 : 0;
        const int vbytes = A.val32 ? 4 : 8;
        // 16-bit columns where the operator has them (built for THIS row-block height; "spmv_kernel" 1 = the plain stream)
        const bool c16 = A.col16 && A.col16_R == R && !A.val32 && L.spmv_kernel != 1;
        int tile = dma_tile(L.lab, R, A.n > 0 ? (double)A.nnz / (double)A.n : 1.0);
        if (vrb) tile = A.rb_tile; // (what the blocks were packed for)
        if (c16) tile = std::min(L.lab.dma_tile_max, (tile + 511) & ~511); // (whole 512-entry column instructions)
        const size_t lds = (size_t)tile * ((c16 ? 2 : 4) + vbytes);
        dim3 dgrid = grid;
        // (not for restriction-like operators, whose gathers range over a vector much longer than their rows: eight
        // workgroups per CU gathering from the 134 MB fine vector cost R_0 of the 256^3 hierarchy 40 us against six)
        if (!partials && !ex.partials2 && !ex.rb_list && (int64_t)A.n_ext <= 2ll * A.n) {
            const int fit = (dma_wg_per_cu(tile, vbytes - (c16 ? 2 : 0)) * L.num_cus + 7) & ~7;
            const int want = std::max(8, std::min(fit, ((nrb + 1) / 2 + 7) & ~7));
            if (want > (int)grid.x) dgrid = dim3(std::min(want, kMaxPartials));
        } else if (!partials && !ex.partials2 && !ex.rb_list && nrb <= 6 * L.num_cus && ((nrb + 7) & ~7) > (int)grid.x) {
            // a restriction onto a small level (R_2 of the 256^3 hierarchy: 451 rows of 562 entries = 57 row-blocks of
            // three tile passes each, on a grid fitted to the 451-row level: 32 workgroups, two row-blocks one after the
            // other -- 65 us for 3 MB): fewer row-blocks than the device holds workgroups -> one each
            dgrid = dim3(std::min((nrb + 7) & ~7, kMaxPartials));
        }
        if (A.val32) PS_NOTE_KERNEL("spmv_csr_dma<%d, %d, float, %s, false, %s>", R, (int)mode, nt ? "true" : "false", nt ? "true" : "false");
        else if (c16) PS_NOTE_KERNEL("spmv_csr_dma<%d, %d, double, %s, true, %s>", R, (int)mode, nt ? "true" : "false", nt ? "true" : "false");
        else PS_NOTE_KERNEL("spmv_csr_dma<%d, %d, double, %s, false, %s>", R, (int)mode, nt ? "true" : "false", (nt && st_nt) ? "true" : "false");
#define PS_DMA_LAUNCH(M, VT, VP, NTF)                                                                               \
    PS_TIMED_LAUNCH((spmv_csr_dma<R, M, VT, NTF>), dgrid, block, lds, L.stream, A.n, A.nnz, A.rowptr, A.col, VP, x, b, y, \
                       partials, done_flag, nrb, rb_per_xcd, xcd_map, ex2, tile, (const int *)nullptr,                  \
                       vrb ? A.rb_start : (const int *)nullptr)
#define PS_DMA_LAUNCH_LD(M)                                                                                         \
    PS_TIMED_LAUNCH((spmv_csr_dma<R, M, double, true, false, false>), dgrid, block, lds, L.stream, A.n, A.nnz, A.rowptr, \
                       A.col, A.val, x, b, y, partials, done_flag, nrb, rb_per_xcd, xcd_map, ex2, tile,                 \
                       (const int *)nullptr, vrb ? A.rb_start : (const int *)nullptr)
#define PS_DMA16_LAUNCH(M, NTF)                                                                                     \
    PS_TIMED_LAUNCH((spmv_csr_dma<R, M, double, NTF, true>), dgrid, block, lds, L.stream, A.n, A.nnz, A.rowptr,         \
                       reinterpret_cast<const int *>(A.col16), A.val, x, b, y, partials, done_flag, nrb, rb_per_xcd,       \
                       xcd_map, ex2, tile, A.rb_base, (const int *)nullptr)
#define PS_DMA_CASE(M)                                                                                              \
    case M:                                                                                                         \
        if (A.val32) {                                                                                              \
            if (nt) PS_DMA_LAUNCH(M, float, A.val32, true);                                                         \
            else PS_DMA_LAUNCH(M, float, A.val32, false);                                                           \
        } else if (c16) {                                                                                           \
            if (nt) PS_DMA16_LAUNCH(M, true);                                                                       \
            else PS_DMA16_LAUNCH(M, false);                                                                         \
        } else {                                                                                                    \
            if (nt && !st_nt) PS_DMA_LAUNCH_LD(M);                                                                  \
            else if (nt) PS_DMA_LAUNCH(M, double, A.val, true);                                                     \
            else PS_DMA_LAUNCH(M, double, A.val, false);                                                            \
        }                                                                                                           \
        break;
        switch (mode) {
            PS_DMA_CASE(SPMV_PLAIN)
            PS_DMA_CASE(SPMV_DOT)
            PS_DMA_CASE(SPMV_RESIDUAL)
            PS_DMA_CASE(SPMV_ADD)
            PS_DMA_CASE(SPMV_CHEB)
            PS_DMA_CASE(SPMV_POWER)
        }
#undef PS_DMA_CASE
#undef PS_DMA16_LAUNCH
#undef PS_DMA_LAUNCH_LD
#undef PS_DMA_LAUNCH
        return;
    }
    // 2 x 14.5 KiB of LDS: five workgroups per CU, whatever the Launch's grid says
    const dim3 pgrid(std::max(8, std::min(L.spmv_grid, (L.num_cus * 5 + 7) & ~7)));
    PS_NOTE_KERNEL("spmv_csr_pipe<%d, %d, %s>", R, (int)mode, A.val32 ? "float" : "double");
#define PS_SPMV_CASE(M)                                                                                          \
    case M:                                                                                                      \
        if (A.val32)                                                                                             \
            PS_TIMED_LAUNCH((spmv_csr_pipe<R, M, float>), pgrid, block, 0, L.stream, A.n, A.nnz, A.rowptr, A.col, \
                               A.val32, x, b, y, partials, done_flag, nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid); \
        else                                                                                                     \
            PS_TIMED_LAUNCH((spmv_csr_pipe<R, M, double>), pgrid, block, 0, L.stream, A.n, A.nnz, A.rowptr, A.col, \
                               A.val, x, b, y, partials, done_flag, nrb, rb_per_xcd, xcd_map, ex, L.spmv_grid);  \
        break;
    switch (mode) {
        PS_SPMV_CASE(SPMV_PLAIN)
        PS_SPMV_CASE(SPMV_DOT)
        PS_SPMV_CASE(SPMV_RESIDUAL)
        PS_SPMV_CASE(SPMV_ADD)
        PS_SPMV_CASE(SPMV_CHEB)
        PS_SPMV_CASE(SPMV_POWER)
    }
#undef PS_SPMV_CASE
}

void launch_spmv(const Launch &L, const CsrDev &A, SpmvMode mode, const double *x, const double *b, double *y,
                 double *partials, const int *done_flag, const SpmvExtra *extra)
{
    SpmvExtra ex = extra ? *extra : SpmvExtra();
    if (A.bsr3 && bsr3_serves(*A.bsr3, mode, L, ex)) {
        launch_spmv_bsr3(L, *A.bsr3, mode, x, b, y, partials, done_flag, ex);
        PS_HIP_CHECK(hipGetLastError());
        return;
    }
    if (A.pat && !A.val32 && L.spmv_kernel != 0 && L.spmv_kernel != 1 && L.spmv_kernel != 2) {
        launch_spmv_pat(L, A, mode, x, b, y, partials, done_flag, ex);
        PS_HIP_CHECK(hipGetLastError());
        return;
    }
    if (A.sell && !ex.rb_list && !A.val32) {
        launch_spmv_sell(L, A, mode, x, b, y, partials, done_flag, ex);
        PS_HIP_CHECK(hipGetLastError());
        return;
    }
    ex.chunk = std::max(1, L.spmv_chunk_rows / A.rows_per_block);
    switch (A.rows_per_block) {
    case 256: launch_spmv_r<256>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    case 128: launch_spmv_r<128>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    case 64: launch_spmv_r<64>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    case 32: launch_spmv_r<32>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    case 16: launch_spmv_r<16>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    default: launch_spmv_r<8>(L, A, mode, x, b, y, partials, done_flag, ex); break;
    }
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void cheb_first_kernel(int n, double alpha, const double *__restrict__ dinv,
                                                             const double *__restrict__ b, double *__restrict__ p,
                                                             double *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double pn = alpha * (dinv[i] * b[i]); // x = 0: residual = b
        p[i] = pn;
        y[i] = pn;
    }
}

void launch_cheb_first(const Launch &L, int n, double alpha, const double *dinv, const double *b, double *p,
                       double *y)
{
    hipLaunchKernelGGL(cheb_first_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, alpha, dinv, b, p, y);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void scale_by_norm_kernel(int n, const double *__restrict__ partials, int np,
                                                                const double *__restrict__ s, double *__restrict__ b0)
{
    __shared__ double red[kBlock / 64];
    const double norm2 = fold_partials(partials, np, red);
    const double f = 1.0 / sqrt(norm2);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) b0[i] = f * s[i];
}

void launch_scale_by_norm(const Launch &L, int n, const double *partials, int np, const double *s, double *b0)
{
    hipLaunchKernelGGL(scale_by_norm_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, partials, np, s, b0);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void scale_expand_kernel(int n, int bs, double a, const double *__restrict__ x,
                                                               double *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) y[i] = a * x[i / bs];
}

__global__ __launch_bounds__(kBlock) void to_f32_kernel(int64_t n, const double *__restrict__ x, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] = (float)x[i];
}

void launch_to_f32(const Launch &L, int64_t n, const double *x, float *y)
{
    hipLaunchKernelGGL(to_f32_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, x, y);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_scale_expand(const Launch &L, int n, int bs, double a, const double *x, double *y)
{
    hipLaunchKernelGGL(scale_expand_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, bs, a, x, y);
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// BLAS-1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void dot_kernel(int n, const double *__restrict__ a,
                                                      const double *__restrict__ b, double *__restrict__ partials)
{
    __shared__ double red[kBlock / 64];
    const int n2 = n >> 1;
    double s = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n2; i += gridDim.x * kBlock) {
        const v2d va = ((const v2d *)a)[i], vb = ((const v2d *)b)[i];
        s += va.x * vb.x;
        s += va.y * vb.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s += a[n - 1] * b[n - 1];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

void launch_dot(const Launch &L, int n, const double *a, const double *b, double *partials)
{
    hipLaunchKernelGGL(dot_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, a, b, partials);
    PS_HIP_CHECK(hipGetLastError());
}

// out[v] = sum(partials[v*stride .. v*stride+np))  for v < nvec
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const double *__restrict__ partials, int np, int stride,
                                                               double *__restrict__ out, int nvec)
{
    __shared__ double red[kBlock / 64];
    for (int v = 0; v < nvec; ++v) {
        const double t = fold_partials(partials + (size_t)v * stride, np, red);
        if (threadIdx.x == 0) out[v] = t;
    }
}

void launch_sum_partials(const Launch &L, const double *partials, int np, int stride, double *out, int nvec)
{
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, L.stream, partials, np, stride, out, nvec);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void axpby_kernel(int n, double a, const double *__restrict__ x, double b,
                                                        double *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        y[i] = (b != 0.0) ? a * x[i] + b * y[i] : a * x[i];
}

void launch_axpby(const Launch &L, int n, double a, const double *x, double b, double *y)
{
    hipLaunchKernelGGL(axpby_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, a, x, b, y);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void fill_kernel(int n, double v, double *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = v;
}

void launch_fill(const Launch &L, int n, double v, double *x)
{
    hipLaunchKernelGGL(fill_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, v, x);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void vmul_kernel(int n, const double *__restrict__ d,
                                                       const double *__restrict__ r, double *__restrict__ z)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) z[i] = d ? d[i] * r[i] : r[i];
}

void launch_vmul(const Launch &L, int n, const double *d, const double *r, double *z)
{
    hipLaunchKernelGGL(vmul_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, d, r, z);
    PS_HIP_CHECK(hipGetLastError());
}

// Eigen::DiagonalPreconditioner::factorize: invdiag = (A(j,j) != 0) ? 1/A(j,j) : 1, duplicates summed.
// bad_count counts rows whose diagonal is non-finite (factorize then fails with ENUMERIC).
template <int GROUP>
__global__ __launch_bounds__(kBlock) void diag_inverse_kernel(int n, const int *__restrict__ rowptr,
                                                               const int *__restrict__ col,
                                                               const double *__restrict__ val,
                                                               double *__restrict__ invdiag, int *bad_count)
{
    // GROUP lanes look through a row (wide rows: one thread per row walked hundreds of entries on the coarse levels)
    const int lane = threadIdx.x % GROUP;
    const int groups = gridDim.x * kBlock / GROUP;
    for (int r = (blockIdx.x * kBlock + threadIdx.x) / GROUP; r < n; r += groups) {
        const int rs = rowptr[r], re = rowptr[r + 1];
        double d = 0.0;
        int hits = 0;
        for (int j = rs + lane; j < re; j += GROUP)
            if (col[j] == r) {
                d += val[j];
                ++hits;
            }
        if (GROUP > 1) {
#pragma unroll
            for (int off = GROUP >> 1; off > 0; off >>= 1) {
                hits += __shfl_xor(hits, off);
                d += __shfl_xor(d, off); // (one hit: the other lanes add zeros)
            }
            if (hits > 1) { // duplicates are summed in the order they are stored
                d = 0.0;
                for (int j = rs; j < re; ++j)
                    if (col[j] == r) d += val[j];
            }
        }
        if (lane == 0) {
            if (!isfinite(d)) atomicAdd(bad_count, 1);
            invdiag[r] = (d != 0.0) ? 1.0 / d : 1.0;
        }
    }
}

void launch_diag_inverse(const Launch &L, const CsrDev &A, double *invdiag, int *bad_count)
{
    const double avg = A.n > 0 ? (double)A.nnz / (double)A.n : 0.0;
    if (avg > 32.0)
        hipLaunchKernelGGL(diag_inverse_kernel<32>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           invdiag, bad_count);
    else if (avg > 8.0)
        hipLaunchKernelGGL(diag_inverse_kernel<8>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           invdiag, bad_count);
    else
        hipLaunchKernelGGL(diag_inverse_kernel<1>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           invdiag, bad_count);
    PS_HIP_CHECK(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------
// Block value types (block_size B): node-local B x B work, one thread per node
// ---------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ void invert_small(const double *X, double *Y, bool &bad)
{
    double a[B * B], inv[B * B];
#pragma unroll
    for (int i = 0; i < B * B; ++i) {
        a[i] = X[i];
        inv[i] = (i % (B + 1) == 0) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < B; ++c) {
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < B; ++r)
            if (r > c && fabs(a[r * B + c]) > fabs(a[piv * B + c])) piv = r;
#pragma unroll
        for (int r = 0; r < B; ++r)
            if (r == piv && piv != c) {
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    double t = a[c * B + k]; a[c * B + k] = a[r * B + k]; a[r * B + k] = t;
                    t = inv[c * B + k]; inv[c * B + k] = inv[r * B + k]; inv[r * B + k] = t;
                }
            }
        const double d = 1.0 / a[c * B + c];
        if (!isfinite(d)) bad = true;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            a[c * B + k] *= d;
            inv[c * B + k] *= d;
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            if (r == c) continue;
            const double f = a[r * B + c];
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

template <int B>
__global__ __launch_bounds__(kBlock) void block_diag_inverse_kernel(int nb, const int *__restrict__ rowptr,
                                                                     const int *__restrict__ col,
                                                                     const double *__restrict__ val,
                                                                     double *__restrict__ dinv_blk, int *bad_count)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double D[B * B];
#pragma unroll
        for (int k = 0; k < B * B; ++k) D[k] = 0.0;
        bool found = false;
        for (int r = 0; r < B; ++r)
            for (int j = rowptr[i * B + r]; j < rowptr[i * B + r + 1]; ++j) {
                const int c = col[j] - i * B;
                if (c >= 0 && c < B) {
                    D[r * B + c] += val[j];
                    found = true;
                }
            }
        if (!found)
            for (int k = 0; k < B; ++k) D[k * B + k] = 1.0;
        double Y[B * B];
        bool bad = false;
        invert_small<B>(D, Y, bad);
        if (bad) atomicAdd(bad_count, 1);
#pragma unroll
        for (int k = 0; k < B * B; ++k) dinv_blk[(size_t)i * B * B + k] = Y[k];
    }
}

// the same from a block copy of the operator (bval: zero-filled b x b blocks; didx[i]: position of block row i's diagonal
// block, -1: none -> identity): one 72-byte read per node instead of a walk over its three scalar rows (elasticity M = 100,
// level 0: 3.66 ms -> the time of a 100 MB stream)
template <int B>
__global__ __launch_bounds__(kBlock) void block_diag_inverse_bsr_kernel(int nb, const int *__restrict__ didx,
                                                                         const double *__restrict__ bval,
                                                                         double *__restrict__ dinv_blk, int *bad_count)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double D[B * B];
        const int d = didx[i];
#pragma unroll
        for (int k = 0; k < B * B; ++k) D[k] = d >= 0 ? bval[(size_t)d * B * B + k] : ((k % (B + 1) == 0) ? 1.0 : 0.0);
        double Y[B * B];
        bool bad = false;
        invert_small<B>(D, Y, bad);
        if (bad) atomicAdd(bad_count, 1);
#pragma unroll
        for (int k = 0; k < B * B; ++k) dinv_blk[(size_t)i * B * B + k] = Y[k];
    }
}

void launch_block_diag_inverse_bsr(const Launch &L, int nb, int bs, const int *didx, const double *bval, double *dinv_blk,
                                   int *bad_count)
{
    PS_REQUIRE(bs == 3 || bs == 2, PSOLVE_HIP_EINVAL, "block_size must be 2 or 3 here");
    if (bs == 3)
        hipLaunchKernelGGL(block_diag_inverse_bsr_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, didx, bval, dinv_blk,
                           bad_count);
    else
        hipLaunchKernelGGL(block_diag_inverse_bsr_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, didx, bval, dinv_blk,
                           bad_count);
    PS_HIP_CHECK(hipGetLastError());
}

void launch_block_diag_inverse(const Launch &L, const CsrDev &A, int bs, double *dinv_blk, int *bad_count)
{
    PS_REQUIRE(bs == 3 || bs == 2, PSOLVE_HIP_EINVAL, "block_size must be 2 or 3 here");
    const int nb = A.n / bs;
    if (bs == 3)
        hipLaunchKernelGGL(block_diag_inverse_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, A.rowptr, A.col,
                           A.val, dinv_blk, bad_count);
    else
        hipLaunchKernelGGL(block_diag_inverse_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, A.rowptr, A.col,
                           A.val, dinv_blk, bad_count);
    PS_HIP_CHECK(hipGetLastError());
}

template <int B>
__global__ __launch_bounds__(kBlock) void block_cheb_update_kernel(int nb, const double *__restrict__ dinv_blk,
                                                                    const double *__restrict__ t,
                                                                    double *__restrict__ p, double *__restrict__ x,
                                                                    double alpha, double beta, int x_is_zero)
{
    // one thread per SCALAR row (round 4): lane j reads row j % B of its node's inverted diagonal block -- 8 B bytes right
    // behind its neighbour's -- and the node's B residuals; p and x are touched lane by lane.  One thread per node had the
    // lanes of a wave 8 B^2 bytes apart on every access (configs[2], the first Chebyshev step of a level-0 solve: 0.48 of
    // peak).  The same products in the same order.
    const long long n = (long long)nb * B;
    for (long long j = (long long)blockIdx.x * kBlock + threadIdx.x; j < n; j += (long long)gridDim.x * kBlock) {
        const long long i = j / B;
        const int r = (int)(j - i * B);
        double res = 0.0;
#pragma unroll
        for (int c = 0; c < B; ++c) res += dinv_blk[(size_t)i * B * B + r * B + c] * t[i * B + c];
        const double pn = (beta != 0.0) ? alpha * res + beta * p[j] : alpha * res;
        p[j] = pn;
        x[j] = x_is_zero ? pn : x[j] + pn;
    }
}

void launch_block_cheb_update(const Launch &L, int n, int bs, const double *dinv_blk, const double *t, double *p,
                              double *x, double alpha, double beta, bool x_is_zero)
{
    const int nb = n / bs;
    if (bs == 3)
        hipLaunchKernelGGL(block_cheb_update_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, dinv_blk, t, p, x,
                           alpha, beta, x_is_zero ? 1 : 0);
    else
        hipLaunchKernelGGL(block_cheb_update_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, dinv_blk, t, p, x,
                           alpha, beta, x_is_zero ? 1 : 0);
    PS_HIP_CHECK(hipGetLastError());
}

template <int B>
__global__ __launch_bounds__(kBlock) void block_power_kernel(int nb, const double *__restrict__ dinv_blk,
                                                              double *__restrict__ t, const double *__restrict__ b0,
                                                              double *__restrict__ partials,
                                                              double *__restrict__ partials2)
{
    __shared__ double red[kBlock / 64];
    double s2 = 0.0, sb = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < nb; i += gridDim.x * kBlock) {
        double tv[B], sv[B];
#pragma unroll
        for (int c = 0; c < B; ++c) tv[c] = t[i * B + c];
        double dotsb = 0.0;
#pragma unroll
        for (int r = 0; r < B; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < B; ++c) v += dinv_blk[(size_t)i * B * B + r * B + c] * tv[c];
            sv[r] = v;
            s2 += v * v;
            dotsb += v * b0[i * B + r];
        }
        sb += fabs(dotsb);
#pragma unroll
        for (int r = 0; r < B; ++r) t[i * B + r] = sv[r];
    }
    const double a = block_sum(s2, red);
    const double b = block_sum(sb, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = a;
        partials2[blockIdx.x] = b;
    }
}

void launch_block_power(const Launch &L, int n, int bs, const double *dinv_blk, double *t, const double *b0,
                        double *partials, double *partials2)
{
    const int nb = n / bs;
    if (bs == 3)
        hipLaunchKernelGGL(block_power_kernel<3>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, dinv_blk, t, b0, partials,
                           partials2);
    else
        hipLaunchKernelGGL(block_power_kernel<2>, dim3(L.grid), dim3(kBlock), 0, L.stream, nb, dinv_blk, t, b0, partials,
                           partials2);
    PS_HIP_CHECK(hipGetLastError());
}


// ---------------------------------------------------------------------------------------------
// AMG numeric setup kernels (device-side refresh of a hierarchy whose patterns are known)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hash_i32_kernel(int64_t n, const int *__restrict__ data,
                                                           unsigned long long *out)
{
    unsigned long long h = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        unsigned long long z = ((unsigned long long)(unsigned)data[i] << 32) ^ (unsigned long long)i;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h += z ^ (z >> 31);
    }
    if (h) atomicAdd(out, h);
}

void launch_hash_i32(const Launch &L, int64_t n, const int *data, unsigned long long *out)
{
    hipLaunchKernelGGL(hash_i32_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, data, out);
    PS_HIP_CHECK(hipGetLastError());
}

// *out += sum of mix(i) over the stored entries with val[i] != 0: identity of the "stored value is nonzero"
// flags, i.e. of the eps_strong = 0 strength graph (a_ij^2 > 0) a hierarchy's patterns were built from
__global__ __launch_bounds__(kBlock) void hash_nonzero_kernel(int64_t n, const double *__restrict__ val,
                                                               unsigned long long *out)
{
    unsigned long long h = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (val[i] == 0.0) continue;
        unsigned long long z = (unsigned long long)i + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h += z ^ (z >> 31);
    }
    if (h) atomicAdd(out, h);
}

void launch_hash_nonzero(const Launch &L, int64_t n, const double *val, unsigned long long *out)
{
    hipLaunchKernelGGL(hash_nonzero_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, val, out);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void gershgorin_kernel(int n, const int *__restrict__ rowptr,
                                                             const int *__restrict__ col,
                                                             const double *__restrict__ val,
                                                             double *__restrict__ partials)
{
    __shared__ double red[kBlock / 64];
    double m = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        double s = 0.0, dia = 1.0;
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            s += fabs(val[j]);
            if (col[j] == r) dia = val[j];
        }
        s *= fabs(1.0 / dia);
        m = fmax(m, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(m));
        int hi = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(m));
        m = fmax(m, __hiloint2double(hi, lo));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Rows of a dozen entries and more (coarse levels: 31 per row on level 1 of the 256^3 hierarchy): one thread per row reads its
// row 12 bytes at a time with the lanes of a wave ~370 bytes apart (1.8 ms for 0.76 GB).  Here G lanes share a row: they bring
// G entries at a time into the group's slice of LDS with one coalesced load each, and ONE lane adds them up in row order -- the
// sums are the kernel's above, bit for bit (round 4).
#define PS_WAVE_SYNC_K()                                       \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

template <int G>
__global__ __launch_bounds__(kBlock) void gershgorin_rows_kernel(int n, const int *__restrict__ rowptr,
                                                                  const int *__restrict__ col,
                                                                  const double *__restrict__ val,
                                                                  double *__restrict__ partials)
{
    __shared__ int lcol[kBlock];
    __shared__ double lval[kBlock];
    __shared__ double red[kBlock / 64];
    const int lane = threadIdx.x % G, base = threadIdx.x - lane;
    const int groups = gridDim.x * (kBlock / G);
    double m = 0.0;
    for (int r = blockIdx.x * (kBlock / G) + threadIdx.x / G; r < n; r += groups) {
        const int rs = rowptr[r], re = rowptr[r + 1];
        double s = 0.0, dia = 1.0;
        for (int c0 = rs; c0 < re; c0 += G) {
            const int j = c0 + lane;
            if (j < re) {
                lcol[threadIdx.x] = col[j];
                lval[threadIdx.x] = val[j];
            }
            PS_WAVE_SYNC_K();
            if (lane == 0) {
                const int cnt = min(G, re - c0);
                for (int t = 0; t < cnt; ++t) {
                    const double a = lval[base + t];
                    s += fabs(a);
                    if (lcol[base + t] == r) dia = a;
                }
            }
            PS_WAVE_SYNC_K();
        }
        if (lane == 0) {
            s *= fabs(1.0 / dia);
            m = fmax(m, s);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int lo = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2loint(m));
        int hi = __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ off) << 2, __double2hiint(m));
        m = fmax(m, __hiloint2double(hi, lo));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

void launch_gershgorin(const Launch &L, const CsrDev &A, double *partials)
{
    if (A.n > 0 && A.nnz >= 12ll * A.n)
        hipLaunchKernelGGL(gershgorin_rows_kernel<16>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           partials);
    else
        hipLaunchKernelGGL(gershgorin_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col, A.val,
                           partials);
    PS_HIP_CHECK(hipGetLastError());
}

// amgcl/coarsening/smoothed_aggregation.hpp: P = (I - omega D_f^-1 A_f) P_tent with P_tent(i, id[i]) = 1.
// A_f drops the weak links (eps^2 a_ii a_jj >= a_ij^2; with eps = 0: the exactly-zero entries) and adds
// them to its diagonal.  dia == nullptr means eps = 0.
__global__ __launch_bounds__(kBlock) void prolongation_values_kernel(int n, const int *__restrict__ rowptr,
                                                                      const int *__restrict__ col,
                                                                      const double *__restrict__ val,
                                                                      const int *__restrict__ id, double omega,
                                                                      const double *__restrict__ dia, double eps2,
                                                                      const int *__restrict__ pptr,
                                                                      const int *__restrict__ pcol,
                                                                      double *__restrict__ pval)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int pb = pptr[i], pe = pptr[i + 1];
        const double eps_dia_i = dia ? eps2 * dia[i] : 0.0;
        if (pe - pb <= 8) {
            // short rows of P (the fine levels: a handful of aggregates per row): the row is summed in registers and
            // stored once -- the same additions in the same order, without a chain of read-modify-writes in HBM
            int pc[8];
            double acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pc[k] = pb + k < pe ? pcol[pb + k] : -1;
                acc[k] = 0.0;
            }
            const int rs = rowptr[i], re = rowptr[i + 1];
            double dsum = 0.0;
            for (int j = rs; j < re; ++j) {
                const int ca = col[j];
                const double a = val[j];
                const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
                if (!strong) dsum += a;
            }
            const double f = -omega * (1.0 / dsum);
            for (int j = rs; j < re; ++j) {
                const int ca = col[j];
                const double a = val[j];
                const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
                if (ca != i && !strong) continue;
                const int cp = id[ca];
                if (cp < 0) continue;
                const double va = (ca == i) ? (1.0 - omega) : f * a;
                bool done = false;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!done && pc[k] == cp) { // (the first match, as the loop below)
                        acc[k] += va;
                        done = true;
                    }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (pb + k < pe) pval[pb + k] = acc[k];
            continue;
        }
        for (int k = pb; k < pe; ++k) pval[k] = 0.0;
        double dsum = 0.0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int ca = col[j];
            const double a = val[j];
            const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
            if (!strong) dsum += a; // the diagonal and the weak links
        }
        const double f = -omega * (1.0 / dsum);
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int ca = col[j];
            const double a = val[j];
            const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
            if (ca != i && !strong) continue;
            const int cp = id[ca];
            if (cp < 0) continue;
            const double va = (ca == i) ? (1.0 - omega) : f * a;
            for (int k = pb; k < pe; ++k)
                if (pcol[k] == cp) {
                    pval[k] += va;
                    break;
                }
        }
    }
}

// the same for rows of a dozen entries and more: G lanes per row stage G entries at a time (column, value, aggregate of the
// column, whether the entry counts) in LDS; lane 0 sums the filtered diagonal in row order, then lane k of the group owns
// entry k of the row of P and adds the staged contributions that belong to it in row order -- the additions of the kernel
// above in the same order (round 4; level 1 of the 256^3 hierarchy: 2.9 ms with one thread per 31-entry row)
template <int G>
__global__ __launch_bounds__(kBlock) void prolongation_values_rows_kernel(int n, const int *__restrict__ rowptr,
                                                                           const int *__restrict__ col,
                                                                           const double *__restrict__ val,
                                                                           const int *__restrict__ id, double omega,
                                                                           const double *__restrict__ dia, double eps2,
                                                                           const int *__restrict__ pptr,
                                                                           const int *__restrict__ pcol,
                                                                           double *__restrict__ pval)
{
    __shared__ int lcp[kBlock];      // aggregate of the entry's column, -1: the entry contributes nothing
    __shared__ double lva[kBlock];   // pass 1: the value if it belongs to the filtered diagonal, else 0 (flag in lcp); pass 2: its contribution
    __shared__ double lf[kBlock / G];
    const int lane = threadIdx.x % G, base = threadIdx.x - lane, grp = threadIdx.x / G;
    const int groups = gridDim.x * (kBlock / G);
    for (int i = blockIdx.x * (kBlock / G) + grp; i < n; i += groups) {
        const int rs = rowptr[i], re = rowptr[i + 1];
        const int pb = pptr[i], pe = pptr[i + 1];
        const double eps_dia_i = dia ? eps2 * dia[i] : 0.0;
        // pass 1: the filtered diagonal = the diagonal and the weak links, in row order
        double dsum = 0.0;
        for (int c0 = rs; c0 < re; c0 += G) {
            const int j = c0 + lane;
            if (j < re) {
                const int ca = col[j];
                const double a = val[j];
                const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
                lcp[threadIdx.x] = strong ? 0 : 1;
                lva[threadIdx.x] = a;
            }
            PS_WAVE_SYNC_K();
            if (lane == 0) {
                const int cnt = min(G, re - c0);
                for (int t = 0; t < cnt; ++t)
                    if (lcp[base + t]) dsum += lva[base + t];
            }
            PS_WAVE_SYNC_K();
        }
        if (lane == 0) lf[grp] = -omega * (1.0 / dsum);
        PS_WAVE_SYNC_K();
        const double f = lf[grp];
        // pass 2: contributions, added per entry of P in row order (lane k owns entries k, k + G, ... of the row of P)
        for (int k = pb + lane; k < pe; k += G) pval[k] = 0.0;
        for (int c0 = rs; c0 < re; c0 += G) {
            const int j = c0 + lane;
            if (j < re) {
                const int ca = col[j];
                const double a = val[j];
                const bool strong = (ca != i) && ((eps_dia_i != 0.0 ? eps_dia_i * dia[ca] : 0.0) < a * a);
                int cp = -1;
                if (ca == i || strong) cp = id[ca];
                lcp[threadIdx.x] = cp < 0 ? -1 : cp;
                lva[threadIdx.x] = (ca == i) ? (1.0 - omega) : f * a;
            }
            PS_WAVE_SYNC_K();
            const int cnt = min(G, re - c0);
            for (int k = pb + lane; k < pe; k += G) {
                const int want = pcol[k];
                double acc = pval[k];
                for (int t = 0; t < cnt; ++t)
                    if (lcp[base + t] == want) acc += lva[base + t];
                pval[k] = acc;
            }
            PS_WAVE_SYNC_K();
        }
    }
}

void launch_prolongation_values(const Launch &L, const CsrDev &A, const int *id, double omega, const double *dia,
                                double eps_strong, CsrMut P)
{
    if (A.n > 0 && A.nnz >= 12ll * A.n)
        hipLaunchKernelGGL(prolongation_values_rows_kernel<16>, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col,
                           A.val, id, omega, dia, eps_strong * eps_strong, P.rowptr, P.col, P.val);
    else
        hipLaunchKernelGGL(prolongation_values_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rowptr, A.col,
                           A.val, id, omega, dia, eps_strong * eps_strong, P.rowptr, P.col, P.val);
    PS_HIP_CHECK(hipGetLastError());
}

// C = A * B with C's pattern (sorted columns) known.  LPR lanes share a row of C; each lane owns output
// entries pos, pos + LPR, ... and, for its column c, walks row i of A in order and looks c up in the
// (sorted) row of B: every output entry is summed in ascending-k order, exactly like the host
// Gustavson product, so the result is deterministic and equal to the host's bit for bit.
template <int LPR>
__global__ __launch_bounds__(kBlock) void spgemm_numeric_kernel(int n, const int *__restrict__ cptr,
                                                                 const int *__restrict__ ccol,
                                                                 double *__restrict__ cval,
                                                                 const int *__restrict__ aptr,
                                                                 const int *__restrict__ acol,
                                                                 const double *__restrict__ aval,
                                                                 const int *__restrict__ bptr,
                                                                 const int *__restrict__ bcol,
                                                                 const double *__restrict__ bval)
{
    const int lane = threadIdx.x % LPR;
    const int rows_per_pass = (gridDim.x * kBlock) / LPR;
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / LPR; i < n; i += rows_per_pass) {
        const int cb = cptr[i], ce = cptr[i + 1];
        const int ab = aptr[i], ae = aptr[i + 1];
        for (int pos = cb + lane; pos < ce; pos += LPR) {
            const int c = ccol[pos];
            double sum = 0.0;
            for (int ja = ab; ja < ae; ++ja) {
                const int ca = acol[ja];
                int lo = bptr[ca], hi = bptr[ca + 1];
                const int end = hi;
                while (lo < hi) {
                    const int mid = lo + ((hi - lo) >> 1); // lo + hi can pass 2^31
                    if (bcol[mid] < c) lo = mid + 1; else hi = mid;
                }
                if (lo < end && bcol[lo] == c) sum += aval[ja] * bval[lo];
            }
            cval[pos] = sum;
        }
    }
}

// The same product, row-wise (Gustavson) with the output row parked in LDS: the LPR lanes of a row keep its sorted
// columns and a zeroed accumulator there, walk row i of A in order and, for entry (i, k), spread over row k of B --
// lane l takes b_kj, finds j in the parked columns by bisection and adds a_ik b_kj to its slot.  Every slot gets
// at most one term per k (the columns of a row of B are distinct) and the k's come in ascending order, so each
// output entry is the same sequence of additions as above, bit for bit; the memory operations drop from
// nnz(C_i) * nnz(A_i) * log nnz(B_k) to nnz(A_i) * nnz(B_k) per row (R (A P) at level 1 of the 216^3 hierarchy:
// ~3100 -> ~300).  The lanes of a row share a wave, whose LDS operations execute in program order: no barrier
// between the k's.  Rows longer than the LDS slot (6 LPR entries up to LPR = 8, else 8 LPR) take the per-entry search.
template <int LPR>
__global__ __launch_bounds__(kBlock) void spgemm_numeric_lds_kernel(int n, const int *__restrict__ cptr,
                                                                     const int *__restrict__ ccol,
                                                                     double *__restrict__ cval,
                                                                     const int *__restrict__ aptr,
                                                                     const int *__restrict__ acol,
                                                                     const double *__restrict__ aval,
                                                                     const int *__restrict__ bptr,
                                                                     const int *__restrict__ bcol,
                                                                     const double *__restrict__ bval)
{
    // 104 LPR bytes of LDS per row group -> 26.6 KiB per workgroup, six of them (24 waves) per CU: the kernel is a chain of
    // dependent loads per row, hidden only by other waves (round 4; with 8 LPR slots and a table of 16 LPR: 40 KiB, four)
    // (up to eight lanes per row -- short rows of B, short output rows; 16 lanes and more keep 8 LPR slots and a table of
    // 16 LPR: with the smaller table R (A P) of the 256^3 hierarchy's level 0 took 7.8 ms instead of 3.1 -- probe chains)
    constexpr int CAP = (LPR <= 8 ? 6 : 8) * LPR, GROUPS = kBlock / LPR, TS = (LPR <= 8 ? 8 : 16) * LPR;
    __shared__ int lcol[GROUPS][CAP];
    __shared__ double lacc[GROUPS][CAP];
    __shared__ int ltab[GROUPS][TS]; // column -> slot of the parked row (open addressing, at most half full)
    const int lane = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int rows_per_pass = (gridDim.x * kBlock) / LPR;
    int *mycol = lcol[grp];
    double *myacc = lacc[grp];
    int *mytab = ltab[grp];
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / LPR; i < n; i += rows_per_pass) {
        const int cb = cptr[i], ce = cptr[i + 1], len = ce - cb;
        const int ab = aptr[i], ae = aptr[i + 1];
        if (len > CAP) { // (uniform over the row's lanes)
            for (int pos = cb + lane; pos < ce; pos += LPR) {
                const int c = ccol[pos];
                double sum = 0.0;
                for (int ja = ab; ja < ae; ++ja) {
                    const int ca = acol[ja];
                    int lo = bptr[ca], hi = bptr[ca + 1];
                    const int end = hi;
                    while (lo < hi) {
                        const int mid = lo + ((hi - lo) >> 1);
                        if (bcol[mid] < c) lo = mid + 1; else hi = mid;
                    }
                    if (lo < end && bcol[lo] == c) sum += aval[ja] * bval[lo];
                }
                cval[pos] = sum;
            }
            continue;
        }
        for (int t = lane; t < TS; t += LPR) mytab[t] = -1;
        for (int t = lane; t < len; t += LPR) {
            const int c = ccol[cb + t];
            mycol[t] = c;
            myacc[t] = 0.0;
            unsigned slot = ((unsigned)c * 2654435761u >> 12) & (TS - 1);
            while (atomicCAS(&mytab[slot], -1, t) != -1) slot = (slot + 1) & (TS - 1);
        }
        // a probe or two in the table instead of a bisection of the parked columns (seven steps for 120)
        auto add = [&](int j, double v) {
            unsigned slot = ((unsigned)j * 2654435761u >> 12) & (TS - 1);
            int t = mytab[slot];
            while (t >= 0 && mycol[t] != j) {
                slot = (slot + 1) & (TS - 1);
                t = mytab[slot];
            }
            if (t >= 0) myacc[t] += v;
        };
        // row i of A, LPR entries at a time: lane l fetches entry l and the extent of ITS row of B (one round of
        // dependent loads for the whole batch instead of one per entry); the entries are then taken in order, the
        // first stride of the next row of B already in flight
        const int gbase = (threadIdx.x & 63) / LPR * LPR;
        for (int base = ab; base < ae; base += LPR) {
            const int ja = base + lane;
            int bb_l = 0, be_l = 0;
            double a_l = 0.0;
            if (ja < ae) {
                const int ca = acol[ja];
                a_l = aval[ja];
                bb_l = bptr[ca];
                be_l = bptr[ca + 1];
            }
            const int cnt = min(LPR, ae - base);
            int bbn = __shfl(bb_l, gbase), ben = __shfl(be_l, gbase);
            int jn = 0;
            double vn = 0.0;
            if (bbn + lane < ben) {
                jn = bcol[bbn + lane];
                vn = bval[bbn + lane];
            }
            for (int t = 0; t < cnt; ++t) {
                const double a = __shfl(a_l, gbase + t);
                const int bb = bbn, be = ben, j0 = jn;
                const double v0 = vn;
                if (t + 1 < cnt) {
                    bbn = __shfl(bb_l, gbase + t + 1);
                    ben = __shfl(be_l, gbase + t + 1);
                    if (bbn + lane < ben) {
                        jn = bcol[bbn + lane];
                        vn = bval[bbn + lane];
                    }
                }
                int jb = bb + lane;
                if (jb < be) add(j0, a * v0);
                for (jb += LPR; jb < be; jb += LPR) add(bcol[jb], a * bval[jb]);
            }
        }
        for (int t = lane; t < len; t += LPR) cval[cb + t] = myacc[t];
    }
}

void launch_spgemm_numeric(const Launch &L, CsrMut C, const CsrDev &A, const CsrDev &B, double avg_c_row)
{
    dim3 g(L.grid), blk(kBlock);
#define PS_SPGEMM(LPR)                                                                                           \
    hipLaunchKernelGGL(spgemm_numeric_lds_kernel<LPR>, g, blk, 0, L.stream, C.n, C.rowptr, C.col, C.val, A.rowptr, A.col, \
                       A.val, B.rowptr, B.col, B.val)
    // the lanes of a row spread over a row of B (so: as many as that row is long), and the LDS slot of 6 / 8 LPR entries
    // should hold the typical row of C with room to spare
    const double avg_b_row = B.n > 0 ? (double)B.nnz / (double)B.n : 1.0;
    int lpr = 4;
    while (lpr < 64 && ((double)lpr < avg_b_row || (lpr <= 8 ? 6.0 : 8.0) * lpr < 1.5 * avg_c_row)) lpr *= 2;
    if (lpr <= 8) g = dim3((unsigned)std::max(8, std::min(L.grid, 6 * L.num_cus))); // (what is resident: 26.6 KiB of LDS each)
    if (lpr == 4) PS_SPGEMM(4);
    else if (lpr == 8) PS_SPGEMM(8);
    else if (lpr == 16) PS_SPGEMM(16);
    else if (lpr == 32) PS_SPGEMM(32);
    else PS_SPGEMM(64);
#undef PS_SPGEMM
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Fused PCG steps -- Eigen::internal::conjugate_gradient's recurrence (oracle: orc_cg_eigen)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pcg_init_dir_kernel(int n, const double *__restrict__ invdiag,
                                                               const double *__restrict__ r,
                                                               double *__restrict__ p,
                                                               double *__restrict__ partials_rz)
{
    __shared__ double red[kBlock / 64];
    double s = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double ri = r[i];
        const double z = invdiag ? invdiag[i] * ri : ri;
        p[i] = z;
        s += ri * z;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) partials_rz[blockIdx.x] = t;
}

void launch_pcg_init_dir(const Launch &L, int n, const double *invdiag, const double *r, double *p,
                         double *partials_rz)
{
    hipLaunchKernelGGL(pcg_init_dir_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, invdiag, r, p, partials_rz);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void pcg_init_state_kernel(PcgState *S, const double *part_rr,
                                                                 const double *part_bb, const double *part_rz,
                                                                 int np_rr, int np_bb, int np_rz, double rel_tol,
                                                                 double abs_tol)
{
    __shared__ double red[kBlock / 64];
    const double rr = fold_partials(part_rr, np_rr, red);
    const double bb = fold_partials(part_bb, np_bb, red);
    const double rz = part_rz ? fold_partials(part_rz, np_rz, red) : 0.0;
    if (threadIdx.x == 0) {
        double thr = rel_tol * rel_tol * bb;
        const double abs2 = abs_tol * abs_tol;
        if (thr < abs2) thr = abs2;
        if (thr < DBL_MIN) thr = DBL_MIN;
        S->rhs_norm2 = bb;
        S->threshold = thr;
        S->abs2 = abs2;
        S->rn2 = rr;
        S->rn2_init = rr;
        S->rz[0] = rz;
        S->rz[1] = 0.0;
        S->passes = 0;
        S->zero_rhs = (bb == 0.0);
        const bool bad = !isfinite(rr) || !isfinite(bb) || !isfinite(rz);
        const int conv = bad || (bb == 0.0) || (rr < thr);
        S->done[0] = conv;
        S->done[1] = conv;
        S->status = bad ? PSOLVE_HIP_NONFINITE_RESIDUAL
                        : (conv ? ((rr < abs2) ? PSOLVE_HIP_REACH_ABSOLUTE_TOLERANCE : PSOLVE_HIP_REACH_RELATIVE_TOLERANCE)
                                : PSOLVE_HIP_RUNNING);
    }
}

void launch_pcg_init_state(const Launch &L, PcgState *S, const double *part_rr, const double *part_bb,
                           const double *part_rz, int np_rr, int np_bb, int np_rz, double rel_tol, double abs_tol)
{
    hipLaunchKernelGGL(pcg_init_state_kernel, dim3(1), dim3(kBlock), 0, L.stream, S, part_rr, part_bb, part_rz, np_rr,
                       np_bb, np_rz, rel_tol, abs_tol);
    PS_HIP_CHECK(hipGetLastError());
}

// K2: r -= alpha q ; partial r.r and r.(M^-1 r)
// POL: cache policy of the streams, bit 0 = non-temporal loads, bit 1 = non-temporal store of r
template <int POL>
__global__ __launch_bounds__(kBlock) void pcg_update_r_kernel(int n, int parity, const PcgState *__restrict__ S,
                                                               const double *__restrict__ part_pq, int np_pq,
                                                               const double *__restrict__ invdiag,
                                                               const double *__restrict__ q, double *__restrict__ r,
                                                               double *__restrict__ part_rr,
                                                               double *__restrict__ part_rz,
                                                               const unsigned short *__restrict__ kind,
                                                               const double *__restrict__ ktab, int nk)
{
    __shared__ double red[kBlock / 64];
    // (row kinds, Launch::kd_*: invdiag[i] read as ktab[kind[i]] -- 2 bytes per row instead of 8, the same value)
    __shared__ double ltab[kKindTabMax];
    if (S->done[parity]) return;
    if (kind) {
        for (int t = threadIdx.x; t < nk; t += kBlock) ltab[t] = ktab[t];
        __syncthreads();
    }
    // Round 3: the vectors of the first two steps are requested BEFORE the partial sums of p.q are folded (every
    // workgroup folds them itself: a few microseconds in which nothing streamed -- 5 % of the kernel), and two steps
    // stay in flight afterwards (96 bytes per thread instead of 48).  Same elements per thread, same order of sums.
    constexpr bool NTL = (POL & 1) != 0;
    const int n2 = n >> 1, stride = gridDim.x * kBlock;
    int i = blockIdx.x * kBlock + threadIdx.x, j = i + stride;
    const v2d zero2 = {0.0, 0.0};
    v2d qa = zero2, ra = zero2, da = zero2, qb = zero2, rb = zero2, db = zero2;
    unsigned ka = 0, kb = 0; // (the kinds of the two rows of a step)
    if (i < n2) {
        qa = load_stream2<NTL>(q + 2 * (size_t)i);
        ra = load_stream2<NTL>(r + 2 * (size_t)i);
        if (kind) ka = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)i);
        else if (invdiag) da = load_stream2<NTL>(invdiag + 2 * (size_t)i);
    }
    if (j < n2) {
        qb = load_stream2<NTL>(q + 2 * (size_t)j);
        rb = load_stream2<NTL>(r + 2 * (size_t)j);
        if (kind) kb = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)j);
        else if (invdiag) db = load_stream2<NTL>(invdiag + 2 * (size_t)j);
    }
    const double pq = fold_partials(part_pq, np_pq, red);
    const double alpha = S->rz[parity] / pq;
    double srr = 0.0, srz = 0.0;
    while (i < n2) {
        if (kind) {
            da.x = ltab[ka & 0xffffu];
            da.y = ltab[ka >> 16];
        }
        v2d rv = ra;
        rv.x -= alpha * qa.x;
        rv.y -= alpha * qa.y;
        store_stream2<(POL & 2) != 0>(r + 2 * (size_t)i, rv);
        srr += rv.x * rv.x;
        srr += rv.y * rv.y;
        if (invdiag) {
            srz += rv.x * (da.x * rv.x);
            srz += rv.y * (da.y * rv.y);
        }
        i = j;
        qa = qb;
        ra = rb;
        da = db;
        ka = kb;
        j += stride;
        if (j < n2) {
            qb = load_stream2<NTL>(q + 2 * (size_t)j);
            rb = load_stream2<NTL>(r + 2 * (size_t)j);
            if (kind) kb = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)j);
            else if (invdiag) db = load_stream2<NTL>(invdiag + 2 * (size_t)j);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int i = n - 1;
        const double ri = r[i] - alpha * q[i];
        r[i] = ri;
        srr += ri * ri;
        if (invdiag) srz += ri * ((kind ? ltab[kind[i]] : invdiag[i]) * ri);
    }
    const double trr = block_sum(srr, red);
    const double trz = invdiag ? block_sum(srz, red) : trr;
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = trr;
        part_rz[blockIdx.x] = trz;
    }
}

void launch_pcg_update_r(const Launch &L, int n, int parity, const PcgState *S, const double *part_pq, int np_pq,
                         const double *invdiag, const double *q, double *r, double *part_rr, double *part_rz)
{
#define PS_K2(P)                                                                                                  \
    case P:                                                                                                       \
        PS_TIMED_LAUNCH(pcg_update_r_kernel<P>, dim3(L.grid), dim3(kBlock), 0, L.stream, n, parity, S, part_pq, np_pq, \
                           invdiag, q, r, part_rr, part_rz, kk, kk ? L.kd_tab : nullptr, kk ? L.kd_n : 0);       \
        break;
    const unsigned short *kk = (invdiag && invdiag == L.kd_for && L.kd_tab && L.kd_n > 0 && L.kd_n <= kKindTabMax) ? L.kd_kind : nullptr;
    if (tl_spmv_kernel_record)
        std::snprintf(tl_vec_kernel_name[0], sizeof(tl_vec_kernel_name[0]), "pcg_update_r_kernel<%d>", L.vec_nt ? (L.vec_policy & 3) : 0);
    switch (L.vec_nt ? (L.vec_policy & 3) : 0) {
        PS_K2(0) PS_K2(1) PS_K2(2) PS_K2(3)
    }
#undef PS_K2
    PS_HIP_CHECK(hipGetLastError());
}

// K3: x += alpha p (always); latch convergence; otherwise p = M^-1 r + beta p
// POL: bit 0 = non-temporal loads, bit 2 = non-temporal store of x, bit 3 = non-temporal store of p
template <int POL>
__global__ __launch_bounds__(kBlock) void pcg_update_xp_kernel(int n, int parity, PcgState *__restrict__ S,
                                                                const double *__restrict__ part_pq, int np_pq,
                                                                const double *__restrict__ part_rr,
                                                                const double *__restrict__ part_rz, int np_rr,
                                                                const double *__restrict__ invdiag,
                                                                const double *__restrict__ r, double *__restrict__ p,
                                                                double *__restrict__ x, int max_iter,
                                                                const unsigned short *__restrict__ kind,
                                                                const double *__restrict__ ktab, int nk)
{
    __shared__ double red[kBlock / 64];
    __shared__ double ltab[kKindTabMax]; // (row kinds: invdiag[i] = ktab[kind[i]], see pcg_update_r_kernel)
    const int done_in = S->done[parity];
    if (done_in) {
        if (blockIdx.x == 0 && threadIdx.x == 0) S->done[parity ^ 1] = 1;
        return;
    }
    if (kind) {
        for (int t = threadIdx.x; t < nk; t += kBlock) ltab[t] = ktab[t];
        __syncthreads();
    }
    // (the first two steps' vectors are requested before the three folds, see K2)
    constexpr bool NTL = (POL & 1) != 0;
    const int n2 = n >> 1, stride = gridDim.x * kBlock;
    int i = blockIdx.x * kBlock + threadIdx.x, j = i + stride;
    const v2d zero2 = {0.0, 0.0};
    v2d pa = zero2, xa = zero2, ra = zero2, da = zero2, pb = zero2, xb = zero2, rb = zero2, db = zero2;
    unsigned ka = 0, kb = 0;
    if (i < n2) {
        pa = load_stream2<NTL>(p + 2 * (size_t)i);
        xa = load_stream2<NTL>(x + 2 * (size_t)i);
        ra = load_stream2<NTL>(r + 2 * (size_t)i);
        if (kind) ka = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)i);
        else if (invdiag) da = load_stream2<NTL>(invdiag + 2 * (size_t)i);
    }
    if (j < n2) {
        pb = load_stream2<NTL>(p + 2 * (size_t)j);
        xb = load_stream2<NTL>(x + 2 * (size_t)j);
        rb = load_stream2<NTL>(r + 2 * (size_t)j);
        if (kind) kb = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)j);
        else if (invdiag) db = load_stream2<NTL>(invdiag + 2 * (size_t)j);
    }
    const double pq = fold_partials(part_pq, np_pq, red);
    const double rn2 = fold_partials(part_rr, np_rr, red);
    const double rz_new = fold_partials(part_rz, np_rr, red);
    const double rz_old = S->rz[parity];
    const double alpha = rz_old / pq;
    const bool bad = !isfinite(rn2) || !isfinite(alpha); // NaN/Inf data or breakdown (p.Ap == 0): stop now
    const bool conv = bad || rn2 < S->threshold;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int passes = S->passes + 1;
        S->passes = passes;
        S->rn2 = rn2;
        S->rz[parity ^ 1] = rz_new;
        S->done[parity ^ 1] = conv ? 1 : 0;
        if (bad)
            S->status = PSOLVE_HIP_NONFINITE_RESIDUAL;
        else if (conv)
            S->status = (rn2 < S->abs2) ? PSOLVE_HIP_REACH_ABSOLUTE_TOLERANCE : PSOLVE_HIP_REACH_RELATIVE_TOLERANCE;
    }
    if (bad) return; // leave x at the last finite iterate
    const double beta = rz_new / rz_old;
    while (i < n2) {
        v2d pv = pa, xv = xa;
        xv.x += alpha * pv.x;
        xv.y += alpha * pv.y;
        store_stream2<(POL & 4) != 0>(x + 2 * (size_t)i, xv);
        if (!conv) {
            v2d zv = ra;
            if (kind) {
                da.x = ltab[ka & 0xffffu];
                da.y = ltab[ka >> 16];
            }
            if (invdiag) {
                zv.x = da.x * ra.x;
                zv.y = da.y * ra.y;
            }
            pv.x = zv.x + beta * pv.x;
            pv.y = zv.y + beta * pv.y;
            store_stream2<(POL & 8) != 0>(p + 2 * (size_t)i, pv);
        }
        i = j;
        pa = pb;
        xa = xb;
        ra = rb;
        da = db;
        ka = kb;
        j += stride;
        if (j < n2) {
            pb = load_stream2<NTL>(p + 2 * (size_t)j);
            xb = load_stream2<NTL>(x + 2 * (size_t)j);
            rb = load_stream2<NTL>(r + 2 * (size_t)j);
            if (kind) kb = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)j);
            else if (invdiag) db = load_stream2<NTL>(invdiag + 2 * (size_t)j);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int i = n - 1;
        const double pi = p[i];
        x[i] += alpha * pi;
        if (!conv) {
            const double z = invdiag ? (kind ? ltab[kind[i]] : invdiag[i]) * r[i] : r[i];
            p[i] = z + beta * pi;
        }
    }
}

void launch_pcg_update_xp(const Launch &L, int n, int parity, PcgState *S, const double *part_pq, int np_pq,
                          const double *part_rr, const double *part_rz, int np_rr, const double *invdiag,
                          const double *r, double *p, double *x, int max_iter)
{
#define PS_K3(P)                                                                                                  \
    case P:                                                                                                       \
        PS_TIMED_LAUNCH(pcg_update_xp_kernel<P>, dim3(L.grid), dim3(kBlock), 0, L.stream, n, parity, S, part_pq, \
                           np_pq, part_rr, part_rz, np_rr, invdiag, r, p, x, max_iter, kk, kk ? L.kd_tab : nullptr,  \
                           kk ? L.kd_n : 0);                                                                      \
        break;
    const unsigned short *kk = (invdiag && invdiag == L.kd_for && L.kd_tab && L.kd_n > 0 && L.kd_n <= kKindTabMax) ? L.kd_kind : nullptr;
    if (tl_spmv_kernel_record)
        std::snprintf(tl_vec_kernel_name[1], sizeof(tl_vec_kernel_name[1]), "pcg_update_xp_kernel<%d>", L.vec_nt ? (L.vec_policy & 13) : 0);
    switch (L.vec_nt ? (L.vec_policy & 13) : 0) {
        PS_K3(0) PS_K3(1) PS_K3(4) PS_K3(5) PS_K3(8) PS_K3(9) PS_K3(12) PS_K3(13)
    }
#undef PS_K3
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Generic-preconditioner PCG steps
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pcg_update_xr_kernel(int n, int parity, const PcgState *__restrict__ S,
                                                                const double *__restrict__ part_pq, int np_pq,
                                                                const double *__restrict__ p,
                                                                const double *__restrict__ q, double *__restrict__ x,
                                                                double *__restrict__ r, double *__restrict__ part_rr)
{
    __shared__ double red[kBlock / 64];
    if (S->done[parity]) return;
    const double pq = fold_partials(part_pq, np_pq, red);
    const double alpha = S->rz[parity] / pq;
    double srr = 0.0;
    // two elements (16 bytes) per lane and access where the vectors are 16-byte aligned (they are: the solver's own buffers)
    const bool al = ((((uintptr_t)p | (uintptr_t)q | (uintptr_t)x | (uintptr_t)r) & 15) == 0);
    const int n2 = al ? (n >> 1) : 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n2; i += gridDim.x * kBlock) {
        const v2d vp = ((const v2d *)p)[i], vq = ((const v2d *)q)[i];
        v2d vx = ((v2d *)x)[i], vr = ((v2d *)r)[i];
        vx.x += alpha * vp.x;
        vx.y += alpha * vp.y;
        vr.x = vr.x - alpha * vq.x;
        vr.y = vr.y - alpha * vq.y;
        ((v2d *)x)[i] = vx;
        ((v2d *)r)[i] = vr;
        srr += vr.x * vr.x;
        srr += vr.y * vr.y;
    }
    for (int i = 2 * n2 + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * q[i];
        r[i] = ri;
        srr += ri * ri;
    }
    const double t = block_sum(srr, red);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = t;
}

void launch_pcg_update_xr(const Launch &L, int n, int parity, const PcgState *S, const double *part_pq, int np_pq,
                          const double *p, const double *q, double *x, double *r, double *part_rr)
{
    hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, parity, S, part_pq, np_pq, p,
                       q, x, r, part_rr);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void pcg_check_kernel(int parity, PcgState *S, const double *part_rr, int np_rr,
                                                            int max_iter)
{
    __shared__ double red[kBlock / 64];
    if (S->done[parity]) {
        if (threadIdx.x == 0) S->done[parity ^ 1] = 1;
        return;
    }
    const double rn2 = fold_partials(part_rr, np_rr, red);
    if (threadIdx.x == 0) {
        const bool bad = !isfinite(rn2);
        const bool conv = bad || rn2 < S->threshold;
        S->passes = S->passes + 1;
        S->rn2 = rn2;
        S->done[parity ^ 1] = conv ? 1 : 0;
        if (bad)
            S->status = PSOLVE_HIP_NONFINITE_RESIDUAL;
        else if (conv)
            S->status = (rn2 < S->abs2) ? PSOLVE_HIP_REACH_ABSOLUTE_TOLERANCE : PSOLVE_HIP_REACH_RELATIVE_TOLERANCE;
    }
}

void launch_pcg_check(const Launch &L, int parity, PcgState *S, const double *part_rr, int np_rr, int max_iter)
{
    hipLaunchKernelGGL(pcg_check_kernel, dim3(1), dim3(kBlock), 0, L.stream, parity, S, part_rr, np_rr, max_iter);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void pcg_update_p_kernel(int n, int parity, PcgState *__restrict__ S,
                                                               const double *__restrict__ part_rz, int np_rz,
                                                               const double *__restrict__ z, double *__restrict__ p)
{
    __shared__ double red[kBlock / 64];
    // done[parity ^ 1] was latched by pcg_check_kernel of THIS iteration (an earlier launch)
    if (S->done[parity ^ 1]) return;
    const double rz_new = fold_partials(part_rz, np_rz, red);
    const double beta = rz_new / S->rz[parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) S->rz[parity ^ 1] = rz_new;
    const bool al = ((((uintptr_t)z | (uintptr_t)p) & 15) == 0);
    const int n2 = al ? (n >> 1) : 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n2; i += gridDim.x * kBlock) {
        const v2d vz = ((const v2d *)z)[i];
        v2d vp = ((v2d *)p)[i];
        vp.x = vz.x + beta * vp.x;
        vp.y = vz.y + beta * vp.y;
        ((v2d *)p)[i] = vp;
    }
    for (int i = 2 * n2 + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] = z[i] + beta * p[i];
}

void launch_pcg_update_p(const Launch &L, int n, int parity, PcgState *S, const double *part_rz, int np_rz,
                         const double *z, double *p)
{
    hipLaunchKernelGGL(pcg_update_p_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, parity, S, part_rz, np_rz, z,
                       p);
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Single-reduction PCG for shards (Chronopoulos & Gear): the same Krylov iterates as the loop above in
// exact arithmetic, but (r.u), (r.r) and (w.u) are reduced TOGETHER, once per iteration, so a sharded
// solve pays one small all-reduce per iteration instead of two:
//     p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = M^-1 r ; w = A u
//     gamma' = r.u, delta = w.u, rr = r.r   -> one all-reduce ->
//     beta' = gamma'/gamma ; alpha' = gamma' / (delta - beta' gamma'/alpha)
// `red` holds the reduced (gamma, rr, delta) of the CURRENT residual; every workgroup reads the same three
// numbers and takes the same decisions.  mode: 0 regular, 1 first iteration (beta = 0), 2 check only.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cg1_update_kernel(int n, int parity, int mode, PcgState *__restrict__ S,
                                                             const double *__restrict__ red3,
                                                             const double *__restrict__ invdiag,
                                                             double *__restrict__ u, const double *__restrict__ w,
                                                             double *__restrict__ p, double *__restrict__ s,
                                                             double *__restrict__ x, double *__restrict__ r,
                                                             double *__restrict__ part_g, double *__restrict__ part_rr,
                                                             const unsigned short *__restrict__ kind,
                                                             const double *__restrict__ ktab, int nk)
{
    __shared__ double red[kBlock / 64];
    __shared__ double ltab[kKindTabMax]; // (row kinds: invdiag[i] = ktab[kind[i]], see pcg_update_r_kernel)
    if (kind) {
        for (int t = threadIdx.x; t < nk; t += kBlock) ltab[t] = ktab[t];
        __syncthreads();
    }
    const double gamma = red3[0], rr = red3[1], delta = red3[2];
    const bool latched = S->done[parity] != 0;
    const bool bad = !isfinite(rr) || !isfinite(gamma) || !isfinite(delta);
    const bool conv = latched || bad || rr < S->threshold;
    if (conv || mode == 2) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (!latched) {
                S->rn2 = rr;
                if (bad) S->status = PSOLVE_HIP_NONFINITE_RESIDUAL;
                else if (conv)
                    S->status = (rr < S->abs2) ? PSOLVE_HIP_REACH_ABSOLUTE_TOLERANCE : PSOLVE_HIP_REACH_RELATIVE_TOLERANCE;
            }
            S->done[parity ^ 1] = conv ? 1 : 0;
        }
        return;
    }
    double beta = 0.0, alpha;
    if (mode == 1) {
        alpha = gamma / delta;
    } else {
        beta = gamma / S->rz[parity];
        alpha = gamma / (delta - beta * gamma / S->alpha[parity]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        S->rz[parity ^ 1] = gamma;
        S->alpha[parity ^ 1] = alpha;
        S->passes = S->passes + 1;
        S->rn2 = rr;
        S->done[parity ^ 1] = 0;
    }
    double sg = 0.0, srr = 0.0;
    // two elements (16 bytes) per lane and access where the vectors are 16-byte aligned (the solver's own buffers are)
    const bool al = ((((uintptr_t)u | (uintptr_t)w | (uintptr_t)p | (uintptr_t)s | (uintptr_t)x | (uintptr_t)r |
                       (uintptr_t)(invdiag ? invdiag : u)) & 15) == 0);
    const int n2 = al ? (n >> 1) : 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n2; i += gridDim.x * kBlock) {
        const v2d vu = ((const v2d *)u)[i], vw = ((const v2d *)w)[i];
        v2d vp = vu, vs = vw;
        if (mode != 1) {
            const v2d op = ((const v2d *)p)[i], os = ((const v2d *)s)[i];
            vp.x = vu.x + beta * op.x;
            vp.y = vu.y + beta * op.y;
            vs.x = vw.x + beta * os.x;
            vs.y = vw.y + beta * os.y;
        }
        ((v2d *)p)[i] = vp;
        ((v2d *)s)[i] = vs;
        v2d vx = ((const v2d *)x)[i], vr = ((const v2d *)r)[i];
        vx.x += alpha * vp.x;
        vx.y += alpha * vp.y;
        ((v2d *)x)[i] = vx;
        vr.x = vr.x - alpha * vs.x;
        vr.y = vr.y - alpha * vs.y;
        ((v2d *)r)[i] = vr;
        v2d un = vr;
        if (kind) {
            const unsigned kk = *reinterpret_cast<const unsigned *>(kind + 2 * (size_t)i);
            un.x = ltab[kk & 0xffffu] * vr.x;
            un.y = ltab[kk >> 16] * vr.y;
        } else if (invdiag) {
            const v2d d = ((const v2d *)invdiag)[i];
            un.x = d.x * vr.x;
            un.y = d.y * vr.y;
        }
        ((v2d *)u)[i] = un;
        sg += vr.x * un.x;
        sg += vr.y * un.y;
        srr += vr.x * vr.x;
        srr += vr.y * vr.y;
    }
    for (int i = 2 * n2 + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double ui = u[i], wi = w[i];
        const double pi = (mode == 1) ? ui : ui + beta * p[i];
        const double si = (mode == 1) ? wi : wi + beta * s[i];
        p[i] = pi;
        s[i] = si;
        x[i] += alpha * pi;
        const double ri = r[i] - alpha * si;
        r[i] = ri;
        const double un = invdiag ? (kind ? ltab[kind[i]] : invdiag[i]) * ri : ri;
        u[i] = un;
        sg += ri * un;
        srr += ri * ri;
    }
    const double tg = block_sum(sg, red);
    const double trr = block_sum(srr, red);
    if (threadIdx.x == 0) {
        part_g[blockIdx.x] = tg;
        part_rr[blockIdx.x] = trr;
    }
}

void launch_cg1_update(const Launch &L, int n, int parity, int mode, PcgState *S, const double *red3,
                       const double *invdiag, double *u, const double *w, double *p, double *s, double *x, double *r,
                       double *part_g, double *part_rr)
{
    const unsigned short *kk = (invdiag && invdiag == L.kd_for && L.kd_tab && L.kd_n > 0 && L.kd_n <= kKindTabMax) ? L.kd_kind : nullptr;
    hipLaunchKernelGGL(cg1_update_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, parity, mode, S, red3, invdiag, u,
                       w, p, s, x, r, part_g, part_rr, kk, kk ? L.kd_tab : nullptr, kk ? L.kd_n : 0);
    PS_HIP_CHECK(hipGetLastError());
}

// out3 = (sum part_g, sum part_rr, sum part_d): the three local sums of one iteration, folded by one workgroup
__global__ __launch_bounds__(kBlock) void cg1_fold_kernel(const double *__restrict__ part_g,
                                                           const double *__restrict__ part_rr, int np,
                                                           const double *__restrict__ part_d, int np_d,
                                                           double *__restrict__ out3)
{
    __shared__ double red[kBlock / 64];
    const double g = fold_partials(part_g, np, red);
    const double rr = fold_partials(part_rr, np, red);
    const double d = fold_partials(part_d, np_d, red);
    if (threadIdx.x == 0) {
        out3[0] = g;
        out3[1] = rr;
        out3[2] = d;
    }
}

void launch_cg1_fold(const Launch &L, const double *part_g, const double *part_rr, int np, const double *part_d,
                     int np_d, double *out3)
{
    hipLaunchKernelGGL(cg1_fold_kernel, dim3(1), dim3(kBlock), 0, L.stream, part_g, part_rr, np, part_d, np_d, out3);
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Synthetic inputs
// ---------------------------------------------------------------------------------------------
// number of stored entries in rows [0, row) of the nx*ny*nz 7-point matrix (closed form)
__host__ __device__ inline int64_t poisson7_before(int nx, int ny, int nz, int64_t row)
{
    const int64_t plane = (int64_t)nx * ny;
    const int64_t k = row / plane, rem = row - k * plane;
    int64_t missing = 0;
    missing += row < plane ? row : plane;                                              // k-1 neighbour absent
    missing += row > (int64_t)(nz - 1) * plane ? row - (int64_t)(nz - 1) * plane : 0;  // k+1
    missing += k * nx + (rem < nx ? rem : nx);                                         // j-1
    missing += k * nx + (rem > (int64_t)(ny - 1) * nx ? rem - (int64_t)(ny - 1) * nx : 0); // j+1
    missing += (row + nx - 1) / nx;                                                    // i-1
    missing += row / nx;                                                               // i+1
    return 7 * row - missing;
}

int64_t poisson7_nnz_before(int nx, int ny, int nz, int64_t row) { return poisson7_before(nx, ny, nz, row); }

__global__ __launch_bounds__(kBlock) void poisson7_kernel(int nx, int ny, int nz, int z0, int z1, int *rowptr,
                                                           int *col, double *val)
{
    const int64_t plane = (int64_t)nx * ny;
    const int64_t row_begin = (int64_t)z0 * plane, nloc = (int64_t)(z1 - z0) * plane;
    const int64_t base = poisson7_before(nx, ny, nz, row_begin);
    for (int64_t lr = (int64_t)blockIdx.x * kBlock + threadIdx.x; lr <= nloc; lr += (int64_t)gridDim.x * kBlock) {
        const int64_t r = row_begin + lr;
        int64_t p = poisson7_before(nx, ny, nz, r) - base;
        rowptr[lr] = (int)p;
        if (lr == nloc) break;
        const int64_t k = r / plane, rem = r - k * plane;
        const int j = (int)(rem / nx), i = (int)(rem - (int64_t)j * nx);
        if (k > 0) { col[p] = (int)(r - plane); val[p++] = -1.0; }
        if (j > 0) { col[p] = (int)(r - nx); val[p++] = -1.0; }
        if (i > 0) { col[p] = (int)(r - 1); val[p++] = -1.0; }
        col[p] = (int)r; val[p++] = 6.0;
        if (i < nx - 1) { col[p] = (int)(r + 1); val[p++] = -1.0; }
        if (j < ny - 1) { col[p] = (int)(r + nx); val[p++] = -1.0; }
        if (k < nz - 1) { col[p] = (int)(r + plane); val[p++] = -1.0; }
    }
}

void launch_poisson7_generate(const Launch &L, int nx, int ny, int nz, int z0, int z1, int *rowptr, int *col,
                              double *val)
{
    hipLaunchKernelGGL(poisson7_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nx, ny, nz, z0, z1, rowptr, col, val);
    PS_HIP_CHECK(hipGetLastError());
}

__device__ __forceinline__ double splitmix_unit(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

__global__ __launch_bounds__(kBlock) void splitmix_kernel(int n, uint64_t seed, int64_t start, double *x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        x[i] = splitmix_unit(seed + (uint64_t)(start + i));
}

void launch_splitmix(const Launch &L, int n, uint64_t seed, int64_t start, double *x)
{
    hipLaunchKernelGGL(splitmix_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, seed, start, x);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void splitmix_indexed_kernel(int n, uint64_t seed, const int *idx, double *x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        x[i] = splitmix_unit(seed + (uint64_t)idx[i]);
}

void launch_splitmix_indexed(const Launch &L, int n, uint64_t seed, const int *idx, double *x)
{
    hipLaunchKernelGGL(splitmix_indexed_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, seed, idx, x);
    PS_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Distributed helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_kernel(int n, const int *__restrict__ idx,
                                                         const double *__restrict__ x, double *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = x[idx[i]];
}

void launch_gather(const Launch &L, int n, const int *idx, const double *x, double *out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, idx, x, out);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void offrange_count_kernel(int64_t nnz, const int *__restrict__ col, int row0,
                                                                 int row1, int *count)
{
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) {
        const int v = col[i];
        c += (v < row0 || v >= row1);
    }
    if (c) atomicAdd(count, c);
}

void launch_offrange_count(const Launch &L, int64_t nnz, const int *col, int row0, int row1, int *count)
{
    hipLaunchKernelGGL(offrange_count_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nnz, col, row0, row1, count);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void offrange_collect_kernel(int64_t nnz, const int *__restrict__ col, int row0,
                                                                   int row1, int *out, int *cursor)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) {
        const int v = col[i];
        if (v < row0 || v >= row1) out[atomicAdd(cursor, 1)] = v;
    }
}

void launch_offrange_collect(const Launch &L, int64_t nnz, const int *col, int row0, int row1, int *out, int *cursor)
{
    hipLaunchKernelGGL(offrange_collect_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nnz, col, row0, row1, out,
                       cursor);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void classify_row_blocks_kernel(int n, int R, int n_local,
                                                                      const int *__restrict__ rowptr,
                                                                      const int *__restrict__ col, int *flags)
{
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        bool halo = false;
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) halo = halo || (col[j] >= n_local);
        if (halo) flags[r / R] = 1; // benign race: every writer stores the same value
    }
}

void launch_classify_row_blocks(const Launch &L, const CsrDev &A, int *flags)
{
    hipLaunchKernelGGL(classify_row_blocks_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, A.n, A.rows_per_block,
                       A.n, A.rowptr, A.col, flags);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void remap_cols_kernel(int64_t nnz, int *__restrict__ col, int row0, int row1,
                                                             int n_local, const int *__restrict__ halo, int n_halo)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) {
        const int v = col[i];
        if (v >= row0 && v < row1) {
            col[i] = v - row0;
        } else {
            int lo = 0, hi = n_halo;
            while (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1); // lo + hi can pass 2^31
                if (halo[mid] < v) lo = mid + 1; else hi = mid;
            }
            col[i] = n_local + lo;
        }
    }
}

void launch_remap_cols(const Launch &L, int64_t nnz, int *col, int row0, int row1, int n_local, const int *halo,
                       int n_halo)
{
    hipLaunchKernelGGL(remap_cols_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nnz, col, row0, row1, n_local,
                       halo, n_halo);
    PS_HIP_CHECK(hipGetLastError());
}

// the inverse of remap_cols: local column ids of a shard back to global ids (out may not alias col)
__global__ __launch_bounds__(kBlock) void unmap_cols_kernel(int64_t nnz, const int *__restrict__ col, int row0,
                                                             int n_local, const int *__restrict__ halo,
                                                             int *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) {
        const int v = col[i];
        out[i] = v < n_local ? v + row0 : halo[v - n_local];
    }
}

void launch_unmap_cols(const Launch &L, int64_t nnz, const int *col, int row0, int n_local, const int *halo, int *out)
{
    hipLaunchKernelGGL(unmap_cols_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, nnz, col, row0, n_local, halo, out);
    PS_HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(kBlock) void add_offset_i32_kernel(int64_t n, int *__restrict__ v, int offset)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) v[i] += offset;
}

void launch_add_offset_i32(const Launch &L, int64_t n, int *v, int offset)
{
    if (n <= 0 || offset == 0) return;
    hipLaunchKernelGGL(add_offset_i32_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, n, v, offset);
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
