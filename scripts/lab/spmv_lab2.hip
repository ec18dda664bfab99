// spmv_lab2.hip -- round-2 laboratory for the CSR SpMV on MI355X (not part of the product).
//   A. rw_mix:     what the memory system gives a streaming read of R bytes with W bytes of stores mixed in
//                  (the SpMV's traffic shape: 92 % reads / 8 % writes), plain vs nt loads, 8- vs 16-B stores
//   B. spmv_dma:   the matrix stream goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB contiguous per
//                  wave instruction => whole 128-B lines per instruction, so `nt` cannot miss L1 twice), no staging
//                  VGPRs; thread t then walks ITS row out of LDS, gathers x, sums in column order (bit-exact with
//                  the scalar loop), writes y.  Single-buffered: overlap comes from 6-8 resident workgroups per CU.
//   C. the same with the row pointers dropped for row-blocks of uniform degree.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off spmv_lab2.hip -o spmv_lab2
// Run:   ./spmv_lab2 [N=256] [reps=20]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t p7_before(int nx, int ny, int nz, int64_t row)
{
    const int64_t plane = (int64_t)nx * ny;
    const int64_t k = row / plane, rem = row - k * plane;
    int64_t m = 0;
    m += row < plane ? row : plane;
    m += row > (int64_t)(nz - 1) * plane ? row - (int64_t)(nz - 1) * plane : 0;
    m += k * nx + (rem < nx ? rem : nx);
    m += k * nx + (rem > (int64_t)(ny - 1) * nx ? rem - (int64_t)(ny - 1) * nx : 0);
    m += (row + nx - 1) / nx;
    m += row / nx;
    return 7 * row - m;
}

__global__ void gen(int nx, int ny, int nz, int *rowptr, int *col, double *val)
{
    const int64_t plane = (int64_t)nx * ny, n = plane * nz;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = p7_before(nx, ny, nz, r);
        rowptr[r] = (int)p;
        if (r == n) break;
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

__global__ void fillx(int n, double *x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t z = 42 + (uint64_t)i;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

__global__ void ref_spmv(int n, const int *rowptr, const int *col, const double *val, const double *x, double *y)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        double s = 0;
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) s += val[k] * x[col[k]];
        y[r] = s;
    }
}

// ------------------------------------------------------------------ A. read/write mix model
// every workgroup walks "steps"; per step it reads RD16 x 256 x 16 B (coalesced, contiguous per step) and stores
// WR8 x 256 x 8 B (contiguous per step) -- the SpMV's shape is RD16 = 5.25 (21.5 KB) and WR8 = 1 (2 KB) per 256 rows
template <int RD16, int WR8, bool NT, bool NTS = false>
__global__ __launch_bounds__(256) void rw_mix(const v4f *__restrict__ a, double *__restrict__ b, float *out, int64_t nsteps)
{
    v4f s = {0, 0, 0, 0};
    for (int64_t st = blockIdx.x; st < nsteps; st += gridDim.x) {
        const v4f *src = a + st * (int64_t)(RD16 * 256);
#pragma unroll
        for (int k = 0; k < RD16; ++k) {
            if (NT) s += __builtin_nontemporal_load(src + k * 256 + threadIdx.x);
            else s += src[k * 256 + threadIdx.x];
        }
#pragma unroll
        for (int k = 0; k < WR8; ++k) {
            double *dst = b + st * (int64_t)(WR8 * 256) + k * 256 + threadIdx.x;
            if (NTS) __builtin_nontemporal_store((double)s.x, dst);
            else *dst = (double)s.x;
        }
    }
    float t = s.x + s.y + s.z + s.w;
    if (t == 1.2345f) out[0] = t;
}

// ------------------------------------------------------------------ B. LDS-DMA staged, thread-per-row
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base, bool nt)
{
    // 16 B per lane: lane l's bytes land at lds_wave_base + 16 l (wave-uniform base + lane * size)
    if (nt) __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// R rows per workgroup step (R = 256: one thread per row; R = 128: threads 128..255 only help with the DMA),
// TILE = LDS capacity in nonzeros (multiple of 256), UNI = use the per-block uniform-degree table instead of rowptr
template <int R, int TILE, bool NT, bool UNI, int MAP>
__global__ __launch_bounds__(256) void spmv_dma(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                 const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ x, double *__restrict__ y,
                                                 const int *__restrict__ blk_lo, const signed char *__restrict__ blk_deg,
                                                 int nrb, int chunk)
{
    __shared__ __attribute__((aligned(16))) int lcol[TILE];
    __shared__ __attribute__((aligned(16))) double lval[TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = MAP ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb;
    for (int l = MAP ? slot : (int)blockIdx.x; l < nloop; l += MAP ? slots : (int)gridDim.x) {
        const int rb = MAP ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : l;
        if (rb >= nrb) continue;
        const int row0 = rb * R;
        int lo, hi, deg = -1;
        if (UNI) {
            lo = blk_lo[rb];
            hi = blk_lo[rb + 1];
            deg = blk_deg[rb];
        } else {
            lo = rowptr[row0];
            hi = rowptr[min(row0 + R, n)];
        }
        const int c0 = lo & ~3;
        const int cnt = hi - c0; // entries to stage (<= TILE assumed in the lab)
        // col: 4 entries per lane, 256 per wave instruction; val: 2 per lane, 128 per wave instruction
#pragma unroll
        for (int k = 0; k < TILE / 1024; ++k) {
            const int e = (k * 4 + wave) * 256; // first entry of this wave's instruction
            if (e < cnt) dma16(col + c0 + e + lane * 4, lcol + e, NT);
        }
#pragma unroll
        for (int k = 0; k < TILE / 512; ++k) {
            const int e = (k * 4 + wave) * 128;
            if (e < cnt) dma16(val + c0 + e + lane * 2, lval + e, NT);
        }
        int rs = 0, re = 0;
        const int r = row0 + tid;
        if (tid < R && r < n) {
            if (UNI && deg >= 0) {
                rs = lo + tid * deg;
                re = rs + deg;
            } else {
                rs = rowptr[r];
                re = rowptr[r + 1];
            }
        }
        __syncthreads(); // hipcc drains vmcnt(0) (the DMA) before the barrier
        if (tid < R && r < n) {
            double acc = 0.0;
            for (int k = rs; k < re; ++k) acc += lval[k - c0] * x[lcol[k - c0]];
            y[r] = acc;
        }
        __syncthreads(); // tile reuse
    }
}


// ------------------------------------------------------------------ D. the product's pipeline, line-exact loads
// Same structure as the product's spmv_csr_pipe<256, PLAIN> (A: gather + products of the prefetched stream -> LDS
// tile | barrier | B: issue the next block's stream loads | C: every row sums its slice), but every load
// instruction of a wave covers WHOLE 128-byte lines: within a wave's 256-entry segment lane l owns the entry
// pairs {2l, 2l+1} and {128+2l, 129+2l}; values by two 16-B loads (1 KiB contiguous per instruction), columns by
// two 8-B loads (512 B contiguous).  With that, `nt` on the matrix stream cannot miss L1 twice on one line.
// DEPTH 2 keeps two blocks of stream in registers (loads issued two blocks ahead).
constexpr int kTile2 = 1856;
typedef int v2i __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ v2i ld_c2(const int *p)
{
    return NT ? __builtin_nontemporal_load((const v2i *)p) : *(const v2i *)p;
}
template <bool NT>
__device__ __forceinline__ v2d ld_v2(const double *p)
{
    return NT ? __builtin_nontemporal_load((const v2d *)p) : *(const v2d *)p;
}

struct Stream2 { // one block's stream: 2 rounds x (2 col pairs + 2 val pairs) = 24 VGPRs
    v2i c[2][2];
    v2d v[2][2];
};
struct Ptr2 { int rs, re, lo, hi; };

template <bool NT, int DEPTH, int MAP>
__global__ __launch_bounds__(256) void spmv_pipe2(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                   const int *__restrict__ col, const double *__restrict__ val,
                                                   const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                   int chunk)
{
    __shared__ double prod[2][kTile2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = MAP ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb;
    const int step = MAP ? slots : (int)gridDim.x;
    auto rb_of = [&](int l) { return MAP ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : l; };
    auto valid = [&](int l) { return l < nloop && rb_of(l) < nrb; };
    // entry offsets of this lane inside the tile, for round k and half h: k*1024 + wave*256 + h*128 + 2*lane
    auto load = [&](Stream2 &S, int lo, int hi) {
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = k * 1024 + wave * 256 + h * 128 + 2 * lane;
                S.c[k][h] = (v2i){0, 0};
                S.v[k][h] = (v2d){0.0, 0.0};
                if (c0 + o < hi && o < kTile2) { // (lab: arrays are padded, no tail case)
                    S.c[k][h] = ld_c2<NT>(col + c0 + o);
                    S.v[k][h] = ld_v2<NT>(val + c0 + o);
                }
            }
    };
    auto products = [&](const Stream2 &S, double *P, int lo, int hi) {
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int o = k * 1024 + wave * 256 + h * 128 + 2 * lane;
                if (c0 + o < hi && o < kTile2) {
                    v2d p;
                    p.x = S.v[k][h].x * x[S.c[k][h].x];
                    p.y = S.v[k][h].y * x[S.c[k][h].y];
                    *(v2d *)(P + o) = p;
                }
            }
    };
    auto row_sum = [&](const double *P, int a, int e) {
        double acc = 0.0;
        int j = a;
        for (; j + 8 <= e; j += 8) {
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += t[q];
        }
        for (; j < e; ++j) acc += P[j];
        return acc;
    };
    auto load_ptr = [&](int rb) {
        Ptr2 p{0, 0, 0, 0};
        const int row0 = rb * 256, r = row0 + tid;
        if (r < n) {
            p.rs = rowptr[r];
            p.re = rowptr[r + 1];
        }
        p.lo = rowptr[row0];
        p.hi = rowptr[min(row0 + 256, n)];
        return p;
    };
    int l = MAP ? slot : (int)blockIdx.x;
    Stream2 S0, S1;
    Ptr2 p0{0, 0, 0, 0}, p1{0, 0, 0, 0};
    if (valid(l)) {
        p0 = load_ptr(rb_of(l));
        load(S0, p0.lo, p0.hi);
    }
    if (DEPTH == 2 && valid(l + step)) {
        p1 = load_ptr(rb_of(l + step));
        load(S1, p1.lo, p1.hi);
    }
    int buf = 0;
    // one pipeline stage: consume (S, p) of block l, refill S with block l + DEPTH*step
#define STAGE(S, p)                                                                       \
    {                                                                                     \
        const int rb = rb_of(l);                                                          \
        const int c0 = p.lo & ~3;                                                         \
        double *P = prod[buf];                                                            \
        products(S, P, p.lo, p.hi);                                                       \
        __syncthreads();                                                                  \
        const Ptr2 cur = p;                                                               \
        const int ln = l + DEPTH * step;                                                  \
        if (valid(ln)) {                                                                  \
            p = load_ptr(rb_of(ln));                                                      \
            load(S, p.lo, p.hi);                                                          \
        }                                                                                 \
        const double acc = row_sum(P, max(cur.rs, c0) - c0, min(cur.re, c0 + kTile2) - c0); \
        const int r = rb * 256 + tid;                                                     \
        if (r < n) y[r] = acc;                                                            \
        buf ^= 1;                                                                         \
        l += step;                                                                        \
    }
    if (DEPTH == 1) {
        while (valid(l)) STAGE(S0, p0)
    } else {
        while (valid(l)) {
            STAGE(S0, p0)
            if (!valid(l)) break;
            STAGE(S1, p1)
        }
    }
#undef STAGE
}


// ------------------------------------------------------------------ E. stream loads issued BEFORE the gathers
// pipe3: the next block's stream loads are issued first, then the current block's gathers, so the two memory
// latencies of an iteration overlap (one exposed latency per row-block instead of two).  Row pointers are loaded
// two blocks ahead, so no load address depends on a load of the same iteration.
// LINE = 0: the product's mapping (16-B column quad + two 16-B value pairs per lane, lane stride 4 entries);
// LINE = 1: line-exact mapping of pipe2 (whole 128-B lines per wave instruction).
struct Stream3 {
    v4i c4[2];
    v2d va[2], vb[2];
};

template <bool NT, int LINE, int MAP>
__global__ __launch_bounds__(256) void spmv_pipe3(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                   const int *__restrict__ col, const double *__restrict__ val,
                                                   const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                   int chunk)
{
    __shared__ double prod[2][kTile2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = MAP ? (((nrb + chunk - 1) / chunk + 7) / 8) * chunk : nrb;
    const int step = MAP ? slots : (int)gridDim.x;
    auto rb_of = [&](int l) { return MAP ? ((l / chunk) * 8 + xcd) * chunk + (l % chunk) : l; };
    auto valid = [&](int l) { return l < nloop && rb_of(l) < nrb; };
    auto off = [&](int k, int h) { // entry offset of this lane's (k, h) group inside the tile
        return LINE ? k * 1024 + wave * 256 + h * 128 + 2 * lane : k * 1024 + tid * 4 + 2 * h;
    };
    auto load = [&](Stream3 &S, int lo, int hi) {
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            S.c4[k] = (v4i){0, 0, 0, 0};
            S.va[k] = (v2d){0.0, 0.0};
            S.vb[k] = (v2d){0.0, 0.0};
            if (LINE) {
                const int o0 = off(k, 0), o1 = off(k, 1);
                if (c0 + o0 < hi && o0 < kTile2) {
                    const v2i c = ld_c2<NT>(col + c0 + o0);
                    S.c4[k].x = c.x;
                    S.c4[k].y = c.y;
                    S.va[k] = ld_v2<NT>(val + c0 + o0);
                }
                if (c0 + o1 < hi && o1 < kTile2) {
                    const v2i c = ld_c2<NT>(col + c0 + o1);
                    S.c4[k].z = c.x;
                    S.c4[k].w = c.y;
                    S.vb[k] = ld_v2<NT>(val + c0 + o1);
                }
            } else {
                const int o = off(k, 0);
                if (c0 + o < hi && o < kTile2) {
                    S.c4[k] = NT ? __builtin_nontemporal_load((const v4i *)(col + c0 + o)) : *(const v4i *)(col + c0 + o);
                    S.va[k] = ld_v2<NT>(val + c0 + o);
                    S.vb[k] = ld_v2<NT>(val + c0 + o + 2);
                }
            }
        }
    };
    auto products = [&](const Stream3 &S, double *P, int lo, int hi) {
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int o0 = off(k, 0), o1 = off(k, 1);
            if (c0 + o0 < hi && o0 < kTile2) {
                v2d p;
                p.x = S.va[k].x * x[S.c4[k].x];
                p.y = S.va[k].y * x[S.c4[k].y];
                *(v2d *)(P + o0) = p;
            }
            if (c0 + o1 < hi && o1 < kTile2) {
                v2d p;
                p.x = S.vb[k].x * x[S.c4[k].z];
                p.y = S.vb[k].y * x[S.c4[k].w];
                *(v2d *)(P + o1) = p;
            }
        }
    };
    auto row_sum = [&](const double *P, int a, int e) {
        double acc = 0.0;
        int j = a;
        for (; j + 8 <= e; j += 8) {
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += t[q];
        }
        for (; j < e; ++j) acc += P[j];
        return acc;
    };
    auto load_ptr = [&](int l) {
        Ptr2 p{0, 0, 0, 0};
        if (!valid(l)) return p;
        const int row0 = rb_of(l) * 256, r = row0 + tid;
        if (r < n) {
            p.rs = rowptr[r];
            p.re = rowptr[r + 1];
        }
        p.lo = rowptr[row0];
        p.hi = rowptr[min(row0 + 256, n)];
        return p;
    };
    int l = MAP ? slot : (int)blockIdx.x;
    Ptr2 pc = load_ptr(l), pn = load_ptr(l + step);
    Stream3 Sc, Sn;
    if (valid(l)) load(Sc, pc.lo, pc.hi);
    int buf = 0;
    while (valid(l)) {
        const bool has_next = valid(l + step);
        if (has_next) load(Sn, pn.lo, pn.hi);       // next block's stream: on the wire while we gather
        const Ptr2 pnn = load_ptr(l + 2 * step);    // pointers two blocks ahead
        double *P = prod[buf];
        products(Sc, P, pc.lo, pc.hi);              // gathers + products of the current block
        __syncthreads();
        const int c0 = pc.lo & ~3;
        const double acc = row_sum(P, max(pc.rs, c0) - c0, min(pc.re, c0 + kTile2) - c0);
        const int r = rb_of(l) * 256 + tid;
        if (r < n) y[r] = acc;
        Sc = Sn;
        pc = pn;
        pn = pnn;
        buf ^= 1;
        l += step;
    }
}


// ------------------------------------------------------------------ F. where does the nt advantage go?
// spmv_dma with switches: GATHER 0 = no x loads (acc += val * col), 1 = real gathers, 2 = gathers folded onto a
// 8 KiB window of x (same instructions, no x traffic beyond L1); YST 0 plain store, 1 nt store, 2 no store
template <bool NT, int GATHER, int YST, int TILE = 2048, bool DOT = false>
__global__ __launch_bounds__(256) void spmv_dma_probe(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                       int chunk)
{
    constexpr int R = 256;
    __shared__ __attribute__((aligned(16))) int lcol[TILE];
    __shared__ __attribute__((aligned(16))) double lval[TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = (((nrb + chunk - 1) / chunk + 7) / 8) * chunk;
    double keep = 0.0;
    for (int l = slot; l < nloop; l += slots) {
        const int rb = ((l / chunk) * 8 + xcd) * chunk + (l % chunk);
        if (rb >= nrb) continue;
        const int row0 = rb * R;
        const int lo = rowptr[row0], hi = rowptr[min(row0 + R, n)];
        const int c0 = lo & ~3;
        const int cnt = hi - c0;
#pragma unroll
        for (int k = 0; k < TILE / 1024; ++k) {
            const int e = (k * 4 + wave) * 256;
            if (e < cnt) dma16(col + c0 + e + lane * 4, lcol + e, NT);
        }
#pragma unroll
        for (int k = 0; k < TILE / 512; ++k) {
            const int e = (k * 4 + wave) * 128;
            if (e < cnt) dma16(val + c0 + e + lane * 2, lval + e, NT);
        }
        int rs = 0, re = 0;
        const int r = row0 + tid;
        if (r < n) {
            rs = rowptr[r];
            re = rowptr[r + 1];
        }
        __syncthreads();
        if (r < n) {
            double acc = 0.0;
            for (int k = rs; k < re; ++k) {
                const int c = lcol[k - c0];
                if (GATHER == 0) acc += lval[k - c0] * (double)c;
                else if (GATHER == 1) acc += lval[k - c0] * x[c];
                else acc += lval[k - c0] * x[c & 1023];
            }
            if (DOT) keep += acc * x[r];
            if (YST == 0) y[r] = acc;
            else if (YST == 1) __builtin_nontemporal_store(acc, y + r);
            else keep += acc;
        }
        __syncthreads();
    }
    if ((YST == 2 || DOT) && keep == 1.2345) y[0] = keep;
}


// ------------------------------------------------------------------ G. inner-loop / pointer variants of the DMA kernel
// VAR 0: plain loop (one gather in flight per row at a time); 1: next block's row pointers prefetched before the
// barrier; 2: #pragma unroll 2; 3: #pragma unroll 4; 4: entries loaded 4 at a time from LDS, gathers batched,
// remainder by the plain loop; 5 = 1 + 4
template <int VAR, bool DOT>
__global__ __launch_bounds__(256) void spmv_dma_var(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                     const int *__restrict__ col, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                     int chunk)
{
    constexpr int TILE = 2048, R = 256;
    __shared__ __attribute__((aligned(16))) int lcol[TILE];
    __shared__ __attribute__((aligned(16))) double lval[TILE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = (((nrb + chunk - 1) / chunk + 7) / 8) * chunk;
    double keep = 0.0;
    auto rb_of = [&](int l) { return ((l / chunk) * 8 + xcd) * chunk + (l % chunk); };
    auto ptrs = [&](int l, int &lo, int &hi, int &rs, int &re) {
        lo = hi = rs = re = 0;
        if (l >= nloop) return;
        const int rb = rb_of(l);
        if (rb >= nrb) return;
        const int row0 = rb * R, r = row0 + tid;
        lo = rowptr[row0];
        hi = rowptr[min(row0 + R, n)];
        if (r < n) {
            rs = rowptr[r];
            re = rowptr[r + 1];
        }
    };
    int lo, hi, rs, re;
    constexpr bool PF = VAR == 1 || VAR == 5;
    if (PF) ptrs(slot, lo, hi, rs, re);
    for (int l = slot; l < nloop; l += slots) {
        const int rb = rb_of(l);
        if (!PF) ptrs(l, lo, hi, rs, re);
        int lo_n = 0, hi_n = 0, rs_n = 0, re_n = 0;
        if (rb < nrb) {
            const int row0 = rb * R;
            const int c0 = lo & ~3;
            const int cnt = hi - c0;
#pragma unroll
            for (int k = 0; k < TILE / 1024; ++k) {
                const int e = (k * 4 + wave) * 256;
                if (e < cnt) dma16(col + c0 + e + lane * 4, lcol + e, true);
            }
#pragma unroll
            for (int k = 0; k < TILE / 512; ++k) {
                const int e = (k * 4 + wave) * 128;
                if (e < cnt) dma16(val + c0 + e + lane * 2, lval + e, true);
            }
            if (PF) ptrs(l + slots, lo_n, hi_n, rs_n, re_n);
            const int r = row0 + tid;
            __syncthreads();
            if (r < n) {
                double acc = 0.0;
                const int a = rs - c0, e = re - c0;
                if (VAR == 2) {
#pragma unroll 2
                    for (int k = a; k < e; ++k) acc += lval[k] * x[lcol[k]];
                } else if (VAR == 3) {
#pragma unroll 4
                    for (int k = a; k < e; ++k) acc += lval[k] * x[lcol[k]];
                } else if (VAR == 4 || VAR == 5) {
                    int k = a;
                    for (; k + 4 <= e; k += 4) {
                        const int c0_ = lcol[k], c1_ = lcol[k + 1], c2_ = lcol[k + 2], c3_ = lcol[k + 3];
                        const double v0 = lval[k], v1 = lval[k + 1], v2 = lval[k + 2], v3 = lval[k + 3];
                        const double x0 = x[c0_], x1 = x[c1_], x2 = x[c2_], x3 = x[c3_];
                        acc += v0 * x0;
                        acc += v1 * x1;
                        acc += v2 * x2;
                        acc += v3 * x3;
                    }
                    for (; k < e; ++k) acc += lval[k] * x[lcol[k]];
                } else {
                    for (int k = a; k < e; ++k) acc += lval[k] * x[lcol[k]];
                }
                if (DOT) keep += acc * x[r];
                __builtin_nontemporal_store(acc, y + r);
            }
            __syncthreads();
        } else if (PF) {
            ptrs(l + slots, lo_n, hi_n, rs_n, re_n);
        }
        if (PF) {
            lo = lo_n;
            hi = hi_n;
            rs = rs_n;
            re = re_n;
        }
    }
    if (DOT && keep == 1.2345) y[0] = keep;
}

// ------------------------------------------------------------------ harness
struct Prob {
    int n;
    int64_t nnz;
    int *rowptr, *col;
    double *val, *x, *y, *yref;
};

template <typename F>
static double timeit(F &&launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

static bool check(const Prob &P, const char *name)
{
    std::vector<double> a(P.n), b(P.n);
    CK(hipMemcpy(a.data(), P.y, (size_t)P.n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), P.yref, (size_t)P.n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int i = 0; i < P.n; ++i) bad += a[i] != b[i];
    if (bad) printf("   !! %s: %zu rows differ from the reference\n", name, bad);
    CK(hipMemset(P.y, 0, (size_t)P.n * 8));
    return bad == 0;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d  N=%d\n", prop.name, cus, N);
    Prob P;
    P.n = N * N * N;
    P.nnz = p7_before(N, N, N, P.n);
    CK(hipMalloc(&P.rowptr, (size_t)(P.n + 1) * 4));
    CK(hipMalloc(&P.col, (size_t)(P.nnz + 4096) * 4));
    CK(hipMalloc(&P.val, (size_t)(P.nnz + 4096) * 8));
    CK(hipMalloc(&P.x, (size_t)P.n * 8));
    CK(hipMalloc(&P.y, (size_t)P.n * 8));
    CK(hipMalloc(&P.yref, (size_t)P.n * 8));
    gen<<<4096, 256>>>(N, N, N, P.rowptr, P.col, P.val);
    fillx<<<4096, 256>>>(P.n, P.x);
    ref_spmv<<<4096, 256>>>(P.n, P.rowptr, P.col, P.val, P.x, P.yref);
    CK(hipDeviceSynchronize());
    const double bytes = 12.0 * P.nnz + 20.0 * P.n;
    auto report = [&](const char *name, double ms) {
        printf("%-52s %8.4f ms  %7.1f GB/s (alg)  %5.1f%% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        fflush(stdout);
    };

    const bool quick = argc > 3;
    // ---- A: read/write mix (1.61 GB read in steps of 21 KB, W per step varied) ----
    if (!quick) {
        const int64_t nsteps = 65536; // 65536 x 21.5 KB = 1.41 GB read (RD16 = 5.25 is not integral: use 5 and 6)
        v4f *a;
        double *b;
        float *o;
        CK(hipMalloc(&a, (size_t)nsteps * 6 * 256 * 16));
        CK(hipMalloc(&b, (size_t)nsteps * 4 * 256 * 8));
        CK(hipMalloc(&o, 64));
        CK(hipMemset(a, 0, (size_t)nsteps * 6 * 256 * 16));
#define RW(RD, WR, NT, G)                                                                                       \
    {                                                                                                           \
        double ms = timeit([&] { rw_mix<RD, WR, NT><<<G, 256>>>(a, b, o, nsteps); }, reps);                       \
        const double rb_ = (double)nsteps * RD * 4096, wb_ = (double)nsteps * WR * 2048;                          \
        printf("rw_mix read %.2f GB + write %.3f GB (%4.1f%% writes) nt=%d grid=%5d: %.4f ms  %.0f GB/s total, read part at 6.6 TB/s -> writes cost %.4f ms\n", \
               rb_ / 1e9, wb_ / 1e9, 100 * wb_ / (rb_ + wb_), NT, G, ms, (rb_ + wb_) / ms / 1e6, ms - rb_ / 6.6e9);  \
        fflush(stdout);                                                                                         \
    }
        for (int g : {cus * 5, cus * 8}) {
            RW(5, 0, false, g) RW(5, 0, true, g) RW(5, 1, false, g) RW(5, 1, true, g) RW(5, 2, false, g) RW(5, 2, true, g)
            RW(5, 4, false, g) RW(5, 4, true, g) RW(6, 1, true, g)
        }
        CK(hipFree(a));
        CK(hipFree(b));
        CK(hipFree(o));
    }

    // ---- B / C: LDS-DMA kernels ----
    const int chunkrows = 8192;
    // uniform-degree table per row-block of R rows
    auto make_tables = [&](int R, int **d_lo, signed char **d_deg) {
        const int nrb = (P.n + R - 1) / R;
        std::vector<int> rp(P.n + 1);
        CK(hipMemcpy(rp.data(), P.rowptr, (size_t)(P.n + 1) * 4, hipMemcpyDeviceToHost));
        std::vector<int> lo(nrb + 1);
        std::vector<signed char> deg(nrb);
        int uni = 0;
        for (int b = 0; b < nrb; ++b) {
            const int r0 = b * R, r1 = std::min(P.n, r0 + R);
            lo[b] = rp[r0];
            int d = rp[r0 + 1] - rp[r0];
            for (int r = r0; r < r1; ++r)
                if (rp[r + 1] - rp[r] != d) d = -1;
            deg[b] = (signed char)d;
            uni += d >= 0;
        }
        lo[nrb] = rp[P.n];
        printf("R=%d: %d of %d row-blocks have uniform degree\n", R, uni, nrb);
        CK(hipMalloc(d_lo, (size_t)(nrb + 1) * 4));
        CK(hipMalloc(d_deg, (size_t)nrb));
        CK(hipMemcpy(*d_lo, lo.data(), (size_t)(nrb + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(*d_deg, deg.data(), (size_t)nrb, hipMemcpyHostToDevice));
        return nrb;
    };
    int *lo256, *lo128;
    signed char *deg256, *deg128;
    const int nrb256 = make_tables(256, &lo256, &deg256);
    const int nrb128 = make_tables(128, &lo128, &deg128);
#define DMA(R, TILE, NT, UNI, MAP, BPC)                                                                            \
    if (!quick) {                                                                                                              \
        const int nrb = R == 256 ? nrb256 : nrb128;                                                                \
        const int grid = (cus * BPC + 7) & ~7;                                                                     \
        char name[128];                                                                                            \
        snprintf(name, sizeof name, "dma R=%d TILE=%d nt=%d uni=%d map=%d wg/cu=%d", R, TILE, NT, UNI, MAP, BPC);    \
        double ms = timeit([&] {                                                                                   \
            spmv_dma<R, TILE, NT, UNI, MAP><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y,             \
                                                           R == 256 ? lo256 : lo128, R == 256 ? deg256 : deg128, nrb, \
                                                           chunkrows / R);                                         \
        }, reps);                                                                                                  \
        check(P, name);                                                                                            \
        report(name, ms);                                                                                          \
    }
    DMA(256, 2048, false, false, 2, 6)
    DMA(256, 2048, true, false, 2, 6)
    DMA(256, 2048, false, false, 0, 6)
    DMA(256, 2048, true, false, 0, 6)
    DMA(256, 2048, true, true, 2, 6)
    DMA(256, 2048, true, true, 0, 6)
    DMA(256, 2048, true, false, 2, 4)
    DMA(256, 2048, true, false, 2, 5)
    DMA(128, 1024, false, false, 2, 8)
    DMA(128, 1024, true, false, 2, 8)
    DMA(128, 1024, true, true, 2, 8)
    DMA(128, 1024, true, true, 0, 8)
    DMA(128, 1024, true, false, 2, 6)
#define PIPE2(NT, DEPTH, MAP, BPC)                                                                              \
    if (!quick) {                                                                                                           \
        const int grid = (cus * BPC + 7) & ~7;                                                                  \
        char name[128];                                                                                         \
        snprintf(name, sizeof name, "pipe2 nt=%d depth=%d map=%d wg/cu=%d", NT, DEPTH, MAP, BPC);                 \
        double ms = timeit([&] {                                                                                \
            spmv_pipe2<NT, DEPTH, MAP><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb256, 32);  \
        }, reps);                                                                                               \
        check(P, name);                                                                                         \
        report(name, ms);                                                                                       \
    }
    PIPE2(false, 1, 2, 5)
    PIPE2(true, 1, 2, 5)
    PIPE2(false, 2, 2, 5)
    PIPE2(true, 2, 2, 5)
    PIPE2(true, 1, 0, 5)
    PIPE2(true, 2, 0, 5)
    PIPE2(true, 1, 2, 4)
    PIPE2(true, 2, 2, 4)
    PIPE2(true, 2, 2, 3)
#define PIPE3(NT, LINE, MAP, BPC)                                                                               \
    if (!quick) {                                                                                                           \
        const int grid = (cus * BPC + 7) & ~7;                                                                  \
        char name[128];                                                                                         \
        snprintf(name, sizeof name, "pipe3 nt=%d line=%d map=%d wg/cu=%d", NT, LINE, MAP, BPC);                   \
        double ms = timeit([&] {                                                                                \
            spmv_pipe3<NT, LINE, MAP><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb256, 32);   \
        }, reps);                                                                                               \
        check(P, name);                                                                                         \
        report(name, ms);                                                                                       \
    }
    PIPE3(false, 0, 2, 5)
    PIPE3(true, 0, 2, 5)
    PIPE3(false, 1, 2, 5)
    PIPE3(true, 1, 2, 5)
    PIPE3(false, 0, 2, 4)
    PIPE3(true, 1, 2, 4)
    PIPE3(false, 0, 0, 5)
    PIPE3(true, 1, 0, 5)
#define PROBE(NT, GATHER, YST)                                                                                  \
    if (!quick) {                                                                                                           \
        const int grid = (cus * 6 + 7) & ~7;                                                                    \
        char name[128];                                                                                         \
        snprintf(name, sizeof name, "dma-probe nt=%d gather=%d ystore=%d", NT, GATHER, YST);                      \
        double ms = timeit([&] {                                                                                \
            spmv_dma_probe<NT, GATHER, YST><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb256, 32); \
        }, reps);                                                                                               \
        report(name, ms);                                                                                       \
    }
    PROBE(false, 0, 2) PROBE(true, 0, 2)
    PROBE(false, 0, 0) PROBE(true, 0, 0) PROBE(true, 0, 1)
    PROBE(false, 2, 0) PROBE(true, 2, 0)
    PROBE(false, 1, 2) PROBE(true, 1, 2)
    PROBE(false, 1, 0) PROBE(true, 1, 0) PROBE(true, 1, 1) PROBE(false, 1, 1)
    {
        const int64_t nsteps = 65536;
        v4f *a;
        double *b;
        float *o;
        CK(hipMalloc(&a, (size_t)nsteps * 6 * 256 * 16));
        CK(hipMalloc(&b, (size_t)nsteps * 4 * 256 * 8));
        CK(hipMalloc(&o, 64));
        CK(hipMemset(a, 0, (size_t)nsteps * 6 * 256 * 16));
#define RW2(RD, WR, NT, NTS, G)                                                                                 \
    if (!quick) {                                                                                                           \
        double ms = timeit([&] { rw_mix<RD, WR, NT, NTS><<<G, 256>>>(a, b, o, nsteps); }, reps);                  \
        const double rb_ = (double)nsteps * RD * 4096, wb_ = (double)nsteps * WR * 2048;                          \
        printf("rw_mix2 read %.2f GB + write %.3f GB (%4.1f%% writes) nt-load=%d nt-store=%d grid=%5d: %.4f ms  %.0f GB/s total\n", \
               rb_ / 1e9, wb_ / 1e9, 100 * wb_ / (rb_ + wb_), NT, NTS, G, ms, (rb_ + wb_) / ms / 1e6);           \
        fflush(stdout);                                                                                         \
    }
        const int g = cus * 8;
        RW2(6, 1, false, false, g) RW2(6, 1, true, false, g) RW2(6, 1, false, true, g) RW2(6, 1, true, true, g)
        RW2(6, 4, false, false, g) RW2(6, 4, true, false, g) RW2(6, 4, false, true, g) RW2(6, 4, true, true, g)   // 25 % writes (K2-like)
        RW2(4, 4, false, false, g) RW2(4, 4, true, false, g) RW2(4, 4, false, true, g) RW2(4, 4, true, true, g)   // 33 % writes (K3-like)
        RW2(2, 4, false, false, g) RW2(2, 4, true, true, g)                                                       // 50 % (copy)
    }
#define PROBE2(NT, GATHER, YST, TILE, DOT, BPC)                                                                 \
    if (!quick) {                                                                                                           \
        const int grid = (cus * BPC + 7) & ~7;                                                                  \
        char name[128];                                                                                         \
        snprintf(name, sizeof name, "dma-probe2 nt=%d gather=%d yst=%d tile=%d dot=%d wg/cu=%d", NT, GATHER, YST, TILE, DOT, BPC); \
        double ms = timeit([&] {                                                                                \
            spmv_dma_probe<NT, GATHER, YST, TILE, DOT><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb256, 32); \
        }, reps);                                                                                               \
        report(name, ms);                                                                                       \
    }
    PROBE2(true, 1, 1, 2048, false, 4) PROBE2(true, 1, 1, 2048, false, 5) PROBE2(true, 1, 1, 2048, false, 6)
    PROBE2(true, 1, 1, 2048, true, 6)
#define VARIANT(VAR, DOT)                                                                                       \
    {                                                                                                           \
        const int grid = (cus * 6 + 7) & ~7;                                                                    \
        char name[128];                                                                                         \
        snprintf(name, sizeof name, "dma-var %d dot=%d", VAR, DOT);                                              \
        double ms = timeit([&] {                                                                                \
            spmv_dma_var<VAR, DOT><<<grid, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb256, 32);      \
        }, reps);                                                                                               \
        check(P, name);                                                                                         \
        report(name, ms);                                                                                       \
    }
    VARIANT(0, false) VARIANT(1, false) VARIANT(2, false) VARIANT(3, false) VARIANT(4, false) VARIANT(5, false)
    VARIANT(0, true) VARIANT(1, true) VARIANT(4, true) VARIANT(5, true)
    VARIANT(0, false)
    return 0;
}
