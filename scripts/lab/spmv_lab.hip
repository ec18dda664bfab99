// spmv_lab.hip -- kernel-variant laboratory for the CSR SpMV (not part of the product).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off spmv_lab.hip -o spmv_lab
// Run:   ./spmv_lab [N=256] [reps=20]
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

// ------------------------------------------------------------------ generator (same as product)
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

// ------------------------------------------------------------------ bandwidth probes
__global__ void copy16(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

template <bool NT>
__global__ void read16(const v4f *__restrict__ a, float *out, size_t n16)
{
    v4f s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        v4f v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        s += v;
    }
    float t = s.x + s.y + s.z + s.w;
    if (t == 1.2345f) out[0] = t; // never
}

// ------------------------------------------------------------------ V0: product kernel (block tile)
template <int BS, int TILE, bool NT, bool XCD, bool GATHER>
__global__ __launch_bounds__(BS) void spmv_block(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  int rb_per_xcd)
{
    __shared__ double prod[TILE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int nloop = XCD ? rb_per_xcd : nrb;
    for (int lrb = XCD ? slot : blockIdx.x; lrb < nloop; lrb += XCD ? slots : gridDim.x) {
        const int rb = XCD ? xcd * rb_per_xcd + lrb : lrb;
        if (rb >= nrb) break;
        const int row0 = rb * BS, r = row0 + tid;
        int rs = 0, re = 0;
        if (r < n) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int lo = rowptr[row0], hi = rowptr[min(row0 + BS, n)];
        double acc = 0.0;
        for (int c0 = lo & ~3; c0 < hi; c0 += TILE) {
            const int cend = min(c0 + TILE, hi);
            for (int i = c0 + tid * 4; i < cend; i += BS * 4) {
                if ((int64_t)i + 3 < nnz) {
                    const v4i c = NT ? __builtin_nontemporal_load((const v4i *)(col + i)) : *(const v4i *)(col + i);
                    const v2d v0 = NT ? __builtin_nontemporal_load((const v2d *)(val + i)) : *(const v2d *)(val + i);
                    const v2d v1 = NT ? __builtin_nontemporal_load((const v2d *)(val + i + 2)) : *(const v2d *)(val + i + 2);
                    double x0, x1, x2, x3;
                    if (GATHER) { x0 = x[c.x]; x1 = x[c.y]; x2 = x[c.z]; x3 = x[c.w]; }
                    else { x0 = (double)c.x; x1 = (double)c.y; x2 = (double)c.z; x3 = (double)c.w; }
                    v2d p0, p1;
                    p0.x = v0.x * x0; p0.y = v0.y * x1; p1.x = v1.x * x2; p1.y = v1.y * x3;
                    *(v2d *)(prod + (i - c0)) = p0;
                    *(v2d *)(prod + (i - c0) + 2) = p1;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if ((int64_t)i + k < nnz) prod[i - c0 + k] = val[i + k] * x[col[i + k]];
                }
            }
            __syncthreads();
            const int a = max(rs, c0), e = min(re, c0 + TILE);
            for (int j = a; j < e; ++j) acc += prod[j - c0];
            __syncthreads();
        }
        if (r < n) y[r] = acc;
    }
}

// ------------------------------------------------------------------ V1: wave-private tiles, no block barrier
// Each wave owns 64 consecutive rows and a private LDS slice; optional register prefetch of the
// next tile's (col,val) stream so that the col->x[col] dependent chain costs one latency per tile.
template <int WT, bool NT, bool PREFETCH>
__global__ __launch_bounds__(256) void spmv_wave(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int ntiles,
                                                  int tiles_per_xcd)
{
    constexpr int ROUNDS = WT / 256; // 4-wide groups per lane per chunk
    __shared__ double lds[4][WT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = lds[w];
    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) * 4 + w, slots = (gridDim.x >> 3) * 4;

    v4i cc[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs_n = 0, re_n = 0;

    auto load_tile = [&](int t, v4i *c, v2d *a, v2d *b, int &rs, int &re) {
        const int row0 = t * 64, r = row0 + lane;
        rs = 0; re = 0;
        if (r < n) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int lo = __builtin_amdgcn_readfirstlane(rowptr[row0]);
        const int hi = __builtin_amdgcn_readfirstlane(rowptr[min(row0 + 64, n)]);
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + lane * 4 + k * 256;
            if (i >= hi) {
                c[k] = (v4i){0, 0, 0, 0};
                a[k] = (v2d){0, 0};
                b[k] = (v2d){0, 0};
            } else if ((int64_t)i + 3 < nnz) {
                c[k] = NT ? __builtin_nontemporal_load((const v4i *)(col + i)) : *(const v4i *)(col + i);
                a[k] = NT ? __builtin_nontemporal_load((const v2d *)(val + i)) : *(const v2d *)(val + i);
                b[k] = NT ? __builtin_nontemporal_load((const v2d *)(val + i + 2)) : *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                a[k] = (v2d){0, 0};
                b[k] = (v2d){0, 0};
                for (int q = 0; q < 4; ++q)
                    if ((int64_t)i + q < nnz) {
                        int cv = col[i + q];
                        double vv = val[i + q];
                        if (q == 0) { c[k].x = cv; a[k].x = vv; }
                        if (q == 1) { c[k].y = cv; a[k].y = vv; }
                        if (q == 2) { c[k].z = cv; b[k].x = vv; }
                        if (q == 3) { c[k].w = cv; b[k].y = vv; }
                    }
            }
        }
    };

    int t = xcd * tiles_per_xcd + slot;
    const int t_end = min((xcd + 1) * tiles_per_xcd, ntiles);
    if (t >= t_end) return;
    int rs, re;
    load_tile(t, cc, va, vb, rs, re);
    for (; t < t_end; t += slots) {
        const int row0 = t * 64, r = row0 + lane;
        const int lo = __builtin_amdgcn_readfirstlane(rs);
        const int hi = __builtin_amdgcn_readlane(re, 63);
        // if the last tile is partial, lane 63 has re = 0: recompute
        const int hi2 = (row0 + 64 <= n) ? hi : rowptr[n];
        const int c0 = lo & ~3;
        v4i cn[ROUNDS];
        v2d an[ROUNDS], bn[ROUNDS];
        const int tn = t + slots;
        if (PREFETCH && tn < t_end) load_tile(tn, cn, an, bn, rs_n, re_n);
        // gather + products of the first chunk from registers
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + lane * 4 + k * 256;
            if (i < hi2) {
                v2d p0, p1;
                p0.x = va[k].x * x[cc[k].x];
                p0.y = va[k].y * x[cc[k].y];
                p1.x = vb[k].x * x[cc[k].z];
                p1.y = vb[k].y * x[cc[k].w];
                *(v2d *)(prod + (i - c0)) = p0;
                *(v2d *)(prod + (i - c0) + 2) = p1;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double acc = 0.0;
        {
            const int a = max(rs, c0), e = min(re, c0 + WT);
            for (int j = a; j < e; ++j) acc += prod[j - c0];
        }
        // remaining chunks (rows longer than the tile average): streamed the slow way
        for (int c1 = c0 + WT; c1 < hi2; c1 += WT) {
            __builtin_amdgcn_wave_barrier();
            for (int i = c1 + lane * 4; i < min(c1 + WT, hi2); i += 256)
                for (int q = 0; q < 4; ++q)
                    if ((int64_t)i + q < nnz) prod[i - c1 + q] = val[i + q] * x[col[i + q]];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int a = max(rs, c1), e = min(re, c1 + WT);
            for (int j = a; j < e; ++j) acc += prod[j - c1];
        }
        if (r < n) y[r] = acc;
        __builtin_amdgcn_wave_barrier();
        if (PREFETCH) {
            if (tn < t_end) {
#pragma unroll
                for (int k = 0; k < ROUNDS; ++k) { cc[k] = cn[k]; va[k] = an[k]; vb[k] = bn[k]; }
                rs = rs_n; re = re_n;
            }
        } else if (tn < t_end) {
            load_tile(tn, cc, va, vb, rs, re);
        }
    }
}

// ------------------------------------------------------------------ V2: thread-per-row, no LDS
template <bool XCD>
__global__ __launch_bounds__(256) void spmv_scalar(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const double *__restrict__ val, const double *__restrict__ x,
                                                    double *__restrict__ y, int nrb, int rb_per_xcd)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    for (int lrb = XCD ? slot : blockIdx.x; lrb < (XCD ? rb_per_xcd : nrb); lrb += XCD ? slots : gridDim.x) {
        const int rb = XCD ? xcd * rb_per_xcd + lrb : lrb;
        if (rb >= nrb) break;
        const int r = rb * 256 + threadIdx.x;
        if (r < n) {
            double acc = 0.0;
            for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) acc += val[j] * x[col[j]];
            y[r] = acc;
        }
    }
}


// ------------------------------------------------------------------ P1/P2: pattern ceilings (wrong results on purpose)
// same global access pattern as spmv_block, but no LDS/reduction: products summed in a register.
template <bool GATHER, bool YSTORE>
__global__ __launch_bounds__(256) void spmv_ceiling(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                     const int *__restrict__ col, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                     int rb_per_xcd)
{
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    for (int lrb = slot; lrb < rb_per_xcd; lrb += slots) {
        const int rb = xcd * rb_per_xcd + lrb;
        if (rb >= nrb) break;
        const int row0 = rb * 256, r = row0 + tid;
        int rs = 0, re = 0;
        if (r < n) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int lo = rowptr[row0], hi = rowptr[min(row0 + 256, n)];
        double acc = (double)(re - rs);
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi && (int64_t)i + 3 < nnz) {
                const v4i c = *(const v4i *)(col + i);
                const v2d v0 = *(const v2d *)(val + i);
                const v2d v1 = *(const v2d *)(val + i + 2);
                if (GATHER) acc += v0.x * x[c.x] + v0.y * x[c.y] + v1.x * x[c.z] + v1.y * x[c.w];
                else acc += v0.x * c.x + v0.y * c.y + v1.x * c.z + v1.y * c.w;
            }
        }
        if (r < n && (YSTORE || acc == 1.2345)) y[r] = acc;
    }
}

// ------------------------------------------------------------------ P4: pipelined, double-buffered LDS, 1 barrier / row-block
// Fast path only (every row-block's nnz fits TILE); unrolled rounds and reduction.
template <int TILE, bool XCD>
__global__ __launch_bounds__(256) void spmv_pipe(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  int rb_per_xcd)
{
    constexpr int ROUNDS = TILE / 1024;
    __shared__ double prod[2][TILE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    int lrb = XCD ? slot : blockIdx.x;
    const int step = XCD ? slots : gridDim.x;
    const int nloop = XCD ? rb_per_xcd : nrb;
    auto rbid = [&](int l) { return XCD ? xcd * rb_per_xcd + l : l; };
    if (lrb >= nloop || rbid(lrb) >= nrb) return;

    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs, re, lo, hi;
    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * 256, r = row0 + tid;
        rs_ = 0; re_ = 0;
        if (r < n) { rs_ = rowptr[r]; re_ = rowptr[r + 1]; }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + 256, n)];
    };
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi_ && (int64_t)i + 3 < nnz) {
                c[k] = *(const v4i *)(col + i);
                va[k] = *(const v2d *)(val + i);
                vb[k] = *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                va[k] = (v2d){0, 0};
                vb[k] = (v2d){0, 0};
                if (i < hi_)
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)i + q < nnz) {
                            int cv = col[i + q];
                            double vv = val[i + q];
                            if (q == 0) { c[k].x = cv; va[k].x = vv; }
                            if (q == 1) { c[k].y = cv; va[k].y = vv; }
                            if (q == 2) { c[k].z = cv; vb[k].x = vv; }
                            if (q == 3) { c[k].w = cv; vb[k].y = vv; }
                        }
            }
        }
    };
    load_ptr(rbid(lrb), rs, re, lo, hi);
    load_stream(lo, hi);
    int buf = 0;
    for (;;) {
        const int rb = rbid(lrb);
        const int c0 = lo & ~3;
        // next row-block's pointers (independent of everything below)
        const int lnext = lrb + step;
        const bool has_next = lnext < nloop && rbid(lnext) < nrb;
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rbid(lnext), rs_n, re_n, lo_n, hi_n);
        // A: gathers + products of the current block into LDS[buf]
        double *P = prod[buf];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi) {
                v2d p0, p1;
                p0.x = va[k].x * x[c[k].x];
                p0.y = va[k].y * x[c[k].y];
                p1.x = vb[k].x * x[c[k].z];
                p1.y = vb[k].y * x[c[k].w];
                *(v2d *)(P + (i - c0)) = p0;
                *(v2d *)(P + (i - c0) + 2) = p1;
            }
        }
        __syncthreads();
        // B: stream loads of the next block go out before the reduction
        if (has_next) load_stream(lo_n, hi_n);
        // C: reduction (in column order), unrolled by 8
        double acc = 0.0;
        {
            int j = rs - c0;
            const int e = re - c0;
            for (; j + 8 <= e; j += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (j + q < e) ? P[j + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j + q < e) acc += t[q];
        }
        const int r = rb * 256 + tid;
        if (r < n) y[r] = acc;
        if (!has_next) break;
        lrb = lnext;
        rs = rs_n; re = re_n; lo = lo_n; hi = hi_n;
        buf ^= 1;
    }
}


// ------------------------------------------------------------------ M: schedule experiments on the pipelined kernel
// MAP 0 round-robin | 1 XCD-contiguous | 2 XCD-contiguous with skewed (non power-of-two) region size |
// 3 chunks of CH row-blocks dealt round-robin to the XCDs
// Fast path only (every row-block's nnz fits TILE); unrolled rounds and reduction.
template <int TILE, int MAP, int CH>
__global__ __launch_bounds__(256) void spmv_map(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  int rb_per_xcd)
{
    constexpr int ROUNDS = TILE / 1024;
    __shared__ double prod[2][TILE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int skew = (MAP == 2) ? rb_per_xcd + 67 : rb_per_xcd; // region size of XCDs 0..6 (XCD 7 takes the rest)
    int lrb = (MAP == 0) ? blockIdx.x : slot;
    const int step = (MAP == 0) ? gridDim.x : slots;
    const int nchunks = (nrb + CH - 1) / CH;
    const int nloop = (MAP == 0) ? nrb : (MAP == 3 ? ((nchunks + 7) / 8) * CH : (MAP == 2 ? (xcd < 7 ? skew : nrb - 7 * skew) : rb_per_xcd));
    auto rbid = [&](int l) {
        if (MAP == 0) return l;
        if (MAP == 1) return xcd * rb_per_xcd + l;
        if (MAP == 2) return xcd * skew + l;
        return ((l / CH) * 8 + xcd) * CH + (l % CH); // chunk (l / CH) of this XCD
    };
    if (lrb >= nloop || rbid(lrb) >= nrb) return;

    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs, re, lo, hi;
    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * 256, r = row0 + tid;
        rs_ = 0; re_ = 0;
        if (r < n) { rs_ = rowptr[r]; re_ = rowptr[r + 1]; }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + 256, n)];
    };
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi_ && (int64_t)i + 3 < nnz) {
                c[k] = *(const v4i *)(col + i);
                va[k] = *(const v2d *)(val + i);
                vb[k] = *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                va[k] = (v2d){0, 0};
                vb[k] = (v2d){0, 0};
                if (i < hi_)
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)i + q < nnz) {
                            int cv = col[i + q];
                            double vv = val[i + q];
                            if (q == 0) { c[k].x = cv; va[k].x = vv; }
                            if (q == 1) { c[k].y = cv; va[k].y = vv; }
                            if (q == 2) { c[k].z = cv; vb[k].x = vv; }
                            if (q == 3) { c[k].w = cv; vb[k].y = vv; }
                        }
            }
        }
    };
    load_ptr(rbid(lrb), rs, re, lo, hi);
    load_stream(lo, hi);
    int buf = 0;
    for (;;) {
        const int rb = rbid(lrb);
        const int c0 = lo & ~3;
        // next row-block's pointers (independent of everything below)
        const int lnext = lrb + step;
        const bool has_next = lnext < nloop && rbid(lnext) < nrb && rbid(lnext) >= 0;
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rbid(lnext), rs_n, re_n, lo_n, hi_n);
        // A: gathers + products of the current block into LDS[buf]
        double *P = prod[buf];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi) {
                v2d p0, p1;
                p0.x = va[k].x * x[c[k].x];
                p0.y = va[k].y * x[c[k].y];
                p1.x = vb[k].x * x[c[k].z];
                p1.y = vb[k].y * x[c[k].w];
                *(v2d *)(P + (i - c0)) = p0;
                *(v2d *)(P + (i - c0) + 2) = p1;
            }
        }
        __syncthreads();
        // B: stream loads of the next block go out before the reduction
        if (has_next) load_stream(lo_n, hi_n);
        // C: reduction (in column order), unrolled by 8
        double acc = 0.0;
        {
            int j = rs - c0;
            const int e = re - c0;
            for (; j + 8 <= e; j += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (j + q < e) ? P[j + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j + q < e) acc += t[q];
        }
        const int r = rb * 256 + tid;
        if (r < n) y[r] = acc;
        if (!has_next) break;
        lrb = lnext;
        rs = rs_n; re = re_n; lo = lo_n; hi = hi_n;
        buf ^= 1;
    }
}


// ------------------------------------------------------------------ O: occupancy experiment: LDS tile < 2048 entries so that 5 workgroups fit a CU
// MAP 0 round-robin | 1 XCD-contiguous | 2 XCD-contiguous with skewed (non power-of-two) region size |
// 3 chunks of CH row-blocks dealt round-robin to the XCDs
// Fast path only (every row-block's nnz fits TILE); unrolled rounds and reduction.
template <int TILE, int MAP, int CH>
__global__ __launch_bounds__(256) void spmv_occ(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  int rb_per_xcd)
{
    constexpr int ROUNDS = (TILE + 1023) / 1024;
    __shared__ double prod[2][TILE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int skew = (MAP == 2) ? rb_per_xcd + 67 : rb_per_xcd; // region size of XCDs 0..6 (XCD 7 takes the rest)
    int lrb = (MAP == 0) ? blockIdx.x : slot;
    const int step = (MAP == 0) ? gridDim.x : slots;
    const int nchunks = (nrb + CH - 1) / CH;
    const int nloop = (MAP == 0) ? nrb : (MAP == 3 ? ((nchunks + 7) / 8) * CH : (MAP == 2 ? (xcd < 7 ? skew : nrb - 7 * skew) : rb_per_xcd));
    auto rbid = [&](int l) {
        if (MAP == 0) return l;
        if (MAP == 1) return xcd * rb_per_xcd + l;
        if (MAP == 2) return xcd * skew + l;
        return ((l / CH) * 8 + xcd) * CH + (l % CH); // chunk (l / CH) of this XCD
    };
    if (lrb >= nloop || rbid(lrb) >= nrb) return;

    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs, re, lo, hi;
    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * 256, r = row0 + tid;
        rs_ = 0; re_ = 0;
        if (r < n) { rs_ = rowptr[r]; re_ = rowptr[r + 1]; }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + 256, n)];
    };
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi_ && (int64_t)i + 3 < nnz) {
                c[k] = *(const v4i *)(col + i);
                va[k] = *(const v2d *)(val + i);
                vb[k] = *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                va[k] = (v2d){0, 0};
                vb[k] = (v2d){0, 0};
                if (i < hi_)
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)i + q < nnz) {
                            int cv = col[i + q];
                            double vv = val[i + q];
                            if (q == 0) { c[k].x = cv; va[k].x = vv; }
                            if (q == 1) { c[k].y = cv; va[k].y = vv; }
                            if (q == 2) { c[k].z = cv; vb[k].x = vv; }
                            if (q == 3) { c[k].w = cv; vb[k].y = vv; }
                        }
            }
        }
    };
    load_ptr(rbid(lrb), rs, re, lo, hi);
    load_stream(lo, hi);
    int buf = 0;
    for (;;) {
        const int rb = rbid(lrb);
        const int c0 = lo & ~3;
        // next row-block's pointers (independent of everything below)
        const int lnext = lrb + step;
        const bool has_next = lnext < nloop && rbid(lnext) < nrb && rbid(lnext) >= 0;
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rbid(lnext), rs_n, re_n, lo_n, hi_n);
        // A: gathers + products of the current block into LDS[buf]
        double *P = prod[buf];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi) {
                v2d p0, p1;
                p0.x = va[k].x * x[c[k].x];
                p0.y = va[k].y * x[c[k].y];
                p1.x = vb[k].x * x[c[k].z];
                p1.y = vb[k].y * x[c[k].w];
                *(v2d *)(P + (i - c0)) = p0;
                *(v2d *)(P + (i - c0) + 2) = p1;
            }
        }
        __syncthreads();
        // B: stream loads of the next block go out before the reduction
        if (has_next) load_stream(lo_n, hi_n);
        // C: reduction (in column order), unrolled by 8
        double acc = 0.0;
        {
            int j = rs - c0;
            const int e = re - c0;
            for (; j + 8 <= e; j += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (j + q < e) ? P[j + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j + q < e) acc += t[q];
        }
        const int r = rb * 256 + tid;
        if (r < n) y[r] = acc;
        if (!has_next) break;
        lrb = lnext;
        rs = rs_n; re = re_n; lo = lo_n; hi = hi_n;
        buf ^= 1;
    }
}


// ------------------------------------------------------------------ S: store-wave designs
// 5 waves per workgroup: waves 0-3 compute, wave 4 only writes y (staged through LDS), so the compute
// waves never have a store outstanding and their s_waitcnt vmcnt(N) never degenerates to a full drain.
// DEPTH2: gathers of block i+1 are issued before the products of block i are formed.
template <int TILE, bool XCD, bool DEPTH2, bool NTX>
__global__ __launch_bounds__(320) void spmv_sw(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                const int *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                int rb_per_xcd)
{
    constexpr int ROUNDS = TILE / 1024;
    __shared__ double prod[2][TILE];
    __shared__ double ybuf[2][256];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int step = XCD ? slots : gridDim.x;
    const int nloop = XCD ? rb_per_xcd : nrb;
    auto rbid = [&](int l) { return XCD ? xcd * rb_per_xcd + l : l; };
    auto valid = [&](int l) { return l < nloop && rbid(l) < nrb; };
    int lrb = XCD ? slot : blockIdx.x;
    if (!valid(lrb)) return;
    // number of iterations of this workgroup (uniform)
    int iters = 0;
    for (int l = lrb; valid(l); l += step) ++iters;

    if (tid >= 256) {
        // ---------------- store wave
        const int lane = tid - 256;
        for (int it = 0; it <= iters; ++it) {
            __syncthreads();
            if (it >= 1) {
                const int rb = rbid(lrb + (it - 1) * step);
                const double *Y = ybuf[(it - 1) & 1];
                const int r0 = rb * 256 + lane * 4;
                v2d a = *(const v2d *)(Y + lane * 4), b = *(const v2d *)(Y + lane * 4 + 2);
                if (r0 + 3 < n) {
                    *(v2d *)(y + r0) = a;
                    *(v2d *)(y + r0 + 2) = b;
                } else {
                    if (r0 < n) y[r0] = a.x;
                    if (r0 + 1 < n) y[r0 + 1] = a.y;
                    if (r0 + 2 < n) y[r0 + 2] = b.x;
                }
            }
        }
        return;
    }
    // ---------------- compute waves
    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    double xg[ROUNDS][4];
    int rs, re, lo, hi;
    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * 256, r = row0 + tid;
        rs_ = 0; re_ = 0;
        if (r < n) { rs_ = rowptr[r]; re_ = rowptr[r + 1]; }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + 256, n)];
    };
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi_ && (int64_t)i + 3 < nnz) {
                c[k] = *(const v4i *)(col + i);
                va[k] = *(const v2d *)(val + i);
                vb[k] = *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                va[k] = (v2d){0, 0};
                vb[k] = (v2d){0, 0};
                if (i < hi_)
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)i + q < nnz) {
                            int cv = col[i + q];
                            double vv = val[i + q];
                            if (q == 0) { c[k].x = cv; va[k].x = vv; }
                            if (q == 1) { c[k].y = cv; va[k].y = vv; }
                            if (q == 2) { c[k].z = cv; vb[k].x = vv; }
                            if (q == 3) { c[k].w = cv; vb[k].y = vv; }
                        }
            }
        }
    };
    auto gather = [&]() {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            xg[k][0] = x[c[k].x];
            xg[k][1] = x[c[k].y];
            xg[k][2] = x[c[k].z];
            xg[k][3] = x[c[k].w];
        }
    };
    load_ptr(rbid(lrb), rs, re, lo, hi);
    load_stream(lo, hi);
    if (DEPTH2) gather();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        const int c0 = lo & ~3;
        const int lnext = lrb + step;
        const bool has_next = it + 1 < iters;
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rbid(lnext), rs_n, re_n, lo_n, hi_n);
        double *P = prod[buf];
        if (!DEPTH2) gather();
        v2d p0[ROUNDS], p1[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            p0[k].x = va[k].x * xg[k][0];
            p0[k].y = va[k].y * xg[k][1];
            p1[k].x = vb[k].x * xg[k][2];
            p1[k].y = vb[k].y * xg[k][3];
        }
        if (DEPTH2 && has_next) {
            // registers of block i are consumed: refill with block i+1 and start its gathers now
            load_stream(lo_n, hi_n);
            gather();
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi) {
                *(v2d *)(P + (i - c0)) = p0[k];
                *(v2d *)(P + (i - c0) + 2) = p1[k];
            }
        }
        __syncthreads();
        if (!DEPTH2 && has_next) load_stream(lo_n, hi_n);
        double acc = 0.0;
        {
            int j = rs - c0;
            const int e = re - c0;
            for (; j + 8 <= e; j += 8) {
                double t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = P[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += t[q];
            }
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (j + q < e) ? P[j + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j + q < e) acc += t[q];
        }
        ybuf[buf][tid] = acc;
        lrb = lnext;
        rs = rs_n; re = re_n; lo = lo_n; hi = hi_n;
    }
    __syncthreads(); // publishes the last ybuf to the store wave
}


// ------------------------------------------------------------------ W: what do the y stores cost?  (ceiling pattern, store flavours)
// MODE 0 plain dwordx2 store, 1 nontemporal store, 2 no store, 3 store every 4th row-block as one burst
// of 4 (registers), 4 plain store but of a constant (no data dependency on the loads)
template <int MODE, int CHUNK = 1>
__global__ __launch_bounds__(256) void spmv_wtest(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                   const int *__restrict__ col, const double *__restrict__ val,
                                                   double *__restrict__ y, int nrb, int rb_per_xcd)
{
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    double keep[4];
    int keep_r[4];
    int nk = 0;
    for (int it = 0;; ++it) {
        // CHUNK consecutive row-blocks per workgroup before jumping by slots*CHUNK
        const int lrb = (it / CHUNK) * (slots * CHUNK) + slot * CHUNK + (it % CHUNK);
        if (lrb >= rb_per_xcd) break;
        const int rb = xcd * rb_per_xcd + lrb;
        if (rb >= nrb) break;
        const int row0 = rb * 256, r = row0 + tid;
        int rs = 0, re = 0;
        if (r < n) { rs = rowptr[r]; re = rowptr[r + 1]; }
        const int lo = rowptr[row0], hi = rowptr[min(row0 + 256, n)];
        double acc = (double)(re - rs);
        const int c0 = lo & ~3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = c0 + tid * 4 + k * 1024;
            if (i < hi && (int64_t)i + 3 < nnz) {
                const v4i c = *(const v4i *)(col + i);
                const v2d v0 = *(const v2d *)(val + i);
                const v2d v1 = *(const v2d *)(val + i + 2);
                acc += v0.x * c.x + v0.y * c.y + v1.x * c.z + v1.y * c.w;
            }
        }
        if (r < n) {
            if (MODE == 0) y[r] = acc;
            if (MODE == 1) __builtin_nontemporal_store(acc, y + r);
            if (MODE == 2 && acc == 1.2345) y[r] = acc;
            if (MODE == 4) { y[r] = 1.0; if (acc == 1.2345) y[r] = acc; }
            if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(y + r), "v"(acc) : "memory");
            if (MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(y + r), "v"(acc) : "memory");
            if (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(y + r), "v"(acc) : "memory");
            if (MODE == 3) {
                keep[nk] = acc; keep_r[nk] = r; ++nk;
                if (nk == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[keep_r[q]] = keep[q];
                    nk = 0;
                }
            }
        }
    }
    if (MODE == 3)
        for (int q = 0; q < nk; ++q) y[keep_r[q]] = keep[q];
}

// reads `nread16` 16-byte words with most workgroups while 1/16 of the workgroups only write `nwrite16` words
__global__ __launch_bounds__(256) void mixed_rw(const v4f *__restrict__ a, v4f *__restrict__ b, float *out,
                                                 size_t nread16, size_t nwrite16)
{
    if ((blockIdx.x & 15) == 15) {
        const size_t wb = blockIdx.x >> 4, nwb = gridDim.x >> 4;
        for (size_t i = wb * 256 + threadIdx.x; i < nwrite16; i += nwb * 256) b[i] = (v4f){1, 2, 3, 4};
    } else {
        const size_t rbk = blockIdx.x - (blockIdx.x >> 4), nrb_ = gridDim.x - (gridDim.x >> 4);
        v4f s = {0, 0, 0, 0};
        for (size_t i = rbk * 256 + threadIdx.x; i < nread16; i += nrb_ * 256) s += a[i];
        float t = s.x + s.y + s.z + s.w;
        if (t == 1.2345f) out[0] = t;
    }
}


// ------------------------------------------------------------------ G: "stage (col,val), walk rows" design
// Phase 1: coalesced 16-B loads of the row-block's (col, val) stream -> registers -> LDS (no gather
// dependency).  Phase 2: thread t walks ITS row out of LDS and gathers x[col] itself, so that gather
// instruction j of a wave touches the j-th entries of 64 consecutive rows: for banded / stencil / FEM
// matrices those addresses are (nearly) contiguous -> 4-5 cache lines per instruction instead of >10.
// Double-buffered LDS, one barrier per row-block; loads of block i+1 are in flight during phase 2 of i.
template <int BS, int TILE, bool XCD, int UNROLL>
__global__ __launch_bounds__(BS) void spmv_stage(int n, int64_t nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  int rb_per_xcd)
{
    constexpr int ROUNDS = TILE / (BS * 4);
    __shared__ double sval[2][TILE];
    __shared__ int scol[2][TILE];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int step = XCD ? slots : gridDim.x;
    const int nloop = XCD ? rb_per_xcd : nrb;
    auto rbid = [&](int l) { return XCD ? xcd * rb_per_xcd + l : l; };
    auto valid = [&](int l) { return l < nloop && rbid(l) < nrb; };
    int lrb = XCD ? slot : blockIdx.x;
    if (!valid(lrb)) return;

    v4i c[ROUNDS];
    v2d va[ROUNDS], vb[ROUNDS];
    int rs, re, lo, hi;
    auto load_ptr = [&](int rb, int &rs_, int &re_, int &lo_, int &hi_) {
        const int row0 = rb * BS, r = row0 + tid;
        rs_ = 0; re_ = 0;
        if (r < n) { rs_ = rowptr[r]; re_ = rowptr[r + 1]; }
        lo_ = rowptr[row0];
        hi_ = rowptr[min(row0 + BS, n)];
    };
    auto load_stream = [&](int lo_, int hi_) {
        const int c0 = lo_ & ~3;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int i = c0 + tid * 4 + k * BS * 4;
            if (i < hi_ && (int64_t)i + 3 < nnz) {
                c[k] = *(const v4i *)(col + i);
                va[k] = *(const v2d *)(val + i);
                vb[k] = *(const v2d *)(val + i + 2);
            } else {
                c[k] = (v4i){0, 0, 0, 0};
                va[k] = (v2d){0, 0};
                vb[k] = (v2d){0, 0};
                if (i < hi_)
                    for (int q = 0; q < 4; ++q)
                        if ((int64_t)i + q < nnz) {
                            int cv = col[i + q];
                            double vv = val[i + q];
                            if (q == 0) { c[k].x = cv; va[k].x = vv; }
                            if (q == 1) { c[k].y = cv; va[k].y = vv; }
                            if (q == 2) { c[k].z = cv; vb[k].x = vv; }
                            if (q == 3) { c[k].w = cv; vb[k].y = vv; }
                        }
            }
        }
    };
    load_ptr(rbid(lrb), rs, re, lo, hi);
    load_stream(lo, hi);
    int buf = 0;
    for (;;) {
        const int rb = rbid(lrb);
        const int c0 = lo & ~3;
        const int lnext = lrb + step;
        const bool has_next = valid(lnext);
        int rs_n = 0, re_n = 0, lo_n = 0, hi_n = 0;
        if (has_next) load_ptr(rbid(lnext), rs_n, re_n, lo_n, hi_n);
        // A: registers -> LDS
        double *SV = sval[buf];
        int *SC = scol[buf];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int o = tid * 4 + k * BS * 4;
            if (c0 + o < hi) {
                *(v4i *)(SC + o) = c[k];
                *(v2d *)(SV + o) = va[k];
                *(v2d *)(SV + o + 2) = vb[k];
            }
        }
        __syncthreads();
        // B: next block's stream goes out now
        if (has_next) load_stream(lo_n, hi_n);
        // C: walk my row
        double acc = 0.0;
        {
            int j = rs - c0;
            const int e = re - c0;
            for (; j + UNROLL <= e; j += UNROLL) {
                int cc[UNROLL];
                double vv[UNROLL], xx[UNROLL];
#pragma unroll
                for (int q = 0; q < UNROLL; ++q) { cc[q] = SC[j + q]; vv[q] = SV[j + q]; }
#pragma unroll
                for (int q = 0; q < UNROLL; ++q) xx[q] = x[cc[q]];
#pragma unroll
                for (int q = 0; q < UNROLL; ++q) acc += vv[q] * xx[q];
            }
            if (j < e) {
                int cc[UNROLL];
                double vv[UNROLL], xx[UNROLL];
#pragma unroll
                for (int q = 0; q < UNROLL; ++q) { const bool ok = j + q < e; cc[q] = ok ? SC[j + q] : 0; vv[q] = ok ? SV[j + q] : 0.0; }
#pragma unroll
                for (int q = 0; q < UNROLL; ++q) xx[q] = (j + q < e) ? x[cc[q]] : 0.0;
#pragma unroll
                for (int q = 0; q < UNROLL; ++q)
                    if (j + q < e) acc += vv[q] * xx[q];
            }
        }
        const int r = rb * BS + tid;
        if (r < n) y[r] = acc;
        if (!has_next) break;
        lrb = lnext;
        rs = rs_n; re = re_n; lo = lo_n; hi = hi_n;
        buf ^= 1;
    }
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
    if (bad) printf("   !! %s: %zu rows differ from the reference kernel\n", name, bad);
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
    CK(hipMalloc(&P.col, (size_t)(P.nnz + 8) * 4));
    CK(hipMalloc(&P.val, (size_t)(P.nnz + 8) * 8));
    CK(hipMalloc(&P.x, (size_t)P.n * 8));
    CK(hipMalloc(&P.y, (size_t)P.n * 8));
    CK(hipMalloc(&P.yref, (size_t)P.n * 8));
    gen<<<4096, 256>>>(N, N, N, P.rowptr, P.col, P.val);
    fillx<<<4096, 256>>>(P.n, P.x);
    CK(hipDeviceSynchronize());
    const double bytes = 12.0 * P.nnz + 20.0 * P.n;
    auto report = [&](const char *name, double ms) {
        printf("%-44s %8.4f ms  %7.1f GB/s (alg)  %5.1f%% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
    };

    // bandwidth probes on the val array (0.94 GB) -> col array region
    {
        size_t n16 = (size_t)P.nnz * 8 / 16;
        float *out;
        CK(hipMalloc(&out, 64));
        v4f *dst;
        CK(hipMalloc(&dst, n16 * 16));
        for (int g : {cus * 8, cus * 16, cus * 32}) {
            double ms = timeit([&] { copy16<<<g, 256>>>((const v4f *)P.val, dst, n16); }, reps);
            printf("copy16   grid=%5d  %8.4f ms  %7.1f GB/s (r+w)\n", g, ms, 2.0 * n16 * 16 / ms / 1e6);
            ms = timeit([&] { read16<false><<<g, 256>>>((const v4f *)P.val, out, n16); }, reps);
            printf("read16   grid=%5d  %8.4f ms  %7.1f GB/s\n", g, ms, 1.0 * n16 * 16 / ms / 1e6);
            ms = timeit([&] { read16<true><<<g, 256>>>((const v4f *)P.val, out, n16); }, reps);
            printf("read16nt grid=%5d  %8.4f ms  %7.1f GB/s\n", g, ms, 1.0 * n16 * 16 / ms / 1e6);
        }
        CK(hipFree(dst));
    }

    const int nrb = (P.n + 255) / 256, rbx = (nrb + 7) / 8;
    // reference result
    spmv_scalar<false><<<cus * 8, 256>>>(P.n, P.rowptr, P.col, P.val, P.x, P.yref, nrb, rbx);
    CK(hipDeviceSynchronize());

    for (int bpc : {4, 5, 6}) {
        const int g = cus * bpc;
        printf("---- grid = %d CUs x %d\n", cus, bpc);
        char name[128];
#define RUN_BLOCK(BS, TILE, NT, XCD, GATHER)                                                                        \
    {                                                                                                               \
        const int nrb_ = (P.n + BS - 1) / BS, rbx_ = (nrb_ + 7) / 8;                                                \
        double ms = timeit([&] { spmv_block<BS, TILE, NT, XCD, GATHER><<<g, BS>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb_, rbx_); }, reps); \
        snprintf(name, sizeof name, "block BS=%d TILE=%d nt=%d xcd=%d gather=%d", BS, TILE, NT, XCD, GATHER);       \
        report(name, ms);                                                                                           \
        if (GATHER) check(P, name);                                                                                 \
    }
        RUN_BLOCK(256, 2048, true, true, true);
        RUN_BLOCK(256, 2048, false, true, true);
        RUN_BLOCK(256, 2048, false, false, true);
        RUN_BLOCK(256, 2048, true, true, false);
        RUN_BLOCK(256, 2048, false, true, false);

        {
            double ms = timeit([&] { spmv_ceiling<false, true><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("ceiling: stream only (no gather, no LDS)", ms);
            ms = timeit([&] { spmv_ceiling<false, false><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("ceiling: stream only, no y store", ms);
            ms = timeit([&] { spmv_ceiling<true, true><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("ceiling: stream + gather (no LDS)", ms);
            CK(hipMemset(P.y, 0, (size_t)P.n * 8));
            ms = timeit([&] { spmv_pipe<2048, true><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("pipe TILE=2048 xcd=1 (dbuf LDS, 1 barrier)", ms);
            check(P, "pipe xcd=1");
            ms = timeit([&] { spmv_pipe<2048, false><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("pipe TILE=2048 xcd=0", ms);
            check(P, "pipe xcd=0");
        }

        {
#define RUN_SW(TILE, XCD, D2)                                                                                       \
    {                                                                                                               \
        double ms = timeit([&] { spmv_sw<TILE, XCD, D2, false><<<g, 320>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps); \
        snprintf(name, sizeof name, "store-wave TILE=%d xcd=%d depth2=%d", TILE, XCD, D2);                          \
        report(name, ms);                                                                                           \
        check(P, name);                                                                                             \
    }
            RUN_SW(2048, true, false);
            RUN_SW(2048, true, true);
            RUN_SW(2048, false, false);
            RUN_SW(2048, false, true);
        }

        {
            const char *nm[5] = {"wtest plain store", "wtest nontemporal store", "wtest no store", "wtest burst-of-4 stores", "wtest store constant"};
            double ms;
            ms = timeit([&] { spmv_wtest<0><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report(nm[0], ms);
            ms = timeit([&] { spmv_wtest<1><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report(nm[1], ms);
            ms = timeit([&] { spmv_wtest<2><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report(nm[2], ms);
            ms = timeit([&] { spmv_wtest<3><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report(nm[3], ms);
            ms = timeit([&] { spmv_wtest<4><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report(nm[4], ms);
            ms = timeit([&] { spmv_wtest<5><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest store sc1", ms);
            ms = timeit([&] { spmv_wtest<6><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest store sc0 sc1", ms);
            ms = timeit([&] { spmv_wtest<7><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest store sc0 sc1 nt", ms);
            ms = timeit([&] { spmv_wtest<0, 4><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest plain store, chunk 4", ms);
            ms = timeit([&] { spmv_wtest<0, 16><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest plain store, chunk 16", ms);
            ms = timeit([&] { spmv_wtest<0, 64><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest plain store, chunk 64", ms);
            ms = timeit([&] { spmv_wtest<2, 16><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.y, nrb, rbx); }, reps); report("wtest no store, chunk 16", ms);
            float *out; CK(hipMalloc(&out, 64));
            size_t nread16 = (size_t)P.nnz * 8 / 16, nwrite16 = (size_t)P.n * 8 / 16;
            ms = timeit([&] { mixed_rw<<<g, 256>>>((const v4f *)P.val, (v4f *)P.y, out, nread16, nwrite16); }, reps);
            printf("mixed_rw: read %.0f MB + write %.0f MB by separate workgroups: %.4f ms -> %.1f GB/s\n", nread16 * 16e-6, nwrite16 * 16e-6, ms, (nread16 + nwrite16) * 16 / ms / 1e6);
            ms = timeit([&] { mixed_rw<<<g, 256>>>((const v4f *)P.val, (v4f *)P.y, out, nread16, 0); }, reps);
            printf("mixed_rw: read only (15/16 of the workgroups): %.4f ms -> %.1f GB/s\n", ms, nread16 * 16 / ms / 1e6);
            CK(hipFree(out));
        }

        {
#define RUN_STAGE(BS, TILE, XCD, UN)                                                                                \
    {                                                                                                               \
        const int nrb_ = (P.n + BS - 1) / BS, rbx_ = (nrb_ + 7) / 8;                                                \
        double ms = timeit([&] { spmv_stage<BS, TILE, XCD, UN><<<g, BS>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb_, rbx_); }, reps); \
        snprintf(name, sizeof name, "stage BS=%d TILE=%d xcd=%d unroll=%d", BS, TILE, XCD, UN);                     \
        report(name, ms);                                                                                           \
        check(P, name);                                                                                             \
    }
            RUN_STAGE(256, 2048, true, 8);
            RUN_STAGE(256, 2048, false, 8);
            RUN_STAGE(256, 2048, true, 4);
            RUN_STAGE(128, 1024, true, 8);
            RUN_STAGE(128, 1024, false, 8);
            RUN_STAGE(512, 4096, true, 8);
        }

        {
#define RUN_MAP(MAP, CH)                                                                                            \
    {                                                                                                               \
        double ms = timeit([&] { spmv_map<2048, MAP, CH><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps); \
        snprintf(name, sizeof name, "map=%d chunk=%d (pipe kernel)", MAP, CH);                                       \
        report(name, ms);                                                                                           \
        check(P, name);                                                                                             \
    }
            RUN_MAP(0, 1);
            RUN_MAP(1, 1);
            RUN_MAP(2, 1);
            RUN_MAP(3, 8);
            RUN_MAP(3, 32);
            RUN_MAP(3, 128);
        }

        {
            double ms = timeit([&] { spmv_occ<1856, 3, 32><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("occ: TILE=1856 (5 WG/CU fit) map=3 chunk=32", ms);
            check(P, "occ1856");
            ms = timeit([&] { spmv_occ<1800, 3, 32><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("occ: TILE=1800 map=3 chunk=32", ms);
            check(P, "occ1800");
        }
#define RUN_WAVE(WT, NT, PF)                                                                                        \
    {                                                                                                               \
        const int nt_ = (P.n + 63) / 64, tx_ = (nt_ + 7) / 8;                                                       \
        double ms = timeit([&] { spmv_wave<WT, NT, PF><<<g, 256>>>(P.n, P.nnz, P.rowptr, P.col, P.val, P.x, P.y, nt_, tx_); }, reps); \
        snprintf(name, sizeof name, "wave WT=%d nt=%d prefetch=%d", WT, NT, PF);                                    \
        report(name, ms);                                                                                           \
        check(P, name);                                                                                             \
    }
        RUN_WAVE(512, false, true);
        {
            double ms = timeit([&] { spmv_scalar<true><<<g, 256>>>(P.n, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("scalar thread-per-row xcd=1", ms);
            ms = timeit([&] { spmv_scalar<false><<<g, 256>>>(P.n, P.rowptr, P.col, P.val, P.x, P.y, nrb, rbx); }, reps);
            report("scalar thread-per-row xcd=0", ms);
        }
    }
    return 0;
}
