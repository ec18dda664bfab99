// read_lab.hip -- what a read-only stream reaches on this box beyond the Infinity Cache (the ceiling of a product whose bytes
// are 92 % matrix reads): 16-byte loads grid-stride (plain / non-temporal), workgroup-contiguous 4 KiB steps unrolled U deep,
// LDS-DMA staging as spmv_csr_dma does it, and the plain copy next to them.  usage: read_lab [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT, int U>
__global__ __launch_bounds__(256) void read_stride(size_t n2, const v2d *__restrict__ x, v2d *out)
{
    v2d acc = {0.0, 0.0};
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * step < n2; i += U * step) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * step) : x[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
    }
    for (; i < n2; i += step) { v2d v = x[i]; acc.x += v.x; acc.y += v.y; }
    if (acc.x == 123.456) out[0] = acc;
}

// every workgroup owns contiguous spans of `span` 16-byte elements, dealt round-robin
template <bool NT, int U>
__global__ __launch_bounds__(256) void read_spans(size_t n2, const v2d *__restrict__ x, v2d *out, int span)
{
    v2d acc = {0.0, 0.0};
    const size_t nspan = n2 / span;
    for (size_t s = blockIdx.x; s < nspan; s += gridDim.x) {
        const v2d *p = x + s * span;
        for (int i = threadIdx.x; i + (U - 1) * 256 < span; i += U * 256) {
            v2d v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
        }
    }
    if (acc.x == 123.456) out[0] = acc;
}

// LDS-DMA: a workgroup step brings `tile` 16-byte elements into LDS, barrier, every thread sums its share, barrier
template <bool NT>
__global__ __launch_bounds__(256) void read_dma(size_t n2, const v2d *__restrict__ x, v2d *out, int tile)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2d *l = reinterpret_cast<v2d *>(smem);
    v2d acc = {0.0, 0.0};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t ntile = n2 / tile;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const v2d *p = x + t * tile;
        for (int e = wave * 64; e < tile; e += 256)
            __builtin_amdgcn_global_load_lds(p + e + lane, (__attribute__((address_space(3))) void *)(l + e), 16, 0, NT ? 2 : 0);
        __syncthreads();
        for (int e = threadIdx.x; e < tile; e += 256) { v2d v = l[e]; acc.x += v.x; acc.y += v.y; }
        __syncthreads();
    }
    if (acc.x == 123.456) out[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void copy_stride(size_t n2, const v2d *__restrict__ x, v2d *__restrict__ y)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
        else y[i] = x[i];
    }
}

// RPW units read per unit written (12: the byte mix of a 7-point product with 32-bit columns; 8: without columns).
// LDP / STP: 0 plain, 1 non-temporal.  BURST: results of BURST steps are kept in registers and written together
// (BURST x 4 KiB contiguous per workgroup), to see whether fewer, longer write bursts cost the read stream less.
template <int LDP, int STP, int RPW, int BURST>
__global__ __launch_bounds__(256) void read_write_mix(size_t n2, const v2d *__restrict__ x, v2d *__restrict__ y)
{
    // a workgroup step reads RPW * 256 consecutive elements of x and produces 256 elements of y
    const size_t nstep = n2 / ((size_t)RPW * 256);
    for (size_t s0 = (size_t)blockIdx.x * BURST; s0 + BURST <= nstep; s0 += (size_t)gridDim.x * BURST) {
        v2d out[BURST];
#pragma unroll
        for (int k = 0; k < BURST; ++k) {
            const v2d *p = x + (s0 + k) * RPW * 256 + threadIdx.x;
            v2d acc = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                const v2d v = LDP ? __builtin_nontemporal_load(p + u * 256) : p[u * 256];
                acc.x += v.x; acc.y += v.y;
            }
            out[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < BURST; ++k) {
            v2d *q = y + (s0 + k) * 256 + threadIdx.x;
            if (STP) __builtin_nontemporal_store(out[k], q);
            else *q = out[k];
        }
    }
}

static hipEvent_t e0, e1;
template <typename F>
static void timeit(const char *name, double bytes, F f)
{
    float best = 1e30f, tot = 0;
    const int reps = 8;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) { tot += t; best = t < best ? t : best; }
    }
    printf("%-44s avg %.4f ms %6.0f GB/s   best %6.0f GB/s\n", name, tot / reps, bytes / (tot / reps) / 1e6, bytes / best / 1e6);
    CK(hipGetLastError());
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t n2 = (size_t)(gib * 1024 * 1024 * 1024 / 16) / (12 * 4096) * (12 * 4096);
    v2d *x, *y;
    CK(hipMalloc(&x, n2 * 16)); CK(hipMalloc(&y, n2 * 16));
    CK(hipMemset(x, 0, n2 * 16)); CK(hipMemset(y, 0, n2 * 16));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double B = (double)n2 * 16;
    char nm[128];
    const int quick0 = argc > 2 ? atoi(argv[2]) : 0;
    if (!quick0)
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        snprintf(nm, sizeof nm, "read stride plain U1 grid %d", grid);
        timeit(nm, B, [&] { hipLaunchKernelGGL((read_stride<false, 1>), dim3(grid), dim3(256), 0, 0, n2, x, y); });
        snprintf(nm, sizeof nm, "read stride plain U4 grid %d", grid);
        timeit(nm, B, [&] { hipLaunchKernelGGL((read_stride<false, 4>), dim3(grid), dim3(256), 0, 0, n2, x, y); });
        snprintf(nm, sizeof nm, "read stride nt    U4 grid %d", grid);
        timeit(nm, B, [&] { hipLaunchKernelGGL((read_stride<true, 4>), dim3(grid), dim3(256), 0, 0, n2, x, y); });
    }
    if (!quick0)
    for (int grid : {1536, 2048, 4096}) {
        for (int span : {1024, 4096}) {
            snprintf(nm, sizeof nm, "read spans %d x16B plain U4 grid %d", span, grid);
            timeit(nm, B, [&] { hipLaunchKernelGGL((read_spans<false, 4>), dim3(grid), dim3(256), 0, 0, n2, x, y, span); });
            snprintf(nm, sizeof nm, "read spans %d x16B nt    U4 grid %d", span, grid);
            timeit(nm, B, [&] { hipLaunchKernelGGL((read_spans<true, 4>), dim3(grid), dim3(256), 0, 0, n2, x, y, span); });
        }
    }
    if (!quick0)
    for (int grid : {1024, 1536, 2048}) {
        for (int tile : {1024, 1536, 2048}) {
            snprintf(nm, sizeof nm, "read lds-dma tile %d KiB plain grid %d", tile * 16 / 1024, grid);
            timeit(nm, B, [&] { hipLaunchKernelGGL((read_dma<false>), dim3(grid), dim3(256), tile * 16, 0, n2, x, y, tile); });
            snprintf(nm, sizeof nm, "read lds-dma tile %d KiB nt    grid %d", tile * 16 / 1024, grid);
            timeit(nm, B, [&] { hipLaunchKernelGGL((read_dma<true>), dim3(grid), dim3(256), tile * 16, 0, n2, x, y, tile); });
        }
    }
    for (int grid : {2048, 8192}) {
        snprintf(nm, sizeof nm, "copy plain grid %d (read + write bytes)", grid);
        timeit(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_stride<false>), dim3(grid), dim3(256), 0, 0, n2, x, y); });
        snprintf(nm, sizeof nm, "copy nt    grid %d (read + write bytes)", grid);
        timeit(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_stride<true>), dim3(grid), dim3(256), 0, 0, n2, x, y); });
    }
#define MIX(LDP, STP, RPW, BURST)                                                                                          \
    for (int grid : {2048, 8192}) {                                                                                         \
        snprintf(nm, sizeof nm, "read %d : write 1  ld %s st %s burst %d grid %d", RPW, LDP ? "nt   " : "plain", STP ? "nt   " : "plain", BURST, grid); \
        timeit(nm, B * (RPW + 1.0) / RPW, [&] { hipLaunchKernelGGL((read_write_mix<LDP, STP, RPW, BURST>), dim3(grid), dim3(256), 0, 0, n2, x, y); }); \
    }
    MIX(0, 0, 12, 1) MIX(1, 1, 12, 1) MIX(1, 0, 12, 1) MIX(0, 1, 12, 1)
    MIX(1, 1, 12, 4) MIX(1, 0, 12, 4) MIX(1, 1, 12, 8) MIX(1, 0, 12, 8)
    MIX(0, 0, 8, 1) MIX(1, 1, 8, 1) MIX(1, 0, 8, 1) MIX(1, 1, 8, 4) MIX(1, 0, 8, 4)
    MIX(1, 1, 24, 1) MIX(1, 0, 24, 1)
    timeit("hipMemcpyAsync D2D (read + write bytes)", 2 * B, [&] { CK(hipMemcpyAsync(y, x, n2 * 16, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
