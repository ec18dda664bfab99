// gridsync_lab.hip -- how long is a grid-wide barrier on this GPU?  (cooperative launch, N syncs in a row;
// also a hand-rolled sense-reversing barrier on one L2 atomic for comparison)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_cg(int iters, double *out)
{
    cg::grid_group g = cg::this_grid();
    double a = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0000001 + 1.0;
        g.sync();
    }
    if (a == 12345.678) out[0] = a;
}
__global__ __launch_bounds__(256) void k_own(int iters, unsigned *bar, double *out)
{
    double a = threadIdx.x;
    const unsigned nb = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0000001 + 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(i + 1) * nb;
            atomicAdd(bar, 1u);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (a == 12345.678) out[0] = a;
}
int main()
{
    double *out; unsigned *bar;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&bar, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {256, 512, 1024, 2048}) {
        int iters = 2000;
        void *args[] = {&iters, &out};
        hipError_t e = hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(256), args, 0, 0);
        if (e != hipSuccess) { printf("blocks=%d cooperative launch: %s\n", blocks, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks=%d cg grid.sync: %.2f us per sync\n", blocks, ms * 1e3 / iters);
        CK(hipMemset(bar, 0, 4));
        void *args2[] = {&iters, &bar, &out};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void *)k_own, dim3(blocks), dim3(256), args2, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks=%d atomic barrier: %.2f us per sync\n", blocks, ms * 1e3 / iters);
    }
    return 0;
}
