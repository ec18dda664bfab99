// mall_lab.hip -- does the 256 MB Infinity Cache keep the tail of what the previous kernel streamed?
// producer: y[i] = a * x[i] over n doubles (forward).  consumer: sum of y, forward or backward.
// If the backward consumer is faster, alternating the sweep direction of consecutive PCG kernels pays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void producer(size_t n2, double a, const v2d *x, v2d *y)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        v2d v = x[i];
        v.x *= a; v.y *= a;
        y[i] = v;
    }
}
__global__ __launch_bounds__(256) void consumer(size_t n2, const v2d *y, const v2d *z, v2d *out, int backward, int both)
{
    v2d acc = {0.0, 0.0};
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n2; k += (size_t)gridDim.x * 256) {
        const size_t i = backward ? n2 - 1 - k : k;
        v2d v = y[i];
        if (both) { v2d w = z[i]; v.x += w.x; v.y += w.y; }
        acc.x += v.x; acc.y += v.y;
        if (both) out[i] = v;
    }
    if (!both && acc.x == 123.456) out[0] = acc;
}
int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)16777216;
    const size_t n2 = n / 2;
    double *x, *y, *z, *o;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&z, n * 8)); CK(hipMalloc(&o, n * 8));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(z, 0, n * 8));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int both = 0; both < 2; ++both)
        for (int backward = 0; backward < 2; ++backward) {
            float tp = 0, tc = 0;
            const int reps = 20;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(producer, dim3(2048), dim3(256), 0, 0, n2, 1.0, (const v2d *)x, (v2d *)y);
                CK(hipEventRecord(e1));
                hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 0, 0, n2, (const v2d *)y, (const v2d *)z, (v2d *)o, backward, both);
                CK(hipEventRecord(e2));
                CK(hipEventSynchronize(e2));
                float a, b;
                CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
                if (r >= 2) { tp += a; tc += b; }
            }
            const double bytes = both ? 3.0 * n * 8 : 1.0 * n * 8;
            printf("n=%zu consumer %s %s: producer %.4f ms, consumer %.4f ms (%.0f GB/s)\n", n, both ? "y+z->o" : "sum(y)",
                   backward ? "BACKWARD" : "forward ", tp / reps, tc / reps, bytes / (tc / reps) / 1e6);
        }
    return 0;
}
