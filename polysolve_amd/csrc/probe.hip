// probe.hip -- what THIS box's memory system does, in five numbers (bench.py's "box.probe" block).
//
// Round 3 saw one binary run the level-1 product of the 256^3 hierarchy in 177 us on one gpurun box and in 284 us on
// another while the streaming kernels moved by 7 %; rocm-smi's clocks and power read the same on both.  These probes
// measure what the gather-bound products depend on directly: the latency of a dependent load out of the L2 (1 MiB
// working set), the Infinity Cache (64 MiB) and HBM (1 GiB), and the rate of independent 8-byte gathers from a vector
// that lives in the L2 (2 MiB) resp. the Infinity Cache (64 MiB).  Not part of the hot path; no reference counterpart.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "solver.hpp"

namespace psolve {

// one cycle over m = 2^k slots of 64 bytes: next(i) = (a i + c) mod m with a = 1 (mod 4), c odd (Hull-Dobell)
__global__ void probe_chain_fill(unsigned long long *buf, unsigned m)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        buf[(size_t)i * 8] = (unsigned)((1664525ull * i + 1013904223ull) & (m - 1));
}

__global__ void probe_chain_walk(const unsigned long long *buf, int hops, unsigned long long *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long i = 0;
    for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(buf + i * 8) & 0xffffffffull; // (nt: no L1 reuse games; one line per hop anyway)
    *out = i;
}

// every thread: `per` independent gathers at pseudo-random positions of v[0, m), m a power of two
__global__ __launch_bounds__(256) void probe_gather(const double *__restrict__ v, unsigned m, int per, double *out)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    for (int k = 0; k < per; k += 4) {
        const unsigned a = s * 1664525u + 1013904223u, b = a * 1664525u + 1013904223u, c = b * 1664525u + 1013904223u,
                       d = c * 1664525u + 1013904223u;
        s = d;
        acc += v[(a >> 7) & (m - 1)] + v[(b >> 7) & (m - 1)] + v[(c >> 7) & (m - 1)] + v[(d >> 7) & (m - 1)];
    }
    if (acc == 1.2345e301) *out = acc; // (never; keeps the loads)
}

// shader clock under load: every CU spins on FMAs; one lane reads the shader-cycle counter (s_memtime) and the constant
// 100 MHz counter (s_memrealtime) around its own loop
__global__ __launch_bounds__(256) void probe_clock(int iters, unsigned long long *out, double *sink)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a = a * b + 1e-9;
        a = a * b + 1e-9;
        a = a * b + 1e-9;
        a = a * b + 1e-9;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (a == 1.2345e301) *sink = a;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
}

void Context::box_probe(double *out, int n_out)
{
    use_device();
    PS_REQUIRE(out && n_out >= 7, PSOLVE_HIP_EINVAL, "box_probe: seven results");
    hipEvent_t e0, e1;
    PS_HIP_CHECK(hipEventCreate(&e0));
    PS_HIP_CHECK(hipEventCreate(&e1));
    DeviceBuffer<unsigned long long> buf, res;
    res.ensure(8);
    const size_t sets[3] = {(size_t)1 << 20, (size_t)64 << 20, (size_t)1 << 30};
    buf.ensure(sets[2] / 8);
    float ms = 0;
    for (int k = 0; k < 3; ++k) {
        const unsigned m = (unsigned)(sets[k] / 64);
        hipLaunchKernelGGL(probe_chain_fill, dim3(2048), dim3(256), 0, stream, buf.ptr, m);
        const int hops = k == 0 ? 40000 : 20000;
        hipLaunchKernelGGL(probe_chain_walk, dim3(1), dim3(64), 0, stream, buf.ptr, (int)std::min<unsigned>(m, 4096u), res.ptr); // warm: TLB, caches
        if (k < 2) hipLaunchKernelGGL(probe_chain_walk, dim3(1), dim3(64), 0, stream, buf.ptr, (int)std::min<unsigned>(m, 1u << 20), res.ptr); // bring the set in
        PS_HIP_CHECK(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(probe_chain_walk, dim3(1), dim3(64), 0, stream, buf.ptr, hops, res.ptr);
        PS_HIP_CHECK(hipEventRecord(e1, stream));
        PS_HIP_CHECK(hipEventSynchronize(e1));
        PS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out[k] = (double)ms * 1e6 / hops; // ns per dependent load
    }
    const size_t gsets[2] = {(size_t)2 << 20, (size_t)64 << 20};
    PS_HIP_CHECK(hipMemsetAsync(buf.ptr, 0, gsets[1], stream));
    const int grid = num_cus_ * 8, per = 256;
    for (int k = 0; k < 2; ++k) {
        const unsigned m = (unsigned)(gsets[k] / 8);
        const double *v = reinterpret_cast<const double *>(buf.ptr);
        hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, stream, v, m, per, reinterpret_cast<double *>(res.ptr));
        PS_HIP_CHECK(hipEventRecord(e0, stream));
        for (int r = 0; r < 4; ++r)
            hipLaunchKernelGGL(probe_gather, dim3(grid), dim3(256), 0, stream, v, m, per, reinterpret_cast<double *>(res.ptr));
        PS_HIP_CHECK(hipEventRecord(e1, stream));
        PS_HIP_CHECK(hipEventSynchronize(e1));
        PS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        out[3 + k] = 4.0 * grid * 256.0 * per / ((double)ms * 1e-3) / 1e9; // G gathers per second
    }
    {
        hipLaunchKernelGGL(probe_clock, dim3(num_cus_ * 8), dim3(256), 0, stream, 200000, res.ptr, reinterpret_cast<double *>(res.ptr + 4));
        unsigned long long h[2] = {0, 0};
        PS_HIP_CHECK(hipMemcpyAsync(h, res.ptr, sizeof(h), hipMemcpyDeviceToHost, stream));
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        out[5] = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0; // s_memtime ticks per microsecond of the 100 MHz counter
        out[6] = (double)h[1] / 100.0;                             // length of the measurement, us
    }
    PS_HIP_CHECK(hipGetLastError());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

} // namespace psolve
