// common.hpp -- shared host-side plumbing of libpsolve_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>

#include "../../include/psolve_hip.h"

namespace psolve {

// Errors travel as exceptions inside the library and are converted to status codes at the C ABI.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define PS_HIP_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            throw ::psolve::Error(PSOLVE_HIP_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_) + \
                                                          " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define PS_REQUIRE(cond, code, msg)                      \
    do {                                                 \
        if (!(cond)) throw ::psolve::Error((code), (msg)); \
    } while (0)

// Owning device allocation.
template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    void swap(DeviceBuffer &o)
    {
        std::swap(ptr, o.ptr);
        std::swap(count, o.count);
    }
    // (re)allocate only when growing or when the size class changes a lot; contents undefined
    void ensure(size_t n)
    {
        if (n <= count && ptr) return;
        release();
        if (n == 0) n = 1;
        PS_HIP_CHECK(hipMalloc((void **)&ptr, n * sizeof(T)));
        count = n;
    }
};

// Pinned host allocation (async D2H polling target).
template <typename T>
struct PinnedBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    ~PinnedBuffer()
    {
        if (ptr) (void)hipHostFree(ptr);
    }
    void ensure(size_t n)
    {
        if (n <= count && ptr) return;
        if (ptr) (void)hipHostFree(ptr);
        PS_HIP_CHECK(hipHostMalloc((void **)&ptr, n * sizeof(T), hipHostMallocDefault));
        count = n;
    }
};

double wall_seconds();

} // namespace psolve
