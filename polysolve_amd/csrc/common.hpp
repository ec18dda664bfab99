// common.hpp -- shared host-side plumbing of libpsolve_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
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
    // shards: every rank throws at the same point of the collective sequence (Context::shards_agree) -- nobody is left
    // blocked in a collective, so the communicators stay as they are
    bool agreed = false;
    Error(int c, const std::string &m, bool all_ranks_agree = false) : std::runtime_error(m), code(c), agreed(all_ranks_agree) {}
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

// Device bytes held by the DeviceBuffers of one handle ("stats.device_bytes"): a handle points the calling thread's
// meter at its own counter on entry (Context::use_device), every allocation of that thread is charged to it and
// credited back on release.  Lets the multi-device tests assert that a shard holds ~1/N of the single-device bytes.
struct AllocMeter {
    std::atomic<long long> bytes{0}, peak{0};
    void add(long long b)
    {
        const long long now = bytes.fetch_add(b) + b;
        long long p = peak.load();
        while (now > p && !peak.compare_exchange_weak(p, now)) {
        }
    }
};
// the meter of the handle this thread last entered (Context::use_device); weak: a handle destroyed on another thread
// leaves nothing behind to charge
extern thread_local std::weak_ptr<AllocMeter> tl_alloc_meter;

// Owning device allocation.
template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    std::shared_ptr<AllocMeter> meter; // shared: a buffer may outlive the handle that was current when it was allocated
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (ptr) {
            (void)hipFree(ptr);
            if (meter) meter->add(-(long long)(count * sizeof(T)));
        }
        ptr = nullptr;
        count = 0;
        meter.reset();
    }
    void swap(DeviceBuffer &o)
    {
        std::swap(ptr, o.ptr);
        std::swap(count, o.count);
        std::swap(meter, o.meter);
    }
    // (re)allocate only when growing or when the size class changes a lot; contents undefined
    void ensure(size_t n)
    {
        if (n <= count && ptr) return;
        release();
        if (n == 0) n = 1;
        PS_HIP_CHECK(hipMalloc((void **)&ptr, n * sizeof(T)));
        count = n;
        meter = tl_alloc_meter.lock();
        if (meter) meter->add((long long)(n * sizeof(T)));
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
