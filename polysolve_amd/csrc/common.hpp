// common.hpp -- shared host-side plumbing of libpsolve_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <cstdint>
#include <cstdlib>
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
//
// Round 4: the meter also keeps the handle's RELEASED blocks for its next allocations (a cache, "stats.device_bytes_cached").
// hipFree synchronises the device and hipMalloc maps fresh pages: a second full AMG setup on one handle (a Newton loop whose
// sparsity pattern changes) spent a tenth of its time in the two.  Everything a handle does is ordered on its one stream, so
// a block that goes from one buffer to the next is written by work queued behind the work that still reads it.  Blocks of at
// least 1 MiB, at most `cap` bytes in total ("lab.alloc_cache_mb", 0 = off); a request takes the smallest cached block that
// holds it if that wastes no more than a quarter; a failed hipMalloc empties the cache and tries again.
struct AllocMeter {
    // the handle's own knobs (round 6: members, not process-wide globals).  "lab.alloc_cache_mb": released device blocks the
    // handle keeps for its next allocations; "lab.alloc_cache_poison": recycled blocks are filled with 0xFF bytes first
    // (tests) -- its default comes from the environment (PSOLVE_ALLOC_CACHE_POISON=1) when the handle is created, which is how
    // the test session poisons every handle without touching any of them.
    int cache_mb = 4096; // (Context's constructor raises it to min(16 GiB, device memory / 16))
    int poison = [] { const char *e = std::getenv("PSOLVE_ALLOC_CACHE_POISON"); return (e && e[0] == '1') ? 1 : ((e && e[0] == '2') ? 2 : 0); }(); // 2: fresh blocks too
    std::atomic<long long> bytes{0}, peak{0};
    void add(long long b)
    {
        const long long now = bytes.fetch_add(b) + b;
        long long p = peak.load();
        while (now > p && !peak.compare_exchange_weak(p, now)) {
        }
    }
    std::mutex mu;
    std::multimap<size_t, void *> blocks; // released blocks by size
    size_t cached = 0;
    bool closed = false; // (the handle is being destroyed: nothing is kept any more)
    // The cache drops hipFree's implicit device synchronisation: a recycled block is safe because whoever writes it next is
    // queued, on the handle's ONE stream, behind whoever still reads it.  Where a handle runs work on a second stream beside its
    // main one (the smoothers' power iterations of an AMG setup, amg.hip: fork ... join), blocks released DURING such a phase
    // are parked (`parked`) and only enter the cache at the join, when both streams have met again; what the cache holds during
    // the phase was released before the fork, i.e. behind an event both streams wait for (round-4 advice).
    std::vector<std::pair<size_t, void *>> parked;
    int forked = 0;
    size_t cap_bytes() const { return (size_t)cache_mb << 20; }
    void *take(size_t need, size_t *got)
    {
        if (need < ((size_t)1 << 20)) return nullptr; // (small requests are not served with blocks of a MiB and more)
        std::lock_guard<std::mutex> lk(mu);
        auto it = blocks.lower_bound(need);
        if (it == blocks.end() || it->first > need + need / 4 + ((size_t)1 << 20)) return nullptr;
        void *p = it->second;
        *got = it->first;
        cached -= it->first;
        blocks.erase(it);
        return p;
    }
    bool give(void *p, size_t block_bytes)
    {
        if (block_bytes < ((size_t)1 << 20)) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (closed || cached + block_bytes > cap_bytes()) return false;
        cached += block_bytes;
        if (forked > 0) parked.emplace_back(block_bytes, p);
        else blocks.emplace(block_bytes, p);
        return true;
    }
    void fork()
    {
        std::lock_guard<std::mutex> lk(mu);
        ++forked;
    }
    void join() // (call after the main stream waits for the side stream's last event)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (forked > 0 && --forked == 0) {
            for (auto &b : parked) blocks.emplace(b.first, b.second);
            parked.clear();
        }
    }
    void trim(bool close = false)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &b : blocks) (void)hipFree(b.second);
        for (auto &b : parked) (void)hipFree(b.second);
        blocks.clear();
        parked.clear();
        cached = 0;
        if (close) closed = true;
    }
    ~AllocMeter() { trim(true); }
};
// the meter of the handle this thread last entered (Context::use_device); weak: a handle destroyed on another thread
// leaves nothing behind to charge
extern thread_local std::weak_ptr<AllocMeter> tl_alloc_meter;
// every live handle's meter (round 6): an allocation that fails although ITS handle's cache is empty asks the other handles of
// the process for the blocks they keep -- with caches of up to 16 GiB per handle a second handle must not starve on what a
// first one merely keeps.  Behind a mutex; entries are weak and expire with their handles.
void register_alloc_meter(const std::shared_ptr<AllocMeter> &m);
void trim_all_alloc_meters();

// Owning device allocation.
template <typename T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    size_t block_bytes = 0; // size of the allocation behind ptr (a block from the handle's cache may be larger than asked for)
    std::shared_ptr<AllocMeter> meter; // shared: a buffer may outlive the handle that was current when it was allocated
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (ptr) {
            if (!(meter && meter->give(ptr, block_bytes))) (void)hipFree(ptr);
            if (meter) meter->add(-(long long)(count * sizeof(T)));
        }
        ptr = nullptr;
        count = 0;
        block_bytes = 0;
        meter.reset();
    }
    void swap(DeviceBuffer &o)
    {
        std::swap(ptr, o.ptr);
        std::swap(count, o.count);
        std::swap(block_bytes, o.block_bytes);
        std::swap(meter, o.meter);
    }
    // (re)allocate only when growing or when the size class changes a lot; contents undefined
    void ensure(size_t n)
    {
        if (n <= count && ptr) return;
        release();
        if (n == 0) n = 1;
        meter = tl_alloc_meter.lock();
        const size_t need = n * sizeof(T);
        void *p = meter ? meter->take(need, &block_bytes) : nullptr;
        if (p && meter->poison) { // tests: a recycled block arrives full of NaN bit patterns
            (void)hipDeviceSynchronize();
            (void)hipMemset(p, 0xFF, block_bytes);
            (void)hipDeviceSynchronize();
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, need);
            if (e != hipSuccess && meter) { // the cache may hold what this allocation needs
                (void)hipGetLastError();
                meter->trim();
                e = hipMalloc(&p, need);
            }
            if (e != hipSuccess) { // ... or the caches of the process's other handles
                (void)hipGetLastError();
                trim_all_alloc_meters();
                e = hipMalloc(&p, need);
            }
            if (e != hipSuccess) {
                meter.reset();
                PS_HIP_CHECK(e);
            }
            block_bytes = need;
            if (meter && meter->poison >= 2) { // tests: what the driver hands out is not zero either (it holds what other processes left)
                (void)hipDeviceSynchronize();
                (void)hipMemset(p, 0xFF, block_bytes);
                (void)hipDeviceSynchronize();
            }
        }
        ptr = (T *)p;
        count = n;
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
