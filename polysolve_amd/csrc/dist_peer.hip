// dist_peer.hip -- peer-mapped collectives for the in-process multi-device handle (round 4; no reference counterpart:
// SURVEY.md 8(e) -- the reference has no collective call sites at all).
//
// Why: strong-scaled over the 8 GPUs of a node a 256^3 system leaves ~57 us of kernels per PCG iteration, against one RCCL
// all-reduce of three doubles (~20 us) and a grouped send / recv of two halo planes.  Inside ONE process every device can
// map its peers' memory (hipDeviceEnablePeerAccess over xGMI), so the two per-iteration exchanges need no library:
//   * all-reduce of <= 8 doubles: every rank STORES its values into a slot of every peer's buffer, then a flag; one
//     workgroup waits for the W flags in its own memory and adds the W contributions in rank order -- every rank computes
//     the same bits, one launch, no ring;
//   * halo exchange: the sender's copy kernel writes its boundary entries straight into the receiver's staging buffer
//     (where the receiver expects them) and raises a flag; the receiver's kernel waits for its sources' flags and moves
//     the staging buffer behind its vector.
// Buffers that peers write and the owner polls are fine-grained allocations (hipDeviceMallocFinegrained: not cached in the
// owner's L2), stores are followed by __threadfence_system() and a system-scope release of the flag, polls are system-scope
// acquires -- the pattern RCCL's own LL protocol uses.  Flags carry the epoch of the call (monotone, never reset), slots
// and staging are double-buffered by its parity: a rank can be at most one call ahead of a peer it exchanges with.
// "dist_collectives" = 1 selects it ("rccl" = 0 stays the default: the north_star's design is what gets measured first);
// everything else (setup-time exchanges, the hierarchy's level halos) stays on RCCL.
// Same-device "peers" (repeated device ids: the test vehicle on one-GPU boxes) run the SAME kernels, host-synchronised
// between posting and collecting -- W kernels that wait for each other on one device could end up behind one another
// in a hardware queue.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <set>

#include "dist.hpp"

namespace psolve {

namespace {
constexpr int kPeerMax = 64;  // ranks
constexpr int kArSlot = 8;    // doubles per contribution
constexpr int kPostParts = 8; // workgroups per destination of a halo post
constexpr long long kSpinLimit = 400000000ll; // polls before a waiting kernel gives up (seconds, not minutes)
constexpr unsigned long long kWaitTicks = 1000000000ull; // ... and 10 s of wall_clock64 (100 MHz), whichever comes first

struct ArArgs {
    double *slots[kPeerMax];             // rank q's buffer: [2][W][kArSlot]
    unsigned long long *flags[kPeerMax]; // rank q's flags:  [2][W]
};

struct PostArgs {
    double *dst[kPeerMax];               // where destination d expects my entries (parity 0; parity 1 at + stride[d])
    long long stride[kPeerMax];          // doubles between the two parities of destination d's staging buffer
    unsigned long long *flag[kPeerMax];  // destination d's flag for me (parity 0; parity 1 at + W)
    long long count[kPeerMax], src_off[kPeerMax];
    int ndest, world;
};

struct CollectArgs {
    int src[kPeerMax];
    int nsrc, world;
};

__device__ __forceinline__ void store_flag(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_flag(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// spin until *p >= want; false on abort / time-out.  Two bounds: a poll count, and kWaitTicks of the constant 100 MHz
// wall_clock64 counter (the poll count alone stretches with the memory latency of a busy fabric).  Every 4096 polls the
// host's abort word and this rank's own `fail` word (pinned host memory, set by whichever waiting kernel gave up first)
// are read: after one time-out the kernels queued behind it return at once instead of each waiting its own ten seconds.
__device__ bool wait_flag(const unsigned long long *p, unsigned long long want, const volatile int *abort_host,
                          const volatile int *fail)
{
    if (load_flag(p) >= want) return true;
    if (*abort_host || *fail) return false;
    const unsigned long long t0 = wall_clock64();
    for (long long spin = 0; spin < kSpinLimit; ++spin) {
        if (load_flag(p) >= want) return true;
        if ((spin & 4095) == 4095) {
            if (*abort_host || *fail) return false;
            if (wall_clock64() - t0 > kWaitTicks) return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}
__device__ __forceinline__ void raise_fail(int *fail)
{
    __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void peer_ar_post_kernel(ArArgs T, int rank, int W, int par, unsigned long long epoch,
                                                           const double *__restrict__ buf, int count)
{
    const int q = threadIdx.x;
    if (q < W) {
        double *dst = T.slots[q] + ((size_t)par * W + rank) * kArSlot;
        for (int k = 0; k < count; ++k) __hip_atomic_store(dst + k, buf[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        store_flag(T.flags[q] + (size_t)par * W + rank, epoch);
    }
}

__global__ __launch_bounds__(64) void peer_ar_collect_kernel(const double *slots, const unsigned long long *flags, int W, int par,
                                                              unsigned long long epoch, double *__restrict__ buf, int count,
                                                              const volatile int *abort_host, int *fail)
{
    __shared__ int ok;
    const int q = threadIdx.x;
    if (q == 0) ok = 1;
    __syncthreads();
    if (q < W && !wait_flag(flags + (size_t)par * W + q, epoch, abort_host, fail)) ok = 0;
    __syncthreads();
    if (!ok) {
        if (q == 0) raise_fail(fail);
        return;
    }
    if (q < count) {
        double s = 0.0;
        for (int r = 0; r < W; ++r) // rank order: the same bits on every rank
            s += __hip_atomic_load(slots + ((size_t)par * W + r) * kArSlot + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[q] = s;
    }
}

__global__ __launch_bounds__(256) void peer_halo_post_kernel(PostArgs P, int rank, int par, unsigned long long epoch,
                                                              const double *__restrict__ send, int *counters)
{
    const int d = blockIdx.y;
    double *dst = P.dst[d] + (size_t)par * P.stride[d];
    const double *src = send + P.src_off[d];
    const long long n = P.count[d];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        __builtin_nontemporal_store(src[i], dst + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(&counters[d], 1);
        if (old == (int)gridDim.x - 1) { // the last part of this destination: everything is on its way
            counters[d] = 0;
            __threadfence_system();
            store_flag(P.flag[d] + (size_t)par * P.world + rank, epoch);
        }
    }
}

__global__ __launch_bounds__(256) void peer_halo_collect_kernel(CollectArgs C, const double *stage, const unsigned long long *flags,
                                                                 int par, unsigned long long epoch, double *__restrict__ recv,
                                                                 long long n_recv, long long stride,
                                                                 const volatile int *abort_host, int *fail)
{
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    if ((int)threadIdx.x < C.nsrc && !wait_flag(flags + (size_t)par * C.world + C.src[threadIdx.x], epoch, abort_host, fail))
        ok = 0;
    __syncthreads();
    if (!ok) {
        if (threadIdx.x == 0) raise_fail(fail);
        return;
    }
    const double *src = stage + (size_t)par * stride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_recv; i += (long long)gridDim.x * 256)
        recv[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace

struct PeerRank {
    int device = 0;
    double *ar_slots = nullptr;             // fine-grained
    unsigned long long *ar_flags = nullptr; // fine-grained
    double *stage = nullptr;                // fine-grained, 2 * stage_cap doubles
    long long stage_cap = 0;
    unsigned long long *hx_flags = nullptr; // fine-grained [2][W]
    int *counters = nullptr;                // device: per destination, parts done
    int *fail = nullptr;                    // pinned, mapped: a waiting kernel gave up (read by the host at its poll points)
    bool halo_ok_local = false;             // this rank's view of the published halo layout (prepare)
    // the halo plan this rank published (prepare): who sends me what, where
    std::vector<int64_t> recv_counts, recv_offsets, send_counts, send_offsets;
    int64_t n_recv = 0;
    unsigned long long ar_epoch = 0, hx_epoch = 0;
    bool halo_ready = false;
};

struct PeerGroup {
    int world = 0;
    bool same_device = false;
    std::vector<PeerRank> rk;
    int *abort_host = nullptr; // pinned, mapped: polled by waiting kernels
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        PS_REQUIRE(!aborted, PSOLVE_HIP_ECOMM, "peer group aborted: another shard failed");
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || aborted; });
            PS_REQUIRE(generation != gen, PSOLVE_HIP_ECOMM, "peer group aborted: another shard failed");
        }
    }
};

static void *finegrained(size_t bytes)
{
    void *p = nullptr;
    PS_HIP_CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    PS_HIP_CHECK(hipMemset(p, 0, bytes));
    return p;
}

PeerGroup *peer_group_create(const std::vector<int> &devices)
{
    const int W = (int)devices.size();
    if (W < 2 || W > kPeerMax) return nullptr;
    int cur = 0;
    (void)hipGetDevice(&cur);
    PeerGroup *g = new PeerGroup();
    g->world = W;
    g->rk.resize((size_t)W);
    g->same_device = std::set<int>(devices.begin(), devices.end()).size() < devices.size();
    try {
        // every device maps every other one
        for (int a = 0; a < W; ++a)
            for (int b = 0; b < W; ++b) {
                if (devices[a] == devices[b]) continue;
                int can = 0;
                PS_HIP_CHECK(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
                PS_REQUIRE(can, PSOLVE_HIP_ECOMM, "peer group: a device cannot map a peer's memory");
                PS_HIP_CHECK(hipSetDevice(devices[a]));
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) PS_HIP_CHECK(e);
                (void)hipGetLastError();
            }
        PS_HIP_CHECK(hipHostMalloc((void **)&g->abort_host, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
        *g->abort_host = 0;
        for (int r = 0; r < W; ++r) {
            PeerRank &R = g->rk[(size_t)r];
            R.device = devices[r];
            PS_HIP_CHECK(hipSetDevice(R.device));
            R.ar_slots = (double *)finegrained((size_t)2 * W * kArSlot * sizeof(double));
            R.ar_flags = (unsigned long long *)finegrained((size_t)2 * W * sizeof(unsigned long long));
            R.hx_flags = (unsigned long long *)finegrained((size_t)2 * W * sizeof(unsigned long long));
            PS_HIP_CHECK(hipMalloc((void **)&R.counters, (size_t)W * sizeof(int)));
            PS_HIP_CHECK(hipMemset(R.counters, 0, (size_t)W * sizeof(int)));
            PS_HIP_CHECK(hipHostMalloc((void **)&R.fail, sizeof(int), hipHostMallocMapped | hipHostMallocPortable));
            *R.fail = 0;
        }
        PS_HIP_CHECK(hipDeviceSynchronize());
    } catch (...) {
        (void)hipSetDevice(cur);
        peer_group_destroy(g);
        throw;
    }
    (void)hipSetDevice(cur);
    return g;
}

void peer_group_destroy(PeerGroup *g)
{
    if (!g) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (PeerRank &R : g->rk) {
        (void)hipSetDevice(R.device);
        if (R.ar_slots) (void)hipFree(R.ar_slots);
        if (R.ar_flags) (void)hipFree(R.ar_flags);
        if (R.hx_flags) (void)hipFree(R.hx_flags);
        if (R.stage) (void)hipFree(R.stage);
        if (R.counters) (void)hipFree(R.counters);
        if (R.fail) (void)hipHostFree(R.fail);
    }
    if (g->abort_host) (void)hipHostFree(g->abort_host);
    (void)hipSetDevice(cur);
    delete g;
}

void peer_group_abort(PeerGroup *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->m);
    g->aborted = true;
    if (g->abort_host) *reinterpret_cast<volatile int *>(g->abort_host) = 1;
    g->cv.notify_all();
}

bool peer_group_aborted(PeerGroup *g) { return g && g->aborted; }

// every shard's thread has been joined and its streams are idle: epochs, flags and counters start over
void peer_group_reset(PeerGroup *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->m);
    if (!g->aborted) {
        g->arrived = 0;
        return;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    const size_t W = (size_t)g->world;
    for (PeerRank &R : g->rk) {
        (void)hipSetDevice(R.device);
        (void)hipDeviceSynchronize();
        (void)hipMemset(R.ar_flags, 0, 2 * W * sizeof(unsigned long long));
        (void)hipMemset(R.hx_flags, 0, 2 * W * sizeof(unsigned long long));
        (void)hipMemset(R.counters, 0, W * sizeof(int));
        *R.fail = 0;
        R.ar_epoch = R.hx_epoch = 0;
        R.halo_ready = false;
    }
    (void)hipSetDevice(cur);
    *g->abort_host = 0;
    g->aborted = false;
    g->arrived = 0;
}

// ---------------------------------------------------------------------------------------------
void Comm::attach_peer(PeerGroup *g, int rank)
{
    peer_ = g;
    peer_rank_ = rank;
}

bool Comm::peer_halo_ready() const { return peer_ && peer_->rk[(size_t)peer_rank_].halo_ready; }

static void check_fail(PeerGroup *g, PeerRank &R, hipStream_t s, const char *what)
{
    // (where the host synchronises anyway: the rehearsal on one device, the end of a solve)
    PS_HIP_CHECK(hipStreamSynchronize(s));
    if (*reinterpret_cast<volatile int *>(R.fail)) {
        peer_group_abort(g);
        throw Error(PSOLVE_HIP_ECOMM, std::string("peer ") + what + ": a peer never arrived");
    }
}

void Comm::peer_check(hipStream_t s)
{
    if (!peer_) return;
    check_fail(peer_, peer_->rk[(size_t)peer_rank_], s, "collective");
}

// the PCG loop's poll points: no synchronisation, one read of pinned host memory -- a time-out on this rank ends the solve
// at the next poll (and raises the abort word, which ends the other ranks' waits) instead of at its last iteration
void Comm::peer_poll()
{
    if (!peer_) return;
    PeerRank &R = peer_->rk[(size_t)peer_rank_];
    if (*reinterpret_cast<volatile int *>(R.fail)) {
        peer_group_abort(peer_);
        throw Error(PSOLVE_HIP_ECOMM, "peer collective: a peer never arrived");
    }
    PS_REQUIRE(!peer_->aborted, PSOLVE_HIP_ECOMM, "peer group aborted: another shard failed");
}

void Comm::peer_allreduce(double *d_buf, int count, hipStream_t s)
{
    PeerGroup *g = peer_;
    PS_REQUIRE(g && !g->aborted, PSOLVE_HIP_ECOMM, "peer group aborted: another shard failed");
    PS_REQUIRE(count >= 1 && count <= kArSlot, PSOLVE_HIP_EINVAL, "peer all-reduce: at most 8 values");
    PeerRank &R = g->rk[(size_t)peer_rank_];
    const unsigned long long epoch = ++R.ar_epoch;
    const int par = (int)(epoch & 1), W = g->world;
    ArArgs T;
    for (int q = 0; q < W; ++q) {
        T.slots[q] = g->rk[(size_t)q].ar_slots;
        T.flags[q] = g->rk[(size_t)q].ar_flags;
    }
    hipLaunchKernelGGL(peer_ar_post_kernel, dim3(1), dim3(64), 0, s, T, peer_rank_, W, par, epoch, d_buf, count);
    if (g->same_device) {
        PS_HIP_CHECK(hipStreamSynchronize(s));
        g->barrier();
    }
    hipLaunchKernelGGL(peer_ar_collect_kernel, dim3(1), dim3(64), 0, s, R.ar_slots, R.ar_flags, W, par, epoch, d_buf, count,
                       g->abort_host, R.fail);
    PS_HIP_CHECK(hipGetLastError());
    if (g->same_device) check_fail(g, R, s, "all-reduce");
}

// collective, host-synchronised (factorize time): every rank publishes where it expects whose entries and (re)allocates its
// staging buffer; afterwards a sender knows every destination's address
void Comm::peer_prepare_halo(const HaloPlan &plan, hipStream_t s)
{
    PeerGroup *g = peer_;
    if (!g) return;
    PeerRank &R = g->rk[(size_t)peer_rank_];
    PS_HIP_CHECK(hipStreamSynchronize(s));
    g->barrier(); // nobody is inside an exchange that uses the old staging buffers
    R.halo_ready = false;
    R.recv_counts = plan.recv_counts;
    R.recv_offsets = plan.recv_offsets;
    R.send_counts = plan.send_counts;
    R.send_offsets = plan.send_offsets;
    R.n_recv = 0;
    for (int64_t c : plan.recv_counts) R.n_recv += c;
    if (R.n_recv > R.stage_cap) {
        if (R.stage) PS_HIP_CHECK(hipFree(R.stage));
        R.stage = nullptr;
        R.stage_cap = R.n_recv + R.n_recv / 8 + 64;
        R.stage = (double *)finegrained((size_t)2 * R.stage_cap * sizeof(double));
    }
    g->barrier(); // every rank's layout and address are published
    bool ok = (int)R.send_counts.size() == g->world && (int)R.recv_counts.size() == g->world;
    for (int q = 0; ok && q < g->world; ++q) {
        const PeerRank &Q = g->rk[(size_t)q];
        ok = (int)Q.recv_counts.size() == g->world && Q.recv_counts[(size_t)peer_rank_] == R.send_counts[(size_t)q];
        // The two staging parities are safe only between mutual neighbours: rank a may overwrite parity e+2 in b's buffer
        // because it has collected b's epoch e+1, which b posted after collecting a's epoch e.  A one-way neighbour has no
        // such back-pressure, so an unsymmetric halo graph (a pattern that is not structurally symmetric) keeps to RCCL.
        if (ok && q != peer_rank_) ok = (R.send_counts[(size_t)q] > 0) == (R.recv_counts[(size_t)q] > 0);
    }
    R.halo_ok_local = ok;
    g->barrier(); // every rank's verdict is published: the peer path is taken by all ranks or by none
    for (int q = 0; q < g->world; ++q) ok = ok && g->rk[(size_t)q].halo_ok_local;
    R.halo_ready = ok;
    g->barrier();
}

void Comm::peer_exchange_halo(const double *d_send, double *d_recv, hipStream_t s)
{
    PeerGroup *g = peer_;
    PS_REQUIRE(g && !g->aborted, PSOLVE_HIP_ECOMM, "peer group aborted: another shard failed");
    PeerRank &R = g->rk[(size_t)peer_rank_];
    PS_REQUIRE(R.halo_ready, PSOLVE_HIP_EINVAL, "peer halo exchange before peer_prepare_halo");
    const unsigned long long epoch = ++R.hx_epoch;
    const int par = (int)(epoch & 1), W = g->world;
    PostArgs P;
    P.ndest = 0;
    P.world = W;
    CollectArgs C;
    C.nsrc = 0;
    C.world = W;
    for (int q = 0; q < W; ++q) {
        if (q == peer_rank_) continue;
        if (R.send_counts[(size_t)q] > 0) {
            const PeerRank &Q = g->rk[(size_t)q];
            const int d = P.ndest++;
            P.dst[d] = Q.stage + Q.recv_offsets[(size_t)peer_rank_];
            P.stride[d] = Q.stage_cap;
            P.flag[d] = Q.hx_flags;
            P.count[d] = R.send_counts[(size_t)q];
            P.src_off[d] = R.send_offsets[(size_t)q];
        }
        if (R.recv_counts[(size_t)q] > 0) C.src[C.nsrc++] = q;
    }
    if (P.ndest > 0)
        hipLaunchKernelGGL(peer_halo_post_kernel, dim3(kPostParts, P.ndest), dim3(256), 0, s, P, peer_rank_, par, epoch, d_send,
                           R.counters);
    if (g->same_device) {
        PS_HIP_CHECK(hipStreamSynchronize(s));
        g->barrier();
    }
    if (C.nsrc > 0) {
        const int blocks = (int)std::max<long long>(1, std::min<long long>(64, (R.n_recv + 4095) / 4096));
        hipLaunchKernelGGL(peer_halo_collect_kernel, dim3(blocks), dim3(256), 0, s, C, R.stage, R.hx_flags, par, epoch, d_recv,
                           (long long)R.n_recv, (long long)R.stage_cap, g->abort_host, R.fail);
    }
    PS_HIP_CHECK(hipGetLastError());
    if (g->same_device) {
        check_fail(g, R, s, "halo exchange");
        g->barrier(); // (rehearsal only: the staging parity is free again before anybody posts two calls ahead)
    }
}

} // namespace psolve
