// dist.cpp -- RCCL binding (dlopen, so a single-GPU user never loads RCCL and a torch process
// shares torch's already-loaded librccl.so.1) and the host-side halo plan.
#include "dist.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace psolve {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static RcclApi g_rccl;

static RcclApi &rccl(const char *path)
{
    if (g_rccl.lib) return g_rccl;
    void *lib = nullptr;
    if (path && *path) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    // a copy that is already mapped (torch.distributed's) wins over the system one
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    PS_REQUIRE(lib, PSOLVE_HIP_ECOMM, std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"));
    RcclApi a;
    a.lib = lib;
#define PS_SYM(name)                                                                          \
    a.name = (decltype(a.name))dlsym(lib, "nccl" #name);                                      \
    PS_REQUIRE(a.name, PSOLVE_HIP_ECOMM, "librccl: missing symbol nccl" #name)
    PS_SYM(GetUniqueId);
    PS_SYM(CommInitRank);
    PS_SYM(CommInitAll);
    PS_SYM(CommDestroy);
    PS_SYM(CommAbort);
    PS_SYM(Broadcast);
    PS_SYM(AllReduce);
    PS_SYM(AllGather);
    PS_SYM(Send);
    PS_SYM(Recv);
    PS_SYM(GroupStart);
    PS_SYM(GroupEnd);
    PS_SYM(GetErrorString);
#undef PS_SYM
    g_rccl = a;
    return g_rccl;
}

#define PS_NCCL_CHECK(expr)                                                                   \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess)                                                                \
            throw Error(PSOLVE_HIP_ECOMM, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// LocalGroup: host-synchronised loopback transport (tests)
// ---------------------------------------------------------------------------------------------
struct LocalGroup {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    std::vector<const void *> send_ptr;
    std::vector<const int64_t *> send_counts, send_offsets;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        PS_REQUIRE(!aborted, PSOLVE_HIP_ECOMM, "loopback group aborted: another shard failed");
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || aborted; });
            PS_REQUIRE(generation != gen, PSOLVE_HIP_ECOMM, "loopback group aborted: another shard failed");
        }
    }
};

LocalGroup *local_group_create(int world)
{
    PS_REQUIRE(world >= 1 && world <= 64, PSOLVE_HIP_EINVAL, "local group: bad world size");
    LocalGroup *g = new LocalGroup();
    g->world = world;
    g->send_ptr.assign((size_t)world, nullptr);
    g->send_counts.assign((size_t)world, nullptr);
    g->send_offsets.assign((size_t)world, nullptr);
    return g;
}

void local_group_destroy(LocalGroup *g) { delete g; }

void local_group_abort(LocalGroup *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->m);
    g->aborted = true;
    g->cv.notify_all();
}

void local_group_reset(LocalGroup *g)
{
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->m);
    g->aborted = false;
    g->arrived = 0;
}

void Comm::init_local(LocalGroup *g, int rank)
{
    PS_REQUIRE(g && rank >= 0 && rank < g->world, PSOLVE_HIP_EINVAL, "comm_init_local: bad group/rank");
    local_ = g;
    rank_ = rank;
    world_ = g->world;
}

static void local_allreduce(LocalGroup *g, int rank, double *d_buf, int count, hipStream_t s)
{
    PS_HIP_CHECK(hipStreamSynchronize(s));
    g->send_ptr[(size_t)rank] = d_buf;
    g->barrier();
    std::vector<double> acc((size_t)count, 0.0), tmp((size_t)count);
    for (int q = 0; q < g->world; ++q) { // rank order: every rank computes the same bits
        PS_HIP_CHECK(hipMemcpyAsync(tmp.data(), g->send_ptr[(size_t)q], (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        for (int k = 0; k < count; ++k) acc[k] += tmp[k];
    }
    g->barrier(); // everyone has read every buffer
    PS_HIP_CHECK(hipMemcpyAsync(d_buf, acc.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
}

template <typename T>
static void local_exchange(LocalGroup *g, int rank, const T *d_send, const std::vector<int64_t> &sc,
                           const std::vector<int64_t> &so, T *d_recv, const std::vector<int64_t> &rc,
                           const std::vector<int64_t> &ro, hipStream_t s)
{
    PS_HIP_CHECK(hipStreamSynchronize(s));
    g->send_ptr[(size_t)rank] = d_send;
    g->send_counts[(size_t)rank] = sc.data();
    g->send_offsets[(size_t)rank] = so.data();
    g->barrier();
    for (int q = 0; q < g->world; ++q) {
        if (q == rank || rc[(size_t)q] <= 0) continue;
        PS_REQUIRE(g->send_counts[(size_t)q][rank] == rc[(size_t)q], PSOLVE_HIP_ECOMM,
                   "local exchange: send/recv counts of a pair of ranks disagree");
        const T *src = (const T *)g->send_ptr[(size_t)q] + g->send_offsets[(size_t)q][rank];
        PS_HIP_CHECK(hipMemcpyAsync(d_recv + ro[(size_t)q], src, (size_t)rc[(size_t)q] * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    // a device-to-device hipMemcpy need not be complete when it returns: finish on OUR stream before
    // anybody may overwrite a send buffer or consume the halo
    PS_HIP_CHECK(hipStreamSynchronize(s));
    g->barrier();
}

void Comm::unique_id(char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path)
{
    static_assert(sizeof(ncclUniqueId) == PSOLVE_HIP_UNIQUE_ID_BYTES, "unique id size");
    RcclApi &R = rccl(rccl_path);
    ncclUniqueId uid;
    PS_NCCL_CHECK(R.GetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
}

void Comm::init(int rank, int world, const char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path)
{
    PS_REQUIRE(world >= 1 && rank >= 0 && rank < world, PSOLVE_HIP_EINVAL, "comm_init: bad rank/world");
    RcclApi &R = rccl(rccl_path);
    if (dead_) { // aborted communicators are already freed
        comm_ = comm_p2p_ = nullptr;
        dead_ = false;
    }
    if (comm_) {
        R.CommDestroy((ncclComm_t)comm_);
        comm_ = nullptr;
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    PS_NCCL_CHECK(R.CommInitRank(&c, world, uid, rank));
    comm_ = c;
    rank_ = rank;
    world_ = world;
    // A second communicator for the point-to-point traffic (halo exchange on the comm stream), so that it never
    // shares a communicator -- whose operations RCCL serialises in issue order -- with the reductions of the main
    // stream.  Its id is made by rank 0 and travels over the first communicator.
    if (comm_p2p_) {
        R.CommDestroy((ncclComm_t)comm_p2p_);
        comm_p2p_ = nullptr;
    }
    if (world > 1) {
        ncclUniqueId uid2;
        std::memset(&uid2, 0, sizeof(uid2));
        if (rank == 0) PS_NCCL_CHECK(R.GetUniqueId(&uid2));
        void *d_id = nullptr;
        hipStream_t st = nullptr;
        PS_HIP_CHECK(hipMalloc(&d_id, sizeof(uid2)));
        try {
            PS_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            PS_HIP_CHECK(hipMemcpyAsync(d_id, &uid2, sizeof(uid2), hipMemcpyHostToDevice, st));
            PS_NCCL_CHECK(R.Broadcast(d_id, d_id, sizeof(uid2), ncclChar, 0, (ncclComm_t)comm_, st));
            PS_HIP_CHECK(hipMemcpyAsync(&uid2, d_id, sizeof(uid2), hipMemcpyDeviceToHost, st));
            PS_HIP_CHECK(hipStreamSynchronize(st));
        } catch (...) {
            if (st) (void)hipStreamDestroy(st);
            (void)hipFree(d_id);
            throw;
        }
        (void)hipStreamDestroy(st);
        (void)hipFree(d_id);
        ncclComm_t c2 = nullptr;
        PS_NCCL_CHECK(R.CommInitRank(&c2, world, uid2, rank));
        comm_p2p_ = c2;
    }
}

void Comm::init_all(const std::vector<Comm *> &comms, const std::vector<int> &devices, const char *rccl_path)
{
    const int world = (int)comms.size();
    PS_REQUIRE(world >= 1 && (int)devices.size() == world, PSOLVE_HIP_EINVAL, "comm init_all: bad device list");
    RcclApi &R = rccl(rccl_path);
    std::vector<ncclComm_t> c((size_t)world, nullptr), c2((size_t)world, nullptr);
    PS_NCCL_CHECK(R.CommInitAll(c.data(), world, devices.data()));
    PS_NCCL_CHECK(R.CommInitAll(c2.data(), world, devices.data())); // the point-to-point clique (see Comm::init)
    for (int r = 0; r < world; ++r) {
        Comm &m = *comms[(size_t)r];
        if (!m.dead_) {
            if (m.comm_) R.CommDestroy((ncclComm_t)m.comm_);
            if (m.comm_p2p_) R.CommDestroy((ncclComm_t)m.comm_p2p_);
        }
        m.dead_ = false;
        m.comm_ = c[(size_t)r];
        m.comm_p2p_ = c2[(size_t)r];
        m.local_ = nullptr;
        m.rank_ = r;
        m.world_ = world;
    }
}

// A rank of an in-process clique has failed outside a collective: the others are (or will be) blocked inside one.
// ncclCommAbort marks the communicator so that its kernels in flight give up; the handle is dead afterwards and the
// clique has to be created again (MultiContext::run_all does).
void Comm::abort()
{
    if (local_ || dead_.exchange(true)) return;
    // The owning thread checks dead_ and enqueues under mu_, so once we hold it no enqueue can be between its check
    // and its use of the pointers we are about to free.  If the owner does not let go within two seconds it sits inside
    // an RCCL call that waits for a peer that will never come: ending that call is what ncclCommAbort is for.
    // (The pointers stay where they are; init_all / the destructor know that an aborted communicator is already freed.)
    std::unique_lock<std::timed_mutex> lk(mu_, std::defer_lock);
    (void)lk.try_lock_for(std::chrono::seconds(2));
    if (g_rccl.CommAbort) {
        if (comm_p2p_) (void)g_rccl.CommAbort((ncclComm_t)comm_p2p_);
        if (comm_) (void)g_rccl.CommAbort((ncclComm_t)comm_);
    }
}

Comm::~Comm()
{
    if (!dead_) {
        if (comm_p2p_ && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)comm_p2p_);
        if (comm_ && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)comm_);
    }
    comm_p2p_ = nullptr;
    comm_ = nullptr;
}

void Comm::allreduce_sum(double *d_buf, int count, hipStream_t s)
{
    if (peer_on() && count <= 8 && world_ > 1) {
        peer_allreduce(d_buf, count, s);
        return;
    }
    if (local_) {
        local_allreduce(local_, rank_, d_buf, count, s);
        return;
    }
    std::lock_guard<std::timed_mutex> lk(mu_);
    PS_REQUIRE(!dead_, PSOLVE_HIP_ECOMM, "communicator aborted: another shard failed");
    PS_NCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)comm_, s));
}

void Comm::allgather_i64(const int64_t *d_send, int64_t *d_recv, int count_per_rank, hipStream_t s)
{
    if (local_) {
        PS_HIP_CHECK(hipStreamSynchronize(s));
        local_->send_ptr[(size_t)rank_] = d_send;
        local_->barrier();
        for (int q = 0; q < world_; ++q)
            PS_HIP_CHECK(hipMemcpyAsync(d_recv + (size_t)q * count_per_rank, local_->send_ptr[(size_t)q],
                                        (size_t)count_per_rank * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        local_->barrier();
        return;
    }
    std::lock_guard<std::timed_mutex> lk(mu_);
    PS_REQUIRE(!dead_, PSOLVE_HIP_ECOMM, "communicator aborted: another shard failed");
    PS_NCCL_CHECK(g_rccl.AllGather(d_send, d_recv, (size_t)count_per_rank, ncclInt64, (ncclComm_t)comm_, s));
}

template <typename T>
static void exchange(void *comm, ncclDataType_t dt, const T *d_send, const std::vector<int64_t> &sc,
                     const std::vector<int64_t> &so, T *d_recv, const std::vector<int64_t> &rc,
                     const std::vector<int64_t> &ro, int rank, int world, hipStream_t s)
{
    bool any = false;
    for (int q = 0; q < world; ++q) any = any || (q != rank && (sc[q] > 0 || rc[q] > 0));
    if (!any) return;
    PS_NCCL_CHECK(g_rccl.GroupStart());
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        if (sc[q] > 0) PS_NCCL_CHECK(g_rccl.Send(d_send + so[q], (size_t)sc[q], dt, q, (ncclComm_t)comm, s));
        if (rc[q] > 0) PS_NCCL_CHECK(g_rccl.Recv(d_recv + ro[q], (size_t)rc[q], dt, q, (ncclComm_t)comm, s));
    }
    PS_NCCL_CHECK(g_rccl.GroupEnd());
}

void Comm::exchange_f64(const double *d_send, const std::vector<int64_t> &sc, const std::vector<int64_t> &so,
                        double *d_recv, const std::vector<int64_t> &rc, const std::vector<int64_t> &ro,
                        hipStream_t s)
{
    if (local_) {
        local_exchange<double>(local_, rank_, d_send, sc, so, d_recv, rc, ro, s);
        return;
    }
    std::lock_guard<std::timed_mutex> lk(mu_);
    PS_REQUIRE(!dead_, PSOLVE_HIP_ECOMM, "communicator aborted: another shard failed");
    exchange<double>(comm_p2p_ ? comm_p2p_ : comm_, ncclFloat64, d_send, sc, so, d_recv, rc, ro, rank_, world_, s);
}

void Comm::exchange_i32(const int32_t *d_send, const std::vector<int64_t> &sc, const std::vector<int64_t> &so,
                        int32_t *d_recv, const std::vector<int64_t> &rc, const std::vector<int64_t> &ro,
                        hipStream_t s)
{
    if (local_) {
        local_exchange<int32_t>(local_, rank_, d_send, sc, so, d_recv, rc, ro, s);
        return;
    }
    std::lock_guard<std::timed_mutex> lk(mu_);
    PS_REQUIRE(!dead_, PSOLVE_HIP_ECOMM, "communicator aborted: another shard failed");
    exchange<int32_t>(comm_p2p_ ? comm_p2p_ : comm_, ncclInt32, d_send, sc, so, d_recv, rc, ro, rank_, world_, s);
}


// ---------------------------------------------------------------------------------------------
void plan_halo(int rank, int world, const int64_t *row_offsets, int64_t n_cols, const int32_t *cols,
               std::vector<int32_t> &halo, std::vector<int64_t> &recv_counts)
{
    PS_REQUIRE(world >= 1 && rank >= 0 && rank < world, PSOLVE_HIP_EINVAL, "plan_halo: bad rank/world");
    for (int q = 0; q < world; ++q)
        PS_REQUIRE(row_offsets[q] <= row_offsets[q + 1], PSOLVE_HIP_EINVAL, "plan_halo: row_offsets not monotone");
    const int64_t lo = row_offsets[rank], hi = row_offsets[rank + 1], n_global = row_offsets[world];
    halo.clear();
    for (int64_t i = 0; i < n_cols; ++i) {
        const int64_t c = cols[i];
        PS_REQUIRE(c >= 0 && c < n_global, PSOLVE_HIP_EINVAL, "plan_halo: column id outside the global matrix");
        if (c < lo || c >= hi) halo.push_back((int32_t)c);
    }
    std::sort(halo.begin(), halo.end());
    halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
    recv_counts.assign((size_t)world, 0);
    int q = 0;
    for (int32_t c : halo) {
        while (c >= row_offsets[q + 1]) ++q; // halo is sorted, owners are monotone
        ++recv_counts[(size_t)q];
    }
}

} // namespace psolve
