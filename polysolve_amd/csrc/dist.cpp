// dist.cpp -- RCCL binding (dlopen, so a single-GPU user never loads RCCL and a torch process
// shares torch's already-loaded librccl.so.1) and the host-side halo plan.
#include "dist.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

namespace psolve {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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
    PS_SYM(CommDestroy);
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
}

Comm::~Comm()
{
    if (comm_ && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)comm_);
    comm_ = nullptr;
}

void Comm::allreduce_sum(double *d_buf, int count, hipStream_t s)
{
    PS_NCCL_CHECK(g_rccl.AllReduce(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)comm_, s));
}

void Comm::allgather_i64(const int64_t *d_send, int64_t *d_recv, int count_per_rank, hipStream_t s)
{
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
    exchange<double>(comm_, ncclFloat64, d_send, sc, so, d_recv, rc, ro, rank_, world_, s);
}

void Comm::exchange_i32(const int32_t *d_send, const std::vector<int64_t> &sc, const std::vector<int64_t> &so,
                        int32_t *d_recv, const std::vector<int64_t> &rc, const std::vector<int64_t> &ro,
                        hipStream_t s)
{
    exchange<int32_t>(comm_, ncclInt32, d_send, sc, so, d_recv, rc, ro, rank_, world_, s);
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
