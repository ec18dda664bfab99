// dist.hpp -- 1-D row-partitioned operation: RCCL (loaded lazily, never linked) + halo plan.
//
// The reference has no distributed path at all (zero NCCL/MPI call sites on this
// path); this is new design.  One process per GPU; the CG scalars are ncclAllReduce'd and the halo
// entries of the SpMV input vector travel by grouped ncclSend/ncclRecv between the ranks that
// actually share matrix columns (for slab partitions: the two neighbours, one xGMI link each).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdint>
#include <string>
#include <vector>

#include "common.hpp"

namespace psolve {

struct RcclApi; // function table, resolved with dlopen/dlsym

struct HaloPlan {
    int rank = 0, world = 1;
    std::vector<int64_t> row_offsets; // world + 1
    std::vector<int32_t> halo;        // sorted unique global column ids owned by other ranks
    std::vector<int64_t> recv_counts; // per owner rank
    std::vector<int64_t> recv_offsets;
    std::vector<int64_t> send_counts; // per destination rank
    std::vector<int64_t> send_offsets;
    int64_t n_send = 0;
};

// Host-only.  cols: global column ids (any order, duplicates allowed, in-range ids ignored).
void plan_halo(int rank, int world, const int64_t *row_offsets, int64_t n_cols, const int32_t *cols,
               std::vector<int32_t> &halo, std::vector<int64_t> &recv_counts);

// In-process loopback group: N handles of ONE process (one thread each, possibly all on the same GPU)
// exchange through host-synchronised device copies.  It exists so that the distributed code path
// (halo plan, column remap, pack/exchange, all-reduced CG scalars) can be run on real kernels on a box
// with fewer GPUs than ranks -- RCCL refuses two ranks on one device.  Not a performance path.
struct LocalGroup;
LocalGroup *local_group_create(int world);
void local_group_destroy(LocalGroup *g);
// A rank that fails in the middle of a collective sequence wakes the ranks blocked in the group's barrier
// (they throw PSOLVE_HIP_ECOMM); reset re-arms the group once every rank's thread has been joined.
void local_group_abort(LocalGroup *g);
void local_group_reset(LocalGroup *g);

// Peer-mapped collectives of an in-process multi-device handle (dist_peer.hip): all-reduce of <= 8 doubles and the halo
// exchange by stores into the peers' memory + epoch flags.  nullptr where a device cannot map a peer.
struct PeerGroup;
PeerGroup *peer_group_create(const std::vector<int> &devices);
void peer_group_destroy(PeerGroup *g);
void peer_group_abort(PeerGroup *g);   // waiting kernels and host barriers give up
bool peer_group_aborted(PeerGroup *g);
void peer_group_reset(PeerGroup *g);   // (every shard's thread joined, streams idle)

class Comm {
public:
    Comm() = default;
    ~Comm();
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;

    static void unique_id(char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path);
    void init(int rank, int world, const char id[PSOLVE_HIP_UNIQUE_ID_BYTES], const char *rccl_path);
    void init_local(LocalGroup *g, int rank);
    // one clique inside ONE process (ncclCommInitAll): comms[r] becomes rank r on devices[r]
    static void init_all(const std::vector<Comm *> &comms, const std::vector<int> &devices, const char *rccl_path);
    void abort(); // RCCL cliques: give up every operation in flight; the communicators are gone afterwards
    bool aborted() const { return dead_; }
    bool active() const { return comm_ != nullptr || local_ != nullptr; }
    bool is_rccl() const { return comm_ != nullptr; } // a real RCCL communicator (not the in-process loopback group)
    int rank() const { return rank_; }
    int world() const { return world_; }

    // peer-mapped per-iteration collectives ("dist_collectives" 1) on top of the transport above
    void attach_peer(PeerGroup *g, int rank);
    bool peer_attached() const { return peer_ != nullptr; }
    void set_peer_collectives(bool on) { use_peer_ = on; }
    bool peer_on() const { return peer_ != nullptr && use_peer_; }
    bool peer_halo_ready() const;
    void peer_prepare_halo(const HaloPlan &plan, hipStream_t s);                   // collective, at factorize
    void peer_exchange_halo(const double *d_send, double *d_recv, hipStream_t s);  // the prepared plan's exchange
    void peer_allreduce(double *d_buf, int count, hipStream_t s);
    // after a solve has been synchronised: did a waiting kernel of this rank give up (a peer never arrived)?  Throws ECOMM.
    void peer_check(hipStream_t s);
    void peer_poll(); // no synchronisation: throws ECOMM when a waiting kernel of this rank has given up

    void allreduce_sum(double *d_buf, int count, hipStream_t s);
    void allgather_i64(const int64_t *d_send, int64_t *d_recv, int count_per_rank, hipStream_t s);
    // grouped point-to-point: for every peer q, send send_counts[q] elements starting at
    // d_send + send_offsets[q] and receive recv_counts[q] at d_recv + recv_offsets[q]
    void exchange_f64(const double *d_send, const std::vector<int64_t> &send_counts,
                      const std::vector<int64_t> &send_offsets, double *d_recv,
                      const std::vector<int64_t> &recv_counts, const std::vector<int64_t> &recv_offsets,
                      hipStream_t s);
    void exchange_i32(const int32_t *d_send, const std::vector<int64_t> &send_counts,
                      const std::vector<int64_t> &send_offsets, int32_t *d_recv,
                      const std::vector<int64_t> &recv_counts, const std::vector<int64_t> &recv_offsets,
                      hipStream_t s);

private:
    // abort() frees the communicators from ANOTHER thread than the one that enqueues on them: every enqueue holds this
    // mutex from its check of dead_ to the return of the RCCL call, and abort() takes it before it frees anything -- with
    // a time limit, because a thread that is stuck INSIDE an RCCL call (what abort exists to end) never releases it
    std::timed_mutex mu_;
    void *comm_ = nullptr;     // reductions, all-gathers (main stream)
    std::atomic<bool> dead_{false}; // abort() was called: RCCL has freed both communicators
    void *comm_p2p_ = nullptr; // grouped send / recv: halo exchange (comm stream), setup-time row exchanges
    LocalGroup *local_ = nullptr;
    PeerGroup *peer_ = nullptr;
    int peer_rank_ = 0;
    bool use_peer_ = false;
    int rank_ = 0, world_ = 1;
};

} // namespace psolve
