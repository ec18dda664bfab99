// multi.hpp -- one Solver object, several GPUs of one node, ONE process.
//
// polysolve::linear::Solver::create("HIP") lives inside the caller's process (PolyFEM, Newton:
// Solver.hpp:31-132), exactly like the reference's in-tree GPU backend, whose pimpl owns its device
// resources (MASSolver.cu:186-196).  For a matrix that is to be partitioned over the GPUs of the node
// the handle therefore owns one Context per device, each driven by its own host thread, joined by an
// in-process RCCL clique (ncclCommInitAll) -- or by the loopback group when device ids repeat (several
// shards on one GPU: how the path is tested on a one-GPU box).  The host contract is unchanged:
// factorize(A) splits the rows itself (contiguous ranges balanced by nonzeros), solve(b, x) scatters
// b / x and gathers x.  SURVEY.md 8(b), 8(e).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "solver.hpp"

namespace psolve {

// contiguous row ranges with about nnz / world stored entries each, cut at multiples of `align` rows (block_size:
// a 3x3 block row never straddles two shards); host-only, exported as psolve_hip_partition_rows for the CPU tests
void partition_rows_by_nnz(int64_t n, const int32_t *outer, int world, int64_t align, std::vector<int64_t> &offsets);

class MultiContext {
public:
    MultiContext(const int *device_ids, int n_devices);
    ~MultiContext();
    MultiContext(const MultiContext &) = delete;
    MultiContext &operator=(const MultiContext &) = delete;

    int world() const { return (int)shards_.size(); }
    Context &shard(int r) { return *shards_[(size_t)r]; }
    bool loopback() const { return group_ != nullptr; }
    const std::vector<int64_t> &row_offsets() const { return row_offsets_; }

    void set_param(const std::string &key, double v);
    double get_param(const std::string &key) const;
    void synchronize();
    void trim(); // every shard's cached device blocks back to the driver (psolve_hip_trim)
    void analyze_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int precond_num);
    void factorize_host(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, const double *values);
    void solve_host(const double *b, double *x);

    psolve_hip_info info{};
    std::string last_error;
    // "reorder" on several devices: is the partitioned system renumbered, and the permutation (host copy)
    bool reordered() const { return reordered_; }
    const std::vector<int32_t> &new_of_old() const { return new_of_old_; }

private:
    // f(rank, shard) on one host thread per shard; the first failure (lowest rank) is rethrown
    void run_all(const std::function<void(int, Context &)> &f);
    void abort_all(); // a shard left the collective sequence on its own: free the ranks blocked in it
    void partition_rows(int64_t n, const int32_t *outer);

    std::vector<std::unique_ptr<Context>> shards_;
    std::vector<int> devices_;
    LocalGroup *group_ = nullptr;
    PeerGroup *peer_ = nullptr; // peer-mapped per-iteration collectives ("dist_collectives" 1); nullptr: the devices cannot map each other
    std::vector<int64_t> row_offsets_;
    int64_t n_ = -1;
    bool factorized_ = false;

    // "reorder": the system is renumbered BEFORE it is partitioned -- contiguous row ranges of a breadth-first order are
    // slabs of the mesh, so a shard talks to its two neighbours only, whatever the caller's numbering was (in the
    // caller's own numbering a scattered mesh makes almost every row a boundary row and the halo the whole vector)
    bool decide_order(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner);
    bool reordered_ = false;
    std::vector<int32_t> order_, new_of_old_;
    uint64_t order_version_ = 0;
    uint64_t ro_hash_ = 0;
    int64_t ro_n_ = -1, ro_nnz_ = -1;
    int ro_block_ = 1, ro_mode_ = 0, ro_reverse_ = -1;
    int64_t ro_searches_ = 0; // Cuthill-McKee searches of this handle ("stats.reorder_searches")
    double ro_min_spread_ = 0.0;
    bool ro_decision_ = false;
    ReorderInfo ro_info_;
    double ro_spread_before_ = 0.0, ro_spread_after_ = 0.0, ro_seconds_ = 0.0;
};

} // namespace psolve
