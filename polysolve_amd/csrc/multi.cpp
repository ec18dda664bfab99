// multi.cpp -- the in-process multi-device handle (see multi.hpp).
#include "multi.hpp"

#include "host_hash.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <set>
#include <thread>

namespace psolve {

MultiContext::MultiContext(const int *device_ids, int n_devices)
{
    PS_REQUIRE(device_ids && n_devices >= 2 && n_devices <= 64, PSOLVE_HIP_EINVAL,
               "create_multi: need 2..64 device ids (one id: psolve_hip_create)");
    devices_.assign(device_ids, device_ids + n_devices);
    for (int d : devices_) shards_.emplace_back(new Context(d)); // validates every id
    const std::set<int> distinct(devices_.begin(), devices_.end());
    const char *force = std::getenv("PSOLVE_HIP_FORCE_LOOPBACK");
    if ((int)distinct.size() < n_devices || (force && *force == '1')) {
        // repeated ids: several shards share a GPU; RCCL refuses that, the host-synchronised loopback does not
        group_ = local_group_create(n_devices);
        for (int r = 0; r < n_devices; ++r) shards_[(size_t)r]->comm_init_local(group_, r);
    } else {
        std::vector<Comm *> comms;
        for (int r = 0; r < n_devices; ++r) comms.push_back(&shards_[(size_t)r]->comm());
        Comm::init_all(comms, devices_, nullptr); // ncclCommInitAll: one clique inside this process
    }
    // the peer-mapped collectives ride on top of either transport (dist_peer.hip); a node whose devices cannot map each
    // other's memory simply has none ("dist.peer_available" 0: "dist_collectives" 1 then keeps RCCL)
    try {
        peer_ = peer_group_create(devices_);
    } catch (const Error &) {
        peer_ = nullptr;
        (void)hipGetLastError();
    }
    if (peer_)
        for (int r = 0; r < n_devices; ++r) shards_[(size_t)r]->comm().attach_peer(peer_, r);
    std::memset(&info, 0, sizeof(info));
    info.true_residual = -1.0;
}

MultiContext::~MultiContext()
{
    shards_.clear(); // communicators die with their contexts, before the groups
    if (group_) local_group_destroy(group_);
    if (peer_) peer_group_destroy(peer_);
}

void MultiContext::run_all(const std::function<void(int, Context &)> &f)
{
    const int W = world();
    std::vector<std::exception_ptr> err((size_t)W);
    if (peer_) peer_group_reset(peer_);
    if (group_) local_group_reset(group_);
    else {
        // an earlier call ended with a shard failing outside a collective: the clique was aborted to free the others and
        // is made again here.  What the shards hold -- matrices, halo plans, hierarchies -- does not live in the
        // communicators: a factorization that was complete stays valid (like on the loopback group), one that was
        // interrupted was never marked complete (factorize_host clears factorized_ first).
        bool any_dead = false;
        for (auto &s : shards_) any_dead = any_dead || s->comm().aborted();
        if (any_dead) {
            std::vector<Comm *> comms;
            for (auto &s : shards_) comms.push_back(&s->comm());
            Comm::init_all(comms, devices_, nullptr);
        }
    }
    std::vector<std::thread> th;
    th.reserve((size_t)W);
    for (int r = 0; r < W; ++r)
        th.emplace_back([&, r] {
            try {
                f(r, *shards_[(size_t)r]);
            } catch (const Error &e) {
                err[(size_t)r] = std::current_exception();
                // a rank that leaves a collective sequence early would block the others for ever: wake them (loopback)
                // or make RCCL give up the operations they are blocked in (in-process clique).  Not for the failures all
                // ranks agreed on (a non-finite diagonal, a numeric failure of the preconditioner setup: Newton catches
                // those and goes on, Newton.cpp:191-202) -- there every rank has left the sequence at the same point,
                // and aborting would cost two ncclCommInitAll at the next call for nothing.
                if (!e.agreed) abort_all();
            } catch (...) {
                err[(size_t)r] = std::current_exception();
                abort_all();
            }
        });
    for (auto &t : th) t.join();
    // report the root cause: an ECOMM "aborted" on rank q is only the echo of another rank's failure
    std::exception_ptr first = nullptr, first_real = nullptr;
    int rank_real = -1;
    for (int r = 0; r < W; ++r) {
        if (!err[(size_t)r]) continue;
        if (!first) first = err[(size_t)r];
        try {
            std::rethrow_exception(err[(size_t)r]);
        } catch (const Error &e) {
            if (e.code != PSOLVE_HIP_ECOMM && !first_real) {
                first_real = err[(size_t)r];
                rank_real = r;
            }
        } catch (...) {
            if (!first_real) {
                first_real = err[(size_t)r];
                rank_real = r;
            }
        }
    }
    if (first_real) {
        try {
            std::rethrow_exception(first_real);
        } catch (const Error &e) {
            throw Error(e.code, "shard " + std::to_string(rank_real) + " (device " +
                                    std::to_string(devices_[(size_t)rank_real]) + "): " + e.what());
        }
    }
    if (first) std::rethrow_exception(first);
}

void MultiContext::abort_all()
{
    if (peer_) peer_group_abort(peer_);
    if (group_) local_group_abort(group_);
    else
        for (auto &s : shards_) s->comm().abort();
}

void MultiContext::set_param(const std::string &key, double v)
{
    for (auto &s : shards_) s->set_param(key, v); // host-only state: no thread needed
}

double MultiContext::get_param(const std::string &key) const
{
    if (key == "devices") return (double)shards_.size();
    if (key == "reorder.active") return reordered_ ? 1 : 0;
    if (key == "reorder.levels") return ro_info_.levels;
    if (key == "reorder.components") return ro_info_.components;
    if (key == "reorder.isolated") return ro_info_.isolated;
    if (key == "reorder.leftover") return ro_info_.leftover;
    if (key == "reorder.spread_before") return ro_spread_before_;
    if (key == "reorder.spread_after") return ro_spread_after_;
    if (key == "reorder.seconds") return ro_seconds_;
    if (key == "stats.reorder_searches") return (double)ro_searches_;
    if (key == "dist.comm_aborted") { // a shard's communicator was aborted (and not yet made again)
        for (auto &s : shards_)
            if (s->comm().aborted()) return 1.0;
        return 0.0;
    }
    if (key == "dist.n_halo") { // the largest halo of a shard
        double v = 0.0;
        for (auto &s : shards_) v = std::max(v, s->get_param(key));
        return v;
    }
    if (key.rfind("stats.", 0) == 0) { // bytes and uploads add up over the shards; counts of calls do not
        const bool additive = key == "stats.h2d_bytes" || key == "stats.d2h_bytes";
        double v = 0.0;
        for (auto &s : shards_) v = additive ? v + s->get_param(key) : std::max(v, s->get_param(key));
        return v;
    }
    return shards_[0]->get_param(key);
}

void MultiContext::synchronize()
{
    for (auto &s : shards_) s->synchronize();
}

void MultiContext::trim()
{
    for (auto &s : shards_) {
        s->synchronize();
        s->meter.trim();
    }
}

void partition_rows_by_nnz(int64_t n, const int32_t *outer, int world, int64_t align, std::vector<int64_t> &offsets)
{
    const int W = world;
    align = std::max<int64_t>(1, align);
    PS_REQUIRE(W >= 1 && n >= (int64_t)W * 16 * align, PSOLVE_HIP_EINVAL,
               "matrix of " + std::to_string(n) + " rows is too small to partition over " + std::to_string(W) +
                   " devices (use a single-device handle)");
    const int64_t nnz = outer[n];
    offsets.assign((size_t)W + 1, 0);
    offsets[(size_t)W] = n;
    for (int r = 1; r < W; ++r) {
        const int64_t target = nnz * r / W;
        int64_t row = std::lower_bound(outer, outer + n + 1, (int32_t)std::min<int64_t>(target, INT32_MAX)) - outer;
        row = (row / align) * align;
        const int64_t lo = offsets[(size_t)r - 1] + align, hi = n - (int64_t)(W - r) * align;
        offsets[(size_t)r] = std::min(std::max(row, lo), hi);
    }
}

void MultiContext::partition_rows(int64_t n, const int32_t *outer)
{
    partition_rows_by_nnz(n, outer, world(), shards_[0]->prm.block_size, row_offsets_);
}

void MultiContext::analyze_pattern(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, int precond_num)
{
    const double t0 = wall_seconds();
    PS_REQUIRE(n > 0 && nnz >= 0 && outer && (inner || nnz == 0), PSOLVE_HIP_EINVAL, "analyze_pattern: null / empty pattern");
    PS_REQUIRE(outer[0] == 0 && outer[n] == nnz, PSOLVE_HIP_EINVAL,
               "analyze_pattern: outer[0] != 0 or outer[n] != nnz (matrix must be compressed)");
    PS_REQUIRE(n < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "global size exceeds int32 column ids");
    (void)precond_num;
    info.time_analyze = wall_seconds() - t0;
}

void MultiContext::factorize_host(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner, const double *values)
{
    const double t0 = wall_seconds();
    factorized_ = false;
    PS_REQUIRE(n > 0 && nnz >= 0 && outer && inner && values, PSOLVE_HIP_EINVAL, "factorize: null / empty matrix arrays");
    PS_REQUIRE(outer[0] == 0 && outer[n] == nnz, PSOLVE_HIP_EINVAL,
               "factorize: outer[0] != 0 or outer[n] != nnz (matrix must be compressed)");
    PS_REQUIRE(n < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "global size exceeds int32 column ids");
    const int bs = shards_[0]->prm.block_size;
    PS_REQUIRE(bs == 1 || n % bs == 0, PSOLVE_HIP_EINVAL, "block_size does not divide the matrix size");
    n_ = n;
    const Params &P = shards_[0]->prm;
    reordered_ = false;
    if (P.reorder == 1 || (P.reorder == 2 && P.precond <= 2 && n >= P.reorder_min_rows)) {
        const double t1 = wall_seconds();
        reordered_ = decide_order(n, nnz, outer, inner);
        ro_seconds_ = wall_seconds() - t1;
    }
    if (!reordered_) {
        partition_rows(n, outer);
        run_all([&](int r, Context &c) {
            c.factorize_host_rows(n, row_offsets_[(size_t)r], row_offsets_[(size_t)r + 1], outer, inner, values);
        });
    } else {
        // row pointers of the renumbered matrix, the partition by ITS stored entries, and every shard packs its rows in
        // the new order (the caller's column ids; the device renames and sorts them)
        std::vector<int32_t> pptr((size_t)n + 1);
        pptr[0] = 0;
        for (int64_t k = 0; k < n; ++k) {
            const int32_t old = order_[(size_t)k];
            pptr[(size_t)k + 1] = pptr[(size_t)k] + (outer[old + 1] - outer[old]);
        }
        partition_rows_by_nnz(n, pptr.data(), world(), bs, row_offsets_);
        run_all([&](int r, Context &c) {
            const int64_t lo = row_offsets_[(size_t)r], hi = row_offsets_[(size_t)r + 1];
            const int64_t k0 = pptr[(size_t)lo], cnt = (int64_t)pptr[(size_t)hi] - k0;
            std::vector<int32_t> ptr((size_t)(hi - lo) + 1), col((size_t)cnt + 1);
            std::vector<double> val((size_t)cnt + 1);
            for (int64_t k = lo; k < hi; ++k) {
                const int32_t old = order_[(size_t)k];
                const int64_t src = outer[old], len = (int64_t)outer[old + 1] - src, dst = (int64_t)pptr[(size_t)k] - k0;
                ptr[(size_t)(k - lo)] = (int32_t)dst;
                std::memcpy(col.data() + dst, inner + src, (size_t)len * sizeof(int32_t));
                std::memcpy(val.data() + dst, values + src, (size_t)len * sizeof(double));
            }
            ptr[(size_t)(hi - lo)] = (int32_t)cnt;
            c.factorize_host_rows_packed(n, lo, hi, ptr.data(), col.data(), val.data(), new_of_old_.data(), order_version_);
        });
    }
    factorized_ = true;
    info.amg_levels = shards_[0]->info.amg_levels;
    info.time_factorize = wall_seconds() - t0;
}

// the order is kept while the pattern stays the same (hash of the caller's arrays, summed over index ranges by the shards'
// host threads), like the single-device handle keeps it (Context::reorder_matrix)
bool MultiContext::decide_order(int64_t n, int64_t nnz, const int32_t *outer, const int32_t *inner)
{
    const Params &P = shards_[0]->prm;
    const int W = world();
    const int b = (P.block_size > 1 && n % P.block_size == 0) ? P.block_size : 1;
    const HostPatternHash hp = hash_host_pattern(n, nnz, outer, inner, std::max(W, 4));
    const uint64_t h = hp.outer * 0x9E3779B97F4A7C15ull + hp.inner;
    const bool same = ro_n_ == n && ro_nnz_ == nnz && ro_block_ == b && ro_hash_ == h && ro_mode_ == P.reorder &&
                      ro_min_spread_ == P.reorder_min_spread && ro_reverse_ == P.reorder_reverse;
    if (same) return ro_decision_;
    ro_n_ = -1;
    ++ro_searches_;
    std::vector<int32_t> order, new_of_old;
    bool take = false;
    std::exception_ptr err;
    std::thread t([&] { // (a thread of its own: the device of shard 0 is this thread's current device only)
        try {
            take = shards_[0]->order_host_pattern(n, nnz, outer, inner, order, new_of_old, ro_info_, ro_spread_before_, ro_spread_after_);
        } catch (...) {
            err = std::current_exception();
        }
    });
    t.join();
    if (err) {
        bool fatal = true;
        try {
            std::rethrow_exception(err);
        } catch (const Error &e) {
            fatal = !(P.reorder == 2 && e.code == PSOLVE_HIP_EDEVICE); // auto: no room for the pattern on one device
        } catch (...) {
        }
        if (fatal) std::rethrow_exception(err);
        take = false;
    }
    if (take) {
        order_.swap(order);
        new_of_old_.swap(new_of_old);
        ++order_version_;
    } else {
        order_.clear();
        new_of_old_.clear();
    }
    ro_decision_ = take;
    ro_n_ = n;
    ro_nnz_ = nnz;
    ro_block_ = b;
    ro_hash_ = h;
    ro_mode_ = P.reorder;
    ro_min_spread_ = P.reorder_min_spread;
    ro_reverse_ = P.reorder_reverse;
    return take;
}

void MultiContext::solve_host(const double *b, double *x)
{
    const double t0 = wall_seconds();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "[HIP] solve before factorize (size mismatch?)");
    PS_REQUIRE(b && x, PSOLVE_HIP_EINVAL, "solve: null vector");
    run_all([&](int r, Context &c) {
        const int64_t r0 = row_offsets_[(size_t)r], r1 = row_offsets_[(size_t)r + 1];
        if (!reordered_) {
            c.solve_host(b + r0, x + r0);
            return;
        }
        // b and the initial guess into the new numbering, x back: every shard moves its own rows
        std::vector<double> bn((size_t)(r1 - r0)), xn((size_t)(r1 - r0));
        for (int64_t k = r0; k < r1; ++k) {
            bn[(size_t)(k - r0)] = b[order_[(size_t)k]];
            xn[(size_t)(k - r0)] = x[order_[(size_t)k]];
        }
        c.solve_host(bn.data(), xn.data());
        for (int64_t k = r0; k < r1; ++k) x[order_[(size_t)k]] = xn[(size_t)(k - r0)];
    });
    const double t_an = info.time_analyze, t_fa = info.time_factorize;
    info = shards_[0]->info; // every rank took the same decisions from the same all-reduced scalars
    info.time_analyze = t_an;
    info.time_factorize = t_fa;
    info.time_solve_device = 0.0;
    for (auto &s : shards_) info.time_solve_device = std::max(info.time_solve_device, s->info.time_solve_device);
    info.time_solve = wall_seconds() - t0;
}

} // namespace psolve
