// multi.cpp -- the in-process multi-device handle (see multi.hpp).
#include "multi.hpp"

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
    std::memset(&info, 0, sizeof(info));
    info.true_residual = -1.0;
}

MultiContext::~MultiContext()
{
    shards_.clear(); // communicators die with their contexts, before the loopback group
    if (group_) local_group_destroy(group_);
}

void MultiContext::run_all(const std::function<void(int, Context &)> &f)
{
    const int W = world();
    std::vector<std::exception_ptr> err((size_t)W);
    if (group_) local_group_reset(group_);
    else if (shards_[0]->comm().aborted()) {
        // an earlier call ended with a shard failing outside a collective: the clique was aborted to free the others
        // and has to be made again; whatever was factorized on it is gone
        std::vector<Comm *> comms;
        for (auto &s : shards_) comms.push_back(&s->comm());
        Comm::init_all(comms, devices_, nullptr);
        factorized_ = false;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)W);
    for (int r = 0; r < W; ++r)
        th.emplace_back([&, r] {
            try {
                f(r, *shards_[(size_t)r]);
            } catch (...) {
                err[(size_t)r] = std::current_exception();
                // a rank that leaves a collective sequence early would block the others for ever: wake them (loopback)
                // or make RCCL give up the operations they are blocked in (in-process clique)
                if (group_) local_group_abort(group_);
                else
                    for (auto &s : shards_) s->comm().abort();
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

void MultiContext::set_param(const std::string &key, double v)
{
    for (auto &s : shards_) s->set_param(key, v); // host-only state: no thread needed
}

double MultiContext::get_param(const std::string &key) const
{
    if (key == "devices") return (double)shards_.size();
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
    partition_rows(n, outer);
    n_ = n;
    run_all([&](int r, Context &c) {
        c.factorize_host_rows(n, row_offsets_[(size_t)r], row_offsets_[(size_t)r + 1], outer, inner, values);
    });
    factorized_ = true;
    info.amg_levels = shards_[0]->info.amg_levels;
    info.time_factorize = wall_seconds() - t0;
}

void MultiContext::solve_host(const double *b, double *x)
{
    const double t0 = wall_seconds();
    PS_REQUIRE(factorized_, PSOLVE_HIP_EINVAL, "[HIP] solve before factorize (size mismatch?)");
    PS_REQUIRE(b && x, PSOLVE_HIP_EINVAL, "solve: null vector");
    run_all([&](int r, Context &c) {
        const int64_t r0 = row_offsets_[(size_t)r];
        c.solve_host(b + r0, x + r0);
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
