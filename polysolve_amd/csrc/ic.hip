// ic.hip -- device side of precond = "ic": upload of the factor, level layout, the two waiting triangular solves.
#include "ic.hpp"

#include <algorithm>
#include <deque>
#include <memory>
#include <mutex>
#include <numeric>

#include "solver.hpp"

namespace psolve {

namespace {

// One thread per row, rows laid out by dependency level.  Row i: acc = rhs_i - sum_k val_k out[col_k], then
// out[i] = acc * dinv[i].  An entry is consumed as soon as flag[col_k] == epoch; the row publishes its own value and
// flag INSIDE the loop (lanes of one wave may depend on each other: a lane that left the loop could not publish
// before its wave-mates leave it too).
//   PRE : rhs_i = scale[i] * r[i]        (forward solve, L y = S r)
//   POST: z[i] = scale[i] * out[i]       (backward solve, L^T w = y; z = S w)
template <bool PRE>
__global__ __launch_bounds__(kBlock) void ic_trisolve_kernel(int n, const int *__restrict__ order,
                                                              const int *__restrict__ ptr, const int *__restrict__ col,
                                                              const double *__restrict__ val,
                                                              const double *__restrict__ dinv,
                                                              const double *__restrict__ scale,
                                                              const double *__restrict__ rhs, double *out, double *z,
                                                              int *flag, int epoch, const int *__restrict__ done_flag,
                                                              int *ticket)
{
    if (done_flag && *done_flag) return;
    // A persistent grid sized to a few levels' worth of rows (IcPrecond::apply): chunks of 256 positions are handed out
    // by a ticket in the order the workgroups ask for them, so the rows a chunk may wait for belong to chunks that are
    // being worked on or done -- and only a few thousand lanes poll at any time (one workgroup per 256 rows of the
    // whole system, all resident and all polling, cost 110 us per level on a 64^3 grid).
    __shared__ int base_sh;
    for (;;) {
        __syncthreads(); // everybody is done with the previous chunk (and with base_sh)
        if (threadIdx.x == 0) base_sh = atomicAdd(ticket, 1) * kBlock;
        __syncthreads();
        const int base = base_sh;
        if (base >= n) break;
        const int t = base + threadIdx.x;
        const bool live = t < n;
        const int i = live ? order[t] : 0;
        double acc = 0.0;
        int k = 0, e = 0;
        if (live) {
            acc = PRE ? scale[i] * rhs[i] : rhs[i];
            k = ptr[i];
            e = ptr[i + 1];
        }
        // The loop condition is WAVE-uniform (as in amg_aggregate.hip's waiting kernels): no lane leaves before all of
        // its wave are done, so the publication below stays inside the loop -- with a per-lane exit the compiler may
        // sink it behind the loop, where a lane waits for its wave-mates, one of which may be waiting for that value.
        bool pending = live;
        while (__any(pending)) {
            bool moved = false;
            if (pending) {
                // the entries whose columns are final, in order (most rows find all of theirs ready at the first look)
                while (k < e) {
                    const int c = col[k];
                    if (__hip_atomic_load(&flag[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) break;
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    acc -= val[k] * __hip_atomic_load(&out[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++k;
                    moved = true;
                }
                if (k >= e) {
                    const double v = acc * dinv[i];
                    __hip_atomic_store(&out[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!PRE) z[i] = scale[i] * v;
                    __hip_atomic_store(&flag[i], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    pending = false;
                    moved = true;
                }
            }
            if (!__any(moved)) __builtin_amdgcn_s_sleep(4); // nobody moved: leave the memory pipe to the frontier
        }
    }
}

// A handful of dependency levels (the AMD ordering: 9 on an N^3 grid where the natural one has 3 N): no waiting at all --
// one launch per level over its rows, the same additions in the same order as the waiting kernel (round 4).
template <bool PRE>
__global__ __launch_bounds__(kBlock) void ic_level_kernel(int t0, int t1, const int *__restrict__ order,
                                                           const int *__restrict__ ptr, const int *__restrict__ col,
                                                           const double *__restrict__ val, const double *__restrict__ dinv,
                                                           const double *__restrict__ scale, const double *__restrict__ rhs,
                                                           double *out, double *__restrict__ z,
                                                           const int *__restrict__ done_flag)
{
    if (done_flag && *done_flag) return;
    for (int t = t0 + blockIdx.x * kBlock + threadIdx.x; t < t1; t += gridDim.x * kBlock) {
        const int i = order[t];
        double acc = PRE ? scale[i] * rhs[i] : rhs[i];
        for (int k = ptr[i]; k < ptr[i + 1]; ++k) acc -= val[k] * out[col[k]];
        const double v = acc * dinv[i];
        out[i] = v;
        if (!PRE) z[i] = scale[i] * v;
    }
}

// order = the rows sorted by level (stable), levels = 1 + the deepest dependency; starts[l] = first position of level l
int level_layout(int n, const std::vector<int> &level, std::vector<int> &order, std::vector<int> *starts = nullptr)
{
    int nl = 0;
    for (int i = 0; i < n; ++i) nl = std::max(nl, level[(size_t)i] + 1);
    std::vector<int> start((size_t)nl + 1, 0);
    for (int i = 0; i < n; ++i) ++start[(size_t)level[(size_t)i] + 1];
    for (int l = 0; l < nl; ++l) start[(size_t)l + 1] += start[(size_t)l];
    if (starts) *starts = start;
    order.resize((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)start[(size_t)level[(size_t)i]]++] = i;
    return nl;
}

template <typename T>
void upload(DeviceBuffer<T> &d, const std::vector<T> &h, hipStream_t s)
{
    d.ensure(h.size() + 4);
    if (!h.empty()) PS_HIP_CHECK(hipMemcpyAsync(d.ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

// The approximate-minimum-degree order is a function of the pattern and costs seconds on the host (128^3: 1.5 s of the 1.9 s
// factorize): a small process-wide cache keyed by the pattern id lets a second handle on the same mesh -- a Newton solver
// per time step, a solver per load case -- skip it.  Four orders at most, oldest dropped.
struct AmdCacheEntry {
    unsigned long long id;
    int64_t n, nnz;
    std::shared_ptr<const std::vector<int32_t>> order;
};
std::mutex g_amd_cache_mutex;
std::deque<AmdCacheEntry> g_amd_cache;

bool amd_cache_lookup(unsigned long long id, int64_t n, int64_t nnz, std::vector<int32_t> &order)
{
    if (id == 0) return false;
    std::shared_ptr<const std::vector<int32_t>> hit;
    {
        std::lock_guard<std::mutex> g(g_amd_cache_mutex);
        for (const AmdCacheEntry &e : g_amd_cache)
            if (e.id == id && e.n == n && e.nnz == nnz) hit = e.order;
    }
    if (!hit) return false;
    order = *hit;
    return true;
}

void amd_cache_store(unsigned long long id, int64_t n, int64_t nnz, const std::vector<int32_t> &order)
{
    if (id == 0) return;
    auto copy = std::make_shared<const std::vector<int32_t>>(order);
    std::lock_guard<std::mutex> g(g_amd_cache_mutex);
    g_amd_cache.push_back(AmdCacheEntry{id, n, nnz, copy});
    while (g_amd_cache.size() > 4) g_amd_cache.pop_front();
}

} // namespace

void IcPrecond::setup(Context &ctx, const CsrDev &A, double initial_shift, int ordering)
{
    hipStream_t s = ctx.stream;
    const int n = A.n;
    PS_REQUIRE(A.n_ext == A.n, PSOLVE_HIP_EINVAL, "incomplete Cholesky: a square (halo-free) operator expected");
    std::vector<int32_t> hp((size_t)n + 1), hc((size_t)A.nnz);
    std::vector<double> hv((size_t)A.nnz);
    PS_HIP_CHECK(hipMemcpyAsync(hp.data(), A.rowptr, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (A.nnz) {
        PS_HIP_CHECK(hipMemcpyAsync(hc.data(), A.col, (size_t)A.nnz * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipMemcpyAsync(hv.data(), A.val, (size_t)A.nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    PS_HIP_CHECK(hipStreamSynchronize(s));
    // Eigen reads the lower triangle of sorted columns: sort the rows of a caller's unsorted matrix first
    for (int i = 0; i < n; ++i) {
        const int b = hp[(size_t)i], e = hp[(size_t)i + 1];
        if (!std::is_sorted(hc.begin() + b, hc.begin() + e)) {
            std::vector<std::pair<int32_t, double>> row;
            for (int k = b; k < e; ++k) row.emplace_back(hc[(size_t)k], hv[(size_t)k]);
            std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for (int k = b; k < e; ++k) {
                hc[(size_t)k] = row[(size_t)(k - b)].first;
                hv[(size_t)k] = row[(size_t)(k - b)].second;
            }
        }
    }
    // Eigen::IncompleteCholesky<double>'s default ordering: the matrix is factored in the approximate-minimum-degree order
    // (analyzePattern: perm = AMDOrdering(A)^-1; factorize: A.twistedBy(perm)), right-hand sides go in and solutions come
    // out through the permutation (_solve_impl)
    // (the order is a function of the pattern: kept across factorizes of the same one -- Newton.cpp:189-193 -- where the
    // handle knows that, i.e. for the operator it factorized itself, not for a shard's diagonal block)
    // (round-4 advice: the kept order carries the id of the pattern it was computed for; "unchanged since the previous
    // factorize" is not that when a failed call lies between)
    const unsigned long long pid = A.rowptr == ctx.A.rowptr ? ctx.pattern_id_of_A() : 0ull;
    const bool keep_order = ordering == 1 && ordering_ == 1 && (int)order_host_.size() == n && pid != 0 && pid == order_id_ &&
                            A.nnz == order_nnz_;
    ordering_ = ordering;
    if (!keep_order) order_host_.clear();
    order_nnz_ = A.nnz;
    order_id_ = 0;
    if (ordering == 1 && n > 1) {
        if (!keep_order && !amd_cache_lookup(pid, n, A.nnz, order_host_)) {
            amd_order(n, hp.data(), hc.data(), order_host_);
            amd_cache_store(pid, n, A.nnz, order_host_);
        }
        order_id_ = pid;
        std::vector<int32_t> new_of_old((size_t)n);
        for (int k = 0; k < n; ++k) new_of_old[(size_t)order_host_[(size_t)k]] = k;
        std::vector<int32_t> pp((size_t)n + 1, 0), pc((size_t)A.nnz);
        std::vector<double> pv((size_t)A.nnz);
        for (int k = 0; k < n; ++k) pp[(size_t)k + 1] = pp[(size_t)k] + (hp[(size_t)order_host_[(size_t)k] + 1] - hp[(size_t)order_host_[(size_t)k]]);
        std::vector<std::pair<int32_t, double>> row;
        for (int k = 0; k < n; ++k) {
            const int o = order_host_[(size_t)k];
            row.clear();
            for (int j = hp[(size_t)o]; j < hp[(size_t)o + 1]; ++j) row.emplace_back(new_of_old[(size_t)hc[(size_t)j]], hv[(size_t)j]);
            std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for (size_t t = 0; t < row.size(); ++t) {
                pc[(size_t)pp[(size_t)k] + t] = row[t].first;
                pv[(size_t)pp[(size_t)k] + t] = row[t].second;
            }
        }
        hp.swap(pp);
        hc.swap(pc);
        hv.swap(pv);
        upload(perm_, order_host_, s);
        upload(iperm_, new_of_old, s);
        rp_.ensure((size_t)n + 2);
        zp_.ensure((size_t)n + 2);
        PS_HIP_CHECK(hipStreamSynchronize(s));
    }
    IcFactor F;
    ic_factorize(n, hp.data(), hc.data(), hv.data(), initial_shift, F);
    PS_REQUIRE(F.ok, PSOLVE_HIP_ENUMERIC, "incomplete Cholesky: no positive pivots after 10 shifts (matrix not SPD?)");
    n_ = n;
    shift_ = F.shift;
    attempts_ = F.attempts;
    ok_ = true;
    // backward solve (L^T by rows = L by columns): strictly-lower entries of column i, rows > i
    const int64_t nnzL = (int64_t)F.colptr[(size_t)n] - n;
    std::vector<int> bptr((size_t)n + 1, 0), bcol((size_t)nnzL), fptr((size_t)n + 1, 0), fcol((size_t)nnzL);
    std::vector<double> bval((size_t)nnzL), fval((size_t)nnzL), dinv((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int cb = F.colptr[(size_t)i], ce = F.colptr[(size_t)i + 1];
        dinv[(size_t)i] = 1.0 / F.vals[(size_t)cb];
        bptr[(size_t)i + 1] = bptr[(size_t)i] + (ce - cb - 1);
        for (int k = cb + 1; k < ce; ++k) {
            bcol[(size_t)(bptr[(size_t)i] + (k - cb - 1))] = F.rowidx[(size_t)k];
            bval[(size_t)(bptr[(size_t)i] + (k - cb - 1))] = F.vals[(size_t)k];
            ++fptr[(size_t)F.rowidx[(size_t)k] + 1];
        }
    }
    // forward solve: L by rows (the transpose of the above), entries of a row in column order
    for (int i = 0; i < n; ++i) fptr[(size_t)i + 1] += fptr[(size_t)i];
    {
        std::vector<int> cur(fptr.begin(), fptr.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int k = bptr[(size_t)j]; k < bptr[(size_t)j + 1]; ++k) {
                const int r = bcol[(size_t)k], w = cur[(size_t)r]++;
                fcol[(size_t)w] = j;
                fval[(size_t)w] = bval[(size_t)k];
            }
    }
    // levels: forward = depth over the strictly-lower entries of the row, backward = depth over the rows below
    std::vector<int> lf((size_t)n, 0), lb((size_t)n, 0), of, ob;
    for (int i = 0; i < n; ++i)
        for (int k = fptr[(size_t)i]; k < fptr[(size_t)i + 1]; ++k) lf[(size_t)i] = std::max(lf[(size_t)i], lf[(size_t)fcol[(size_t)k]] + 1);
    for (int i = n - 1; i >= 0; --i)
        for (int k = bptr[(size_t)i]; k < bptr[(size_t)i + 1]; ++k) lb[(size_t)i] = std::max(lb[(size_t)i], lb[(size_t)bcol[(size_t)k]] + 1);
    lev_f_ = level_layout(n, lf, of, &start_f_);
    lev_b_ = level_layout(n, lb, ob, &start_b_);
    upload(fptr_, fptr, s);
    upload(fcol_, fcol, s);
    upload(fval_, fval, s);
    upload(bptr_, bptr, s);
    upload(bcol_, bcol, s);
    upload(bval_, bval, s);
    upload(dinv_, dinv, s);
    upload(scale_, F.scale, s);
    upload(order_f_, of, s);
    upload(order_b_, ob, s);
    flag_f_.ensure((size_t)n + 1);
    flag_b_.ensure((size_t)n + 1);
    ticket_.ensure(4);
    y_.ensure((size_t)n + 2);
    w_.ensure((size_t)n + 2);
    PS_HIP_CHECK(hipMemsetAsync(flag_f_.ptr, 0, ((size_t)n + 1) * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(flag_b_.ptr, 0, ((size_t)n + 1) * sizeof(int), s));
    epoch_ = 0;
    PS_HIP_CHECK(hipStreamSynchronize(s)); // the host vectors of this frame were the sources of the copies
}

void IcPrecond::apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag)
{
    PS_REQUIRE(ok_ && n_ > 0, PSOLVE_HIP_EINVAL, "incomplete Cholesky: not factorized");
    hipStream_t s = ctx.stream;
    if (++epoch_ == 0x7fffffff) { // (flags compare equal to the epoch of the running apply)
        PS_HIP_CHECK(hipMemsetAsync(flag_f_.ptr, 0, ((size_t)n_ + 1) * sizeof(int), s));
        PS_HIP_CHECK(hipMemsetAsync(flag_b_.ptr, 0, ((size_t)n_ + 1) * sizeof(int), s));
        epoch_ = 1;
    }
    // about three levels' worth of rows in flight (at least a workgroup per 8 CUs, at most 8 per CU)
    const int chunks = (n_ + kBlock - 1) / kBlock;
    auto grid_for = [&](int levels) {
        const int64_t per_level = ((int64_t)n_ + levels - 1) / std::max(1, levels);
        const int64_t want = (3 * per_level + kBlock - 1) / kBlock;
        return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(chunks, 8 * 256), std::max<int64_t>(want, 32))));
    };
    const dim3 block(kBlock);
    PS_HIP_CHECK(hipMemsetAsync(ticket_.ptr, 0, 2 * sizeof(int), s));
    const bool permuted = !order_host_.empty();
    Launch Lg = ctx.launch_config();
    Lg.stream = s;
    const double *rin = d_r;
    double *zout = d_z;
    if (permuted) { // the factor's numbering: r_p[k] = r[order[k]]
        launch_gather(Lg, n_, perm_.ptr, d_r, rp_.ptr);
        rin = rp_.ptr;
        zout = zp_.ptr;
    }
    auto by_levels = [&](bool pre, const std::vector<int> &start, int levels) {
        for (int l = 0; l < levels; ++l) {
            const int t0 = start[(size_t)l], t1 = start[(size_t)l + 1];
            if (t1 <= t0) continue;
            const dim3 g((unsigned)std::min<int64_t>(8 * 256, ((int64_t)(t1 - t0) + kBlock - 1) / kBlock));
            if (pre)
                hipLaunchKernelGGL(ic_level_kernel<true>, g, block, 0, s, t0, t1, order_f_.ptr, fptr_.ptr, fcol_.ptr, fval_.ptr, dinv_.ptr,
                                   scale_.ptr, rin, y_.ptr, (double *)nullptr, done_flag);
            else
                hipLaunchKernelGGL(ic_level_kernel<false>, g, block, 0, s, t0, t1, order_b_.ptr, bptr_.ptr, bcol_.ptr, bval_.ptr, dinv_.ptr,
                                   scale_.ptr, y_.ptr, w_.ptr, zout, done_flag);
        }
    };
    if (lev_f_ <= kLevelLaunchMax && (int)start_f_.size() == lev_f_ + 1)
        by_levels(true, start_f_, lev_f_);
    else
        hipLaunchKernelGGL(ic_trisolve_kernel<true>, grid_for(lev_f_), block, 0, s, n_, order_f_.ptr, fptr_.ptr, fcol_.ptr, fval_.ptr, dinv_.ptr,
                           scale_.ptr, rin, y_.ptr, (double *)nullptr, flag_f_.ptr, epoch_, done_flag, ticket_.ptr);
    if (lev_b_ <= kLevelLaunchMax && (int)start_b_.size() == lev_b_ + 1)
        by_levels(false, start_b_, lev_b_);
    else
        hipLaunchKernelGGL(ic_trisolve_kernel<false>, grid_for(lev_b_), block, 0, s, n_, order_b_.ptr, bptr_.ptr, bcol_.ptr, bval_.ptr, dinv_.ptr,
                           scale_.ptr, y_.ptr, w_.ptr, zout, flag_b_.ptr, epoch_, done_flag, ticket_.ptr + 1);
    if (permuted) launch_gather(Lg, n_, iperm_.ptr, zp_.ptr, d_z); // z[i] = z_p[new_of_old[i]]
    PS_HIP_CHECK(hipGetLastError());
}

} // namespace psolve
