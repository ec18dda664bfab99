// amg_sweep.hip -- amgcl's ORDERED relaxations on the device (round 6): gauss_seidel and ilu0
// ("/AMGCL/precond/relax/type", /root/reference/linear-solver-spec.json:393-397, passed on by AMGCL.cpp:67-92;
// amgcl/relaxation/gauss_seidel.hpp, ilu0.hpp, detail/ilu_solve.hpp; oracle: amg_oracle.c gs_sweep / ilu0_factor / ilu0_solve).
//
// Both are sweeps in ROW ORDER: row i needs the new values of the rows before it (after it, backwards) that it is coupled
// to.  amgcl's builtin backend runs them serially or by dependency levels -- the same numbers either way.  Here a lane
// (rows of up to a dozen entries: sweep_kernel), 8 or 16 lanes (up to 48: sweep_group_kernel) or a wave (wider rows:
// sweep_wave_kernel) take one (block) row; rows are handed out in sweep order by a ticket counter (a wave never waits for a
// ticket that has not been drawn), and a row that still misses a value polls for it:
//   * no flags and no fences: every output array starts as all-ones bit patterns (a NaN no arithmetic produces), every
//     result is stored once with a device-scope atomic store, and "ready" = "reads as something else" -- each 64-bit value
//     publishes itself;
//   * dependencies INSIDE a wave (a stencil row waits for its left neighbour, the lane before it) are why the wait is a
//     loop over the whole wave in which every lane advances as far as it can: a lane never spins while the lane it waits
//     for is parked at a reconvergence point;
//   * a time limit (20 s of the 100 MHz counter; PSOLVE_SWEEP_LIMIT_MS) ends a sweep that does not make progress by
//     publishing NaN: the solve then reports a non-finite residual instead of hanging the device;
//   * what bounds a sweep is the round trip through the L2 of every dependency between waves (~5-7 us per hop) and, on wide
//     rows, how many rows the device holds at a time -- not the instructions of a turn (profiles/r06_sweeps.md).
// Same operations in the same order as the oracle's serial loops: results are bit-equal.
#include "amg_symbolic.hpp"

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace psolve {

namespace {

constexpr int kSwBlock = 256;
constexpr int kPollWindow = 8; // lanes of a wave (from its first unfinished one) that poll other waves' results
// 20 s of the 100 MHz counter (PSOLVE_SWEEP_LIMIT_MS: another limit, for debugging)
static const long long kSweepLimitTicks = [] {
    const char *e = std::getenv("PSOLVE_SWEEP_LIMIT_MS");
    return e ? std::atoll(e) * 100000ll : 2000000000ll;
}();

__device__ __forceinline__ double ld_live(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool is_unset(double v) { return __double_as_longlong(v) == -1ll; }
// (a result that happens to carry the marker's bits -- only a NaN handed in by the caller can -- goes out as a plain NaN)
__device__ __forceinline__ void st_live(double *p, double v)
{
    if (is_unset(v)) v = __longlong_as_double(0x7ff8000000000000ll);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double quiet_nan() { return __longlong_as_double(0x7ff8000000000000ll); }

// Z = X Y, the oracle's blk_mul
template <int B> __device__ __forceinline__ void mul_bb(const double *X, const double *Y, double *Z)
{
#pragma unroll
    for (int i = 0; i < B; ++i)
#pragma unroll
        for (int j = 0; j < B; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < B; ++k) s += X[i * B + k] * Y[k * B + j];
            Z[i * B + j] = s;
        }
}

// Gauss-Jordan with partial pivoting: the operation order of invert_block_dev (amg_block.hip) / the oracle's blk_inv
template <int B> __device__ __forceinline__ void invert_bb(const double *X, double *Y)
{
    double a[B * B], inv[B * B];
#pragma unroll
    for (int i = 0; i < B * B; ++i) {
        a[i] = X[i];
        inv[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < B; ++i) inv[i * B + i] = 1.0;
#pragma unroll
    for (int c = 0; c < B; ++c) {
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < B; ++r)
            if (fabs(a[r * B + c]) > fabs(a[piv * B + c])) piv = r;
        if (piv != c) {
#pragma unroll
            for (int k = 0; k < B; ++k) {
                double t = a[c * B + k];
                a[c * B + k] = a[piv * B + k];
                a[piv * B + k] = t;
                t = inv[c * B + k];
                inv[c * B + k] = inv[piv * B + k];
                inv[piv * B + k] = t;
            }
        }
        const double d = 1.0 / a[c * B + c];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            a[c * B + k] *= d;
            inv[c * B + k] *= d;
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            if (r == c) continue;
            const double f = a[r * B + c];
            if (f == 0.0) continue;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                a[r * B + k] -= f * a[c * B + k];
                inv[r * B + k] -= f * inv[c * B + k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < B * B; ++i) Y[i] = inv[i];
}

// MODE 0 gauss_seidel forward, 1 backward: X = in_i - sum_{c != i} a_ic x_c (new values on the side already swept, `old` on the
// other), out_i = D_i^-1 X.  MODE 2 forward substitution with the unit lower factor: out_i = in_i - sum_{c < i} l_ic out_c.
// MODE 3 backward substitution: out_i = D_i^-1 (in_i - sum_{c > i} u_ic out_c).  Entries in storage order in every mode.
//
// A turn of the wave = every lane walks up to four entries of its row, one after the other (a rolled loop: ~200 instructions a
// turn; the version that staged eight entries in registers and consumed them through unrolled, predicated code took ~1 450,
// and the turns of all resident waves are bound by instruction issue -- profiles/r06_sweeps.md).  A dependency on a row of the
// SAME ticket (the lane's left neighbour on a stencil: a chain of 64 through every wave) is served from LDS, where a finished
// lane leaves its result; a dependency on another wave's row is polled in the L2, by the first waiting lanes of the wave only
// and not at every turn.
template <int B, int MODE, int U>
__global__ __launch_bounds__(kSwBlock) void sweep_kernel(int nb, const int *__restrict__ ptr, const int *__restrict__ col,
                                                         const double *__restrict__ val, const double *__restrict__ dinv,
                                                         const double *__restrict__ in, const double *__restrict__ old, double *out,
                                                         int *ctrl, long long limit_ticks, const int *__restrict__ done,
                                                         unsigned nap_cap, unsigned period_mask)
{
    if (done && *done) return;
    constexpr bool kBack = (MODE & 1) != 0, kSolve = MODE >= 2, kScale = MODE != 2;
    constexpr int BB = B * B;
    // this wave's results of the current ticket, by lane.  A plain __shared__ array with wavefront-scope fences around the
    // hand-over: declared volatile its accesses were compiled to FLAT loads and stores instead of ds_read / ds_write
    __shared__ double fw_all[kSwBlock * B];
    const int fwb = (threadIdx.x & ~63) * B;
    const int lane = threadIdx.x & 63;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ctrl[0], 64);
        base = __shfl(base, 0);
        if (base >= nb) return;
        const int t = base + lane;
        bool active = t < nb;
        const int i = kBack ? nb - 1 - t : t;
        int j = 0, end = 0;
        double X[B];
#pragma unroll
        for (int r = 0; r < B; ++r) X[r] = 0.0;
        if (active) {
            j = ptr[i];
            end = ptr[i + 1];
#pragma unroll
            for (int r = 0; r < B; ++r) X[r] = in[(size_t)i * B + r];
        }
        unsigned long long published = 0; // lanes of this ticket whose result is in LDS
        unsigned spins = 0, idle = 0, turn = 0;
        bool stalled = true;
        while (__any(active)) {
            bool moved = false, finished = false;
            // other waves' rows: asked for by every waiting lane, every second turn or when the wave stands still
            const bool may_poll = stalled || (turn & period_mask) == 0;
            if (active) {
#pragma unroll 1
                for (int it = 0; it < U && j < end; ++it) {
                    const int c = col[j];
                    const bool fresh = kBack ? c > i : c < i;
                    if (c != i && !(kSolve && !fresh)) {
                        double xc[B];
                        bool got = true;
                        if (fresh) {
                            const int tc = (kBack ? nb - 1 - c : c) - base;
                            if (tc >= 0 && tc < 64) {
                                got = (published >> tc) & 1ull;
                                if (got) {
#pragma unroll
                                    for (int r = 0; r < B; ++r) xc[r] = fw_all[fwb + tc * B + r];
                                }
                            } else if (may_poll) {
#pragma unroll
                                for (int r = 0; r < B; ++r) {
                                    xc[r] = ld_live(out + (size_t)c * B + r);
                                    got = got && !is_unset(xc[r]);
                                }
                            } else {
                                got = false;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < B; ++r) xc[r] = old[(size_t)c * B + r];
                        }
                        if (!got) break;
                        const double *v = val + (size_t)j * BB;
#pragma unroll
                        for (int r = 0; r < B; ++r) {
                            double s = 0.0;
#pragma unroll
                            for (int q = 0; q < B; ++q) s += v[r * B + q] * xc[q];
                            X[r] -= s;
                        }
                    }
                    ++j;
                    moved = true;
                }
                if (j >= end) {
                    double y[B];
                    if (kScale) {
                        const double *d = dinv + (size_t)i * BB;
#pragma unroll
                        for (int r = 0; r < B; ++r) {
                            double s = 0.0;
#pragma unroll
                            for (int q = 0; q < B; ++q) s += d[r * B + q] * X[q];
                            y[r] = s;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < B; ++r) y[r] = X[r];
                    }
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        if (is_unset(y[r])) y[r] = quiet_nan();
                        st_live(out + (size_t)i * B + r, y[r]);
                        fw_all[fwb + lane * B + r] = y[r];
                    }
                    active = false;
                    moved = true;
                    finished = true;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (what finished lanes left in LDS ...)
            published |= __ballot(finished);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (... is what the next turn reads)
            if ((++spins & 255u) == 0 && (long long)wall_clock64() - t0 > limit_ticks) {
                if (active) { // give up: whoever waits for this row goes on with NaN
#pragma unroll
                    for (int r = 0; r < B; ++r) st_live(out + (size_t)i * B + r, quiet_nan());
                    active = false;
                }
                ctrl[1] = 1;
            }
            ++turn;
            stalled = !__any(moved);
            if (!stalled) idle = 0;
            else {
                idle = min(idle + 1, nap_cap);
                for (unsigned z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
}

// ilu0, IKJ order on A's own pattern.  `work`: A's values on entry, the row's running values (its lane only); `lu`: the
// published factors (strictly lower part: the multipliers; upper part and diagonal: the eliminated row), all-ones on entry;
// `dinv`: the inverted pivots, all-ones on entry, stored LAST by a row -- a row that reads D_c as set finds row c's upper part
// stored or on its way.  ctrl[1]: time limit hit; ctrl[2]: a row that is not sorted; ctrl[3]: a row without its diagonal.
template <int B>
__global__ __launch_bounds__(kSwBlock) void ilu0_factor_kernel(int nb, const int *__restrict__ ptr, const int *__restrict__ col,
                                                               double *work, double *lu, double *dinv, int *ctrl,
                                                               long long limit_ticks)
{
    constexpr int BB = B * B;
    const int lane = threadIdx.x & 63;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ctrl[0], 64);
        base = __shfl(base, 0);
        if (base >= nb) return;
        const int i = base + lane;
        bool active = i < nb;
        int j = 0, end = 0, prev = -1;
        if (active) {
            j = ptr[i];
            end = ptr[i + 1];
        }
        unsigned spins = 0;
        while (__any(active)) {
            bool moved = false;
            if (active) {
                for (;;) {
                    int c = j < end ? col[j] : INT_MAX;
                    if (j < end && c <= prev) {
                        ctrl[2] = 1;
                        c = INT_MAX;
                    }
                    if (c >= i) {
                        double D[BB], Dv[BB];
                        if (c == i) {
#pragma unroll
                            for (int e = 0; e < BB; ++e) D[e] = work[(size_t)j * BB + e];
                        } else {
                            ctrl[3] = 1;
#pragma unroll
                            for (int e = 0; e < BB; ++e) D[e] = (e % (B + 1) == 0) ? 1.0 : 0.0;
                        }
                        invert_bb<B>(D, Dv);
                        for (int k = j; k < end; ++k)
#pragma unroll
                            for (int e = 0; e < BB; ++e) st_live(lu + (size_t)k * BB + e, work[(size_t)k * BB + e]);
#pragma unroll
                        for (int e = 0; e < BB; ++e) st_live(dinv + (size_t)i * BB + e, Dv[e]);
                        active = false;
                        moved = true;
                        break;
                    }
                    // a lower entry: row c must be finished
                    double Dc[BB];
                    bool ready = true;
#pragma unroll
                    for (int e = 0; e < BB; ++e) {
                        Dc[e] = ld_live(dinv + (size_t)c * BB + e);
                        ready = ready && !is_unset(Dc[e]);
                    }
                    if (!ready) break;
                    double W[BB], tl[BB];
#pragma unroll
                    for (int e = 0; e < BB; ++e) W[e] = work[(size_t)j * BB + e];
                    mul_bb<B>(W, Dc, tl);
#pragma unroll
                    for (int e = 0; e < BB; ++e) {
                        work[(size_t)j * BB + e] = tl[e];
                        st_live(lu + (size_t)j * BB + e, tl[e]);
                    }
                    int p = j + 1;
                    const int ce = ptr[c + 1];
                    for (int k = ptr[c]; k < ce; ++k) {
                        const int ck = col[k];
                        if (ck <= c) continue;
                        while (p < end && col[p] < ck) ++p;
                        if (p >= end) break;
                        if (col[p] != ck) continue;
                        double U[BB], prod[BB];
#pragma unroll
                        for (int e = 0; e < BB; ++e) {
                            unsigned tries = 0;
                            do {
                                U[e] = ld_live(lu + (size_t)k * BB + e);
                            } while (is_unset(U[e]) && ((++tries & 1023u) || (long long)wall_clock64() - t0 <= limit_ticks)); // (on its way: stored before D_c was)
                        }
                        mul_bb<B>(tl, U, prod);
#pragma unroll
                        for (int e = 0; e < BB; ++e) work[(size_t)p * BB + e] -= prod[e];
                    }
                    prev = c;
                    ++j;
                    moved = true;
                }
            }
            if ((++spins & 255u) == 0 && (long long)wall_clock64() - t0 > limit_ticks) {
                if (active) {
                    for (int k = ptr[i]; k < end; ++k)
#pragma unroll
                        for (int e = 0; e < BB; ++e) st_live(lu + (size_t)k * BB + e, quiet_nan());
#pragma unroll
                    for (int e = 0; e < BB; ++e) st_live(dinv + (size_t)i * BB + e, quiet_nan());
                    active = false;
                }
                ctrl[1] = 1;
            }
            if (!__any(moved)) __builtin_amdgcn_s_sleep(8);
        }
    }
}

// ---- wide rows: one WAVE per (block) row -------------------------------------------------------------------------------
// A coarse level's row has dozens to hundreds of entries and, on the coarsest levels, waits for most of the rows before it: a
// lane per row walks them one memory round trip after the other while its 63 neighbours idle (Poisson 64^3, the 769-row
// coarsest level: ~5 ms per sweep).  Here the 64 lanes load 64 entries and their operands at once; the products are then
// subtracted in storage order by reading the lanes one after the other (a register read each), so the sum is still the
// serial loop's, bit for bit.
template <int B, int MODE>
__global__ __launch_bounds__(kSwBlock) void sweep_wave_kernel(int nb, const int *__restrict__ ptr, const int *__restrict__ col,
                                                              const double *__restrict__ val, const double *__restrict__ dinv,
                                                              const double *__restrict__ in, const double *__restrict__ old,
                                                              double *out, int *ctrl, long long limit_ticks,
                                                              const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr bool kBack = (MODE & 1) != 0, kSolve = MODE >= 2, kScale = MODE != 2;
    constexpr int BB = B * B;
    const int lane = threadIdx.x & 63;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        // (everything that steers the wave is made a scalar: the loops below are uniform branches, not exec-mask loops)
        int t = 0;
        if (lane == 0) t = atomicAdd(&ctrl[0], 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nb) return;
        const int i = kBack ? nb - 1 - t : t;
        int j = __builtin_amdgcn_readfirstlane(ptr[i]);
        const int end = __builtin_amdgcn_readfirstlane(ptr[i + 1]);
        double X[B];
#pragma unroll
        for (int r = 0; r < B; ++r) X[r] = in[(size_t)i * B + r];
        unsigned spins = 0;
        bool gave_up = false;
        while (j < end && !gave_up) {
            // 64 entries at a time; a lane keeps its entry until its operand is there (only the lanes still waiting poll)
            const int jj = j + lane;
            const int count = min(64, end - j);
            const int c = jj < end ? col[jj] : -1;
            const bool fresh = kBack ? c > i : c < i;
            const bool skip = c < 0 || c == i || (kSolve && !fresh);
            bool ok = skip;
            double p[B];
#pragma unroll
            for (int r = 0; r < B; ++r) p[r] = 0.0;
            const unsigned long long skips = __ballot(skip);
            int cons = 0;
            unsigned idle = 0;
            while (cons < count && !gave_up) {
                if (!ok && (!fresh || lane < cons + 2 * kPollWindow)) { // (consumed in order: the lanes far behind `cons` need not ask yet)
                    double xs[B];
                    bool got = true;
                    if (fresh) {
#pragma unroll
                        for (int r = 0; r < B; ++r) {
                            xs[r] = ld_live(out + (size_t)c * B + r);
                            got = got && !is_unset(xs[r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < B; ++r) xs[r] = old[(size_t)c * B + r];
                    }
                    if (got) {
                        const double *v = val + (size_t)jj * BB;
#pragma unroll
                        for (int r = 0; r < B; ++r) {
                            double s0 = 0.0;
#pragma unroll
                            for (int q = 0; q < B; ++q) s0 += v[r * B + q] * xs[q];
                            p[r] = s0;
                        }
                        ok = true;
                    }
                }
                const unsigned long long waiting = ~__ballot(ok) >> cons; // (lanes below cons are done; lanes >= count skip)
                const int lead = __builtin_amdgcn_readfirstlane(min(waiting ? __ffsll((long long)waiting) - 1 : 64, count - cons));
                for (int k = cons; k < cons + lead; ++k) {
                    if ((skips >> k) & 1ull) continue;
#pragma unroll
                    for (int r = 0; r < B; ++r) X[r] -= __shfl(p[r], k);
                }
                cons += lead;
                if (lead == 0) {
                    if ((++spins & 63u) == 0 && (long long)wall_clock64() - t0 > limit_ticks) gave_up = true;
                    idle = min(idle + 1, 16u);
                    for (unsigned z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(8);
                } else idle = 0;
            }
            j += count;
        }
        double y[B];
        if (kScale) {
            const double *d = dinv + (size_t)i * BB;
#pragma unroll
            for (int r = 0; r < B; ++r) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < B; ++q) s += d[r * B + q] * X[q];
                y[r] = s;
            }
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r) y[r] = X[r];
        }
        // EVERY lane stores the (same) result: with the store under `lane == 0` the compiler moves the whole row -- ballots and
        // lane reads included -- under that condition and leaves lanes 1 .. 63 circling on row 0 (seen in the ISA, round 6)
#pragma unroll
        for (int r = 0; r < B; ++r) st_live(out + (size_t)i * B + r, gave_up ? quiet_nan() : y[r]);
        if (gave_up) ctrl[1] = 1;
    }
}

// ---- rows of a dozen to a few dozen entries: G = 16 or 32 lanes per row, 64 / G rows per wave ---------------------------
// With a whole wave per row the device holds 8 192 rows at a time -- less than one plane of a coarse 106^3 grid, whose rows
// each wait for their neighbour in the plane before: the sweep then moves at (rows in flight) / (one hop through the L2),
// 37 ns per row on level 1 of the 216^3 hierarchy.  Smaller groups put two or four times as many rows in flight.  A group
// works like the wave of sweep_wave_kernel on G entries at a time; the groups of a wave run the same loops (a group that is
// done idles), and a group that waits for the row of the group before it in the same wave finds it like any other row's.
template <int B, int MODE, int G>
__global__ __launch_bounds__(kSwBlock) void sweep_group_kernel(int nb, const int *__restrict__ ptr, const int *__restrict__ col,
                                                               const double *__restrict__ val, const double *__restrict__ dinv,
                                                               const double *__restrict__ in, const double *__restrict__ old,
                                                               double *out, int *ctrl, long long limit_ticks,
                                                               const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr bool kBack = (MODE & 1) != 0, kSolve = MODE >= 2, kScale = MODE != 2;
    constexpr int BB = B * B, NG = 64 / G;
    const int lane = threadIdx.x & 63, gi = lane / G, sl = lane % G, g0 = gi * G;
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull));
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ctrl[0], NG);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= nb) return;
        const int t = base + gi;
        bool busy = t < nb; // this group's row is not finished yet
        const int i = kBack ? nb - 1 - min(t, nb - 1) : min(t, nb - 1);
        int j = ptr[i];
        const int end = busy ? ptr[i + 1] : j;
        double X[B];
#pragma unroll
        for (int r = 0; r < B; ++r) X[r] = in[(size_t)i * B + r];
        // the batch in flight: this lane's entry, its product once the operand is there
        int c = -1, cons = 0, count = 0;
        bool skip = true, fresh = false, ok = true;
        double p[B];
#pragma unroll
        for (int r = 0; r < B; ++r) p[r] = 0.0;
        unsigned spins = 0, idle = 0;
        bool gave_up = false;
        while (__any(busy)) {
            if (busy && cons == count) {
                if (j < end) { // next batch of the row
                    count = min(G, end - j);
                    cons = 0;
                    const int jj = j + sl;
                    c = jj < end ? col[jj] : -1;
                    fresh = kBack ? c > i : c < i;
                    skip = c < 0 || c == i || (kSolve && !fresh);
                    ok = skip;
                }
            }
            if (busy && !ok && (!fresh || sl < cons + kPollWindow)) { // (consumed in order: the lanes far behind need not ask yet)
                double xs[B];
                bool got = true;
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        xs[r] = ld_live(out + (size_t)c * B + r);
                        got = got && !is_unset(xs[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < B; ++r) xs[r] = old[(size_t)c * B + r];
                }
                if (got) {
                    const double *v = val + (size_t)(j + sl) * BB;
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        double s0 = 0.0;
#pragma unroll
                        for (int q = 0; q < B; ++q) s0 += v[r * B + q] * xs[q];
                        p[r] = s0;
                    }
                    ok = true;
                }
            }
            // per group: how many entries from `cons` on are ready
            const unsigned long long okb = (__ballot(ok) >> g0) & gmask, skb = (__ballot(skip) >> g0) & gmask;
            const unsigned long long waiting = (~okb & gmask) >> cons;
            int lead = waiting ? __ffsll((long long)waiting) - 1 : G;
            lead = busy ? min(lead, count - cons) : 0;
            int maxlead = lead;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) maxlead = max(maxlead, __shfl_xor(maxlead, d));
            for (int k = 0; k < maxlead; ++k) {
                const bool take = k < lead && !((skb >> (cons + k)) & 1ull);
#pragma unroll
                for (int r = 0; r < B; ++r) {
                    const double pv = __shfl(p[r], g0 + ((cons + k) & (G - 1)));
                    if (take) X[r] -= pv;
                }
            }
            cons += lead;
            bool finished = false;
            if (busy && cons == count) {
                j += count;
                count = cons = 0;
                if (j >= end) finished = true;
            }
            if (finished || (busy && gave_up)) {
                double y[B];
                if (kScale) {
                    const double *d = dinv + (size_t)i * BB;
#pragma unroll
                    for (int r = 0; r < B; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int q = 0; q < B; ++q) s += d[r * B + q] * X[q];
                        y[r] = s;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < B; ++r) y[r] = X[r];
                }
                // (every lane of the group stores the same result: see sweep_wave_kernel)
#pragma unroll
                for (int r = 0; r < B; ++r) st_live(out + (size_t)i * B + r, gave_up ? quiet_nan() : y[r]);
                if (gave_up) ctrl[1] = 1;
                busy = false;
                ok = skip = true;
            }
            if (maxlead == 0 && !__any(finished)) {
                if ((++spins & 63u) == 0 && (long long)wall_clock64() - t0 > limit_ticks) gave_up = true;
                idle = min(idle + 1, 16u);
                for (unsigned z = 0; z < idle; ++z) __builtin_amdgcn_s_sleep(8);
            } else idle = 0;
        }
    }
}

// ilu0 on wide rows, a wave per (block) row: the lower entries in order (a wave-uniform loop), the updates of an entry -- one
// per entry of row c's upper part that row i stores too, found by bisection in row i -- spread over the lanes.
template <int B>
__global__ __launch_bounds__(kSwBlock) void ilu0_factor_wave_kernel(int nb, const int *__restrict__ ptr, const int *__restrict__ col,
                                                                    double *work, double *lu, double *dinv, int *ctrl,
                                                                    long long limit_ticks)
{
    constexpr int BB = B * B;
    const int lane = threadIdx.x & 63;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        int i = 0;
        if (lane == 0) i = atomicAdd(&ctrl[0], 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= nb) return;
        const int beg = __builtin_amdgcn_readfirstlane(ptr[i]), end = __builtin_amdgcn_readfirstlane(ptr[i + 1]);
        // sorted?  where is the diagonal?
        int dpos = end;
        bool unsorted = false;
        for (int jb = beg; jb < end; jb += 64) {
            const int jj = jb + lane;
            const int c = jj < end ? col[jj] : INT_MAX;
            if (jj + 1 < end && col[jj + 1] <= c) unsorted = true;
            const unsigned long long m = __ballot(c >= i);
            if (m && dpos == end) dpos = jb + __ffsll((long long)m) - 1;
        }
        dpos = __builtin_amdgcn_readfirstlane(dpos);
        if (__any(unsorted)) ctrl[2] = 1;
        bool has_diag = dpos < end;
        if (has_diag) has_diag = __builtin_amdgcn_readfirstlane(col[dpos]) == i;
        bool gave_up = false;
        for (int j = beg; j < dpos && !gave_up; ++j) {
            const int c = __builtin_amdgcn_readfirstlane(col[j]);
            double De = 0.0;
            unsigned spins = 0;
            bool waiting = true;
            while (waiting) {
                De = lane < BB ? ld_live(dinv + (size_t)c * BB + lane) : 0.0;
                waiting = __any(lane < BB && is_unset(De)) != 0;
                if (waiting) {
                    if ((++spins & 63u) == 0 && (long long)wall_clock64() - t0 > limit_ticks) {
                        gave_up = true;
                        waiting = false;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (gave_up) continue;
            double Dc[BB], W[BB], tl[BB];
#pragma unroll
            for (int e = 0; e < BB; ++e) {
                Dc[e] = __shfl(De, e);
                W[e] = work[(size_t)j * BB + e];
            }
            mul_bb<B>(W, Dc, tl);
#pragma unroll
            for (int e = 0; e < BB; ++e) { // (every lane, the same values: see sweep_wave_kernel)
                work[(size_t)j * BB + e] = tl[e];
                st_live(lu + (size_t)j * BB + e, tl[e]);
            }
            const int cb = __builtin_amdgcn_readfirstlane(ptr[c]), ce = __builtin_amdgcn_readfirstlane(ptr[c + 1]);
            for (int kb = cb; kb < ce; kb += 64) {
                const int k = kb + lane;
                if (k >= ce) continue;
                const int ck = col[k];
                if (ck <= c) continue;
                int lo = j + 1, hi = end;
                while (lo < hi) {
                    const int mid = lo + ((hi - lo) >> 1);
                    if (col[mid] < ck) lo = mid + 1; else hi = mid;
                }
                if (lo >= end || col[lo] != ck) continue;
                double U[BB], prod[BB];
#pragma unroll
                for (int e = 0; e < BB; ++e) {
                    unsigned tries = 0;
                    do {
                        U[e] = ld_live(lu + (size_t)k * BB + e);
                    } while (is_unset(U[e]) && ((++tries & 1023u) || (long long)wall_clock64() - t0 <= limit_ticks)); // (on its way: stored before D_c was)
                }
                mul_bb<B>(tl, U, prod);
#pragma unroll
                for (int e = 0; e < BB; ++e) work[(size_t)lo * BB + e] -= prod[e];
            }
            __threadfence_block(); // the next entry reads what this one's lanes stored
        }
        double D[BB], Dv[BB];
#pragma unroll
        for (int e = 0; e < BB; ++e) D[e] = has_diag ? work[(size_t)dpos * BB + e] : ((e % (B + 1) == 0) ? 1.0 : 0.0);
        if (!has_diag) ctrl[3] = 1;
        invert_bb<B>(D, Dv);
        for (int k = (gave_up ? beg : dpos) + lane; k < end; k += 64)
#pragma unroll
            for (int e = 0; e < BB; ++e) st_live(lu + (size_t)k * BB + e, gave_up ? quiet_nan() : work[(size_t)k * BB + e]);
#pragma unroll
        for (int e = 0; e < BB; ++e) st_live(dinv + (size_t)i * BB + e, gave_up ? quiet_nan() : Dv[e]);
        if (gave_up) ctrl[1] = 1;
    }
}

// rows of more than a dozen entries (half a dozen blocks) take a wave each
inline bool wide_rows(const SweepView &A)
{
    static const int forced = [] {
        const char *e = std::getenv("PSOLVE_SWEEP_WAVE"); // (debugging: 0 never, 1 always)
        return e ? std::atoi(e) : -1;
    }();
    if (forced >= 0) return forced != 0;
    return (double)A.nnzb > (A.b == 1 ? 12.0 : 6.0) * (double)A.nb;
}
inline int sweep_wave_grid(const Launch &L, int nb) { return std::max(1, std::min((nb + 3) / 4, L.num_cus * 8)); }
inline int sweep_grid(const Launch &L, int nb) { return std::max(1, std::min((nb + kSwBlock - 1) / kSwBlock, L.num_cus * 8)); }

template <int B>
void launch_sweep_b(const Launch &L, const SweepView &A, int mode, const double *dinv, const double *in, const double *old,
                    double *out, int *ctrl, const int *done)
{
    const dim3 blk(kSwBlock);
    if (wide_rows(A)) {
        const dim3 gw(sweep_wave_grid(L, A.nb));
        const double avg = (double)A.nnzb / std::max(1, A.nb);
        if (avg <= 48.0) { // four (G = 16) or eight (G = 8) rows per wave; wider rows keep a wave each (G = 32 on 70-entry rows: 4.9 against 4.3 ms)
#define PS_GROUP_LAUNCH(GG)                                                                                                        \
    switch (mode) {                                                                                                                \
    case 0: hipLaunchKernelGGL((sweep_group_kernel<B, 0, GG>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break; \
    case 1: hipLaunchKernelGGL((sweep_group_kernel<B, 1, GG>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break; \
    case 2: hipLaunchKernelGGL((sweep_group_kernel<B, 2, GG>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break; \
    default: hipLaunchKernelGGL((sweep_group_kernel<B, 3, GG>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break; \
    }
            static const int gforce = [] { const char *e = std::getenv("PSOLVE_SWEEP_G"); return e ? std::atoi(e) : 0; }();
            // rows in flight are what moves a sweep whose rows wait for their neighbours in the plane before: level 1 of the 216^3
            // hierarchy (1.2 M rows of 32 entries) 44.7 | 32.9 | 24.1 | 18.1 ms per sweep with 64 | 32 | 16 | 8 lanes per row, a
            // lane per row 212 ms; the 263 552 rows of the 128^3 hierarchy's level 1: 5.7 | - | 3.3 | 3.9 ms
            const int gsel = gforce ? gforce : (A.nb >= 500000 ? 8 : 16);
            if (gsel == 8) { PS_GROUP_LAUNCH(8) } else if (gsel == 16) { PS_GROUP_LAUNCH(16) } else { PS_GROUP_LAUNCH(32) }
#undef PS_GROUP_LAUNCH
            return;
        }
        switch (mode) {
        case 0: hipLaunchKernelGGL((sweep_wave_kernel<B, 0>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break;
        case 1: hipLaunchKernelGGL((sweep_wave_kernel<B, 1>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break;
        case 2: hipLaunchKernelGGL((sweep_wave_kernel<B, 2>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break;
        default: hipLaunchKernelGGL((sweep_wave_kernel<B, 3>), gw, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done); break;
        }
        return;
    }
    const dim3 g(sweep_grid(L, A.nb));
    constexpr int U = 4; // entries of a row a lane walks per turn
    // naps of a wave that stands still: at most 4 x 0.25 us; other waves' rows asked for at every turn by the first sixteen waiting
    // lanes (nap cap 0 ... 32 x poll period 1 ... 4, 64^3 | 128^3, ms per solve of 66 | 111 iterations: 115-149 | 943-1 451; this
    // pair 118 | 943)
    // naps of a wave that stands still: at most 16 x 0.25 us; other waves' rows are asked for every second turn (or when the wave
    // stands still) by EVERY waiting lane.  A window of polling lanes behind the first unfinished one -- tried to save L2
    // requests -- serialises the sweep whenever a ticket holds rows that do not depend on its earlier rows (a grid whose lines
    // are not a multiple of 64 rows: the start of the next line waits behind the end of this one, which waits for the previous
    // ticket ...): Poisson 100^3 1.9 s per sweep instead of 2.5 ms, 130^3 5.9 s instead of 5.6 ms (scripts/r6/sweep_sizes.py)
    const unsigned nap_cap = 16u, period_mask = 1u;
    switch (mode) {
    case 0: hipLaunchKernelGGL((sweep_kernel<B, 0, U>), g, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done, nap_cap, period_mask); break;
    case 1: hipLaunchKernelGGL((sweep_kernel<B, 1, U>), g, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done, nap_cap, period_mask); break;
    case 2: hipLaunchKernelGGL((sweep_kernel<B, 2, U>), g, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done, nap_cap, period_mask); break;
    default: hipLaunchKernelGGL((sweep_kernel<B, 3, U>), g, blk, 0, L.stream, A.nb, A.ptr, A.col, A.val, dinv, in, old, out, ctrl, kSweepLimitTicks, done, nap_cap, period_mask); break;
    }
}

} // namespace

void launch_sweep(const Launch &L, const SweepView &A, int mode, const double *dinv, const double *in, const double *old, double *out,
                  int *ctrl, const int *done)
{
    PS_REQUIRE(A.b >= 1 && A.b <= 3, PSOLVE_HIP_EINVAL, "amg.relax_type gauss_seidel / ilu0: block_size 1, 2 or 3");
    PS_HIP_CHECK(hipMemsetAsync(ctrl, 0, 8 * sizeof(int), L.stream));
    PS_HIP_CHECK(hipMemsetAsync(out, 0xFF, (size_t)A.nb * A.b * sizeof(double), L.stream));
    if (A.b == 1) launch_sweep_b<1>(L, A, mode, dinv, in, old, out, ctrl, done);
    else if (A.b == 2) launch_sweep_b<2>(L, A, mode, dinv, in, old, out, ctrl, done);
    else launch_sweep_b<3>(L, A, mode, dinv, in, old, out, ctrl, done);
    PS_HIP_CHECK(hipGetLastError());
    if (std::getenv("PSOLVE_SWEEP_DEBUG")) { // a sweep that does not end within 3 s: where is it?
        const double t0 = wall_seconds();
        while (hipStreamQuery(L.stream) == hipErrorNotReady && wall_seconds() - t0 < 3.0) {}
        if (hipStreamQuery(L.stream) == hipErrorNotReady) {
            hipStream_t side;
            (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
            std::vector<int> hc(8), hp((size_t)A.nb + 1);
            std::vector<double> ho((size_t)A.nb * A.b);
            (void)hipMemcpyAsync(hc.data(), ctrl, 8 * sizeof(int), hipMemcpyDeviceToHost, side);
            (void)hipMemcpyAsync(hp.data(), A.ptr, hp.size() * sizeof(int), hipMemcpyDeviceToHost, side);
            (void)hipMemcpyAsync(ho.data(), out, ho.size() * sizeof(double), hipMemcpyDeviceToHost, side);
            (void)hipStreamSynchronize(side);
            int unset = 0, first = -1, last = -1;
            for (int i = 0; i < A.nb; ++i) {
                long long bits;
                std::memcpy(&bits, &ho[(size_t)i * A.b], 8);
                if (bits == -1ll) {
                    ++unset;
                    if (first < 0) first = i;
                    last = i;
                }
            }
            std::fprintf(stderr, "[psolve sweep debug] mode %d b %d nb %d wide %d: tickets %d, gave_up %d, rows unset %d (first %d, last %d), row lengths %d %d\n",
                         mode, A.b, A.nb, (int)wide_rows(A), hc[0], hc[1], unset, first, last, first >= 0 ? hp[first + 1] - hp[first] : -1,
                         last >= 0 ? hp[last + 1] - hp[last] : -1);
            std::abort();
        }
    }
}

void device_ilu0_factor(const Launch &L, const SweepView &A, DeviceBuffer<double> &work, DeviceBuffer<double> &lu, double *dinv,
                        int *ctrl)
{
    PS_REQUIRE(A.b >= 1 && A.b <= 3, PSOLVE_HIP_EINVAL, "amg.relax_type ilu0: block_size 1, 2 or 3");
    const size_t bb = (size_t)A.b * A.b, nv = (size_t)A.nnzb * bb;
    work.ensure(nv + 2);
    lu.ensure(nv + 2);
    PS_HIP_CHECK(hipMemcpyAsync(work.ptr, A.val, nv * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
    PS_HIP_CHECK(hipMemsetAsync(lu.ptr, 0xFF, nv * sizeof(double), L.stream));
    PS_HIP_CHECK(hipMemsetAsync(dinv, 0xFF, (size_t)A.nb * bb * sizeof(double), L.stream));
    PS_HIP_CHECK(hipMemsetAsync(ctrl, 0, 4 * sizeof(int), L.stream));
    const dim3 g(sweep_grid(L, A.nb)), gw(sweep_wave_grid(L, A.nb)), blk(kSwBlock);
    if (wide_rows(A)) {
        if (A.b == 1)
            hipLaunchKernelGGL(ilu0_factor_wave_kernel<1>, gw, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
        else if (A.b == 2)
            hipLaunchKernelGGL(ilu0_factor_wave_kernel<2>, gw, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
        else
            hipLaunchKernelGGL(ilu0_factor_wave_kernel<3>, gw, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
    } else if (A.b == 1)
        hipLaunchKernelGGL(ilu0_factor_kernel<1>, g, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
    else if (A.b == 2)
        hipLaunchKernelGGL(ilu0_factor_kernel<2>, g, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
    else
        hipLaunchKernelGGL(ilu0_factor_kernel<3>, g, blk, 0, L.stream, A.nb, A.ptr, A.col, work.ptr, lu.ptr, dinv, ctrl, kSweepLimitTicks);
    PS_HIP_CHECK(hipGetLastError());
    int hc[4] = {0, 0, 0, 0};
    PS_HIP_CHECK(hipMemcpyAsync(hc, ctrl, sizeof(hc), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    PS_REQUIRE(hc[2] == 0, PSOLVE_HIP_EINVAL, "amg.relax_type ilu0: the rows of the matrix must be sorted by column");
    PS_REQUIRE(hc[3] == 0, PSOLVE_HIP_ENUMERIC, "amg.relax_type ilu0: no diagonal value in system matrix");
    PS_REQUIRE(hc[1] == 0, PSOLVE_HIP_ENUMERIC, "amg.relax_type ilu0: the factorization did not finish within its time limit");
    work.release(); // (the running values are the published ones now)
}

} // namespace psolve
