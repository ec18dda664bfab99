// amg_aggregate.hip -- the greedy aggregation sweep of amgcl/coarsening/plain_aggregates.hpp, computed on
// the device WITHOUT changing its result.
//
// The sweep is sequential by definition: vertex i seeds a new aggregate iff it is still unassigned when the
// loop reaches it, claims its strong neighbours (stealing them from earlier aggregates) and tentatively
// claims their unassigned strong neighbours.  It has a closed form in terms of the PREDECESSORS of a vertex
// (the vertices that reach it in one or two hops: rows of the transposed strength graph, which is the graph
// itself when the strength pattern is symmetric -- the SPD case):
//   * the seeds are the lexicographically first maximal set such that no seed is reached from an earlier
//     seed in one or two hops (i is a seed iff none of its predecessors j < i is a seed);
//   * a vertex claimed directly belongs to the LAST (largest) seed that claims it (claims overwrite);
//     otherwise to the FIRST (smallest) seed that reaches it -- itself if it is a seed -- because tentative
//     claims do not overwrite; aggregates are numbered in seed order, the ones emptied by later claims
//     (possible only on unsymmetric patterns) are dropped.
// The seed set is found by dependency rounds: every undecided vertex scans its earlier two-hop
// neighbours -- COVERED ones are skipped for good (a resume pointer), a SEED covers it, an UNDECIDED one
// blocks it; a blocked vertex parks on its blocker's wait list and is re-examined in the round after the
// blocker was decided.  Each vertex is re-examined a handful of times; the number of rounds is the depth
// of the dependency chains (about a thousand for a 256^3 grid in natural order).  Graphs whose chains
// are too long fall back to the host sweep (amg_setup.cpp).
#include "amg_symbolic.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>

namespace psolve {


namespace {

enum : int { kUndecided = 0, kSeed = 1, kCovered = 2, kGone = 3 };

// violations += rows that are not strictly ascending, entries (i, c) without (c, i)
__global__ __launch_bounds__(kBlock) void graph_check_kernel(int n, const int *__restrict__ sptr,
                                                              const int *__restrict__ scol, int *__restrict__ bad)
{
    int v = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int b = sptr[i], e = sptr[i + 1];
        for (int j = b; j < e; ++j) {
            const int c = scol[j];
            if (j > b && scol[j - 1] >= c) ++v;
            if (c == i) continue;
            int lo = sptr[c], hi = sptr[c + 1];
            const int end = hi;
            while (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1);
                if (scol[mid] < i) lo = mid + 1; else hi = mid;
            }
            if (lo >= end || scol[lo] != i) ++v;
        }
    }
    if (v) atomicAdd(bad, v);
}

// the same for wide rows: G lanes stride a row (level 1 of the 216^3 hierarchy, 31 entries per row: 1.7 ms with one lane)
template <int G>
__global__ __launch_bounds__(kBlock) void graph_check_group_kernel(int n, const int *__restrict__ sptr,
                                                                    const int *__restrict__ scol, int *__restrict__ bad)
{
    const int lane = threadIdx.x % G;
    int v = 0;
    for (int i = (blockIdx.x * kBlock + threadIdx.x) / G; i < n; i += gridDim.x * (kBlock / G)) {
        const int b = sptr[i], e = sptr[i + 1];
        for (int j = b + lane; j < e; j += G) {
            const int c = scol[j];
            if (j > b && scol[j - 1] >= c) ++v;
            if (c == i) continue;
            int lo = sptr[c], hi = sptr[c + 1];
            const int end = hi;
            while (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1);
                if (scol[mid] < i) lo = mid + 1; else hi = mid;
            }
            if (lo >= end || scol[lo] != i) ++v;
        }
    }
    if (v) atomicAdd(bad, v);
}

struct AggState {
    int *state;          // kUndecided / kSeed / kCovered / kGone
    int *pa, *pb;        // resume position of the scan: entry of N(v), entry of N(N(v)[pa]) (-1: the neighbour itself)
    int *whead, *wnext;  // wait lists: vertices parked on an undecided vertex
    int *wl[2];          // work lists (ping-pong)
    int *dl[2];          // vertices decided in a round
    int *counts;         // [0,1] work-list sizes, [2,3] decided-list sizes, [4] decided so far
};

__global__ __launch_bounds__(kBlock) void agg_init_kernel(int n, const int *__restrict__ id0, AggState S)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        const bool gone = id0[v] != -1;
        S.state[v] = gone ? kGone : kUndecided;
        S.pa[v] = 0;
        S.pb[v] = -1;
        S.whead[v] = -1;
        S.wnext[v] = -1;
        if (!gone) S.wl[0][atomicAdd(&S.counts[0], 1)] = v;
    }
}

// one round, part A: examine the work list.  (pptr, pcol) = predecessor lists, sorted rows.  A group of
// GROUP lanes owns one vertex: lane l takes the l-th first-hop predecessor c (and its successors in steps
// of GROUP), looks at c itself and then at c's earlier predecessors; the group stops at the first first-hop
// entry (in list order) behind which something is undecided -- everything before it is covered for good
// and never looked at again -- or as soon as any lane meets a seed.
template <int GROUP>
__global__ __launch_bounds__(kBlock) void agg_scan_kernel(int round, const int *__restrict__ pptr,
                                                           const int *__restrict__ pcol, AggState S)
{
    const int cur = round & 1;
    const int cnt = S.counts[cur];
    int *dl = S.dl[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) { // book-keeping of the previous round: nobody reads these here
        S.counts[4] += S.counts[2 + (cur ^ 1)];
        S.counts[2 + (cur ^ 1)] = 0;
    }
    const int lane = threadIdx.x % GROUP;
    const int gbase = (threadIdx.x & 63) / GROUP * GROUP;
    const unsigned long long gmask = GROUP == 64 ? ~0ull : (((1ull << GROUP) - 1ull) << gbase);
    const int ngroups = gridDim.x * kBlock / GROUP;
    for (int idx = (blockIdx.x * kBlock + threadIdx.x) / GROUP; idx < cnt; idx += ngroups) {
        const int v = S.wl[cur][idx];
        if (__hip_atomic_load(&S.state[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kUndecided) continue;
        const int vb = pptr[v], deg = pptr[v + 1] - vb;
        const int a0 = S.pa[v], b0 = S.pb[v];
        bool covered = false;
        int stop_a = -1, stop_b = -1, blocker = -1;
        for (int base = a0; base < deg; base += GROUP) {
            const int a = base + lane;
            int kind = 0, myb = -1, blk = -1; // 0 passed, 1 blocked, 2 met a seed
            if (a < deg) {
                const int c = pcol[vb + a];
                if (c != v) {
                    int bb = (a == a0) ? b0 : -1;
                    if (bb < 0) {
                        if (c < v) {
                            const int st = __hip_atomic_load(&S.state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (st == kSeed) kind = 2;
                            else if (st == kUndecided) {
                                kind = 1;
                                blk = c;
                            }
                        }
                        bb = 0;
                    }
                    if (kind == 0) {
                        // four entries per step: their loads (ids, then states) are in flight together -- the chain
                        // of dependent loads is what a round costs -- and they are looked at in list order
                        const int cb = pptr[c], clen = pptr[c + 1] - cb;
                        bool end = false;
                        for (; bb < clen && !end && kind == 0; bb += 4) {
                            int js[4], st[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) js[q] = bb + q < clen ? pcol[cb + bb + q] : INT_MAX;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                st[q] = (js[q] < v && js[q] != c)
                                            ? __hip_atomic_load(&S.state[js[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : kCovered;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (end || kind) continue;
                                if (js[q] >= v) end = true; // sorted rows: the earlier vertices are a prefix
                                else if (st[q] == kSeed) kind = 2;
                                else if (st[q] == kUndecided) {
                                    kind = 1;
                                    blk = js[q];
                                    myb = bb + q;
                                }
                            }
                        }
                    }
                }
            }
            const unsigned long long seeds = __ballot(kind == 2) & gmask;
            const unsigned long long stops = __ballot(kind == 1) & gmask;
            if (seeds) {
                covered = true;
                break;
            }
            if (stops) {
                const int first = __ffsll((long long)stops) - 1; // lane of the wave, lowest = earliest entry
                stop_a = base + (first - gbase);
                stop_b = __shfl(myb, first);
                blocker = __shfl(blk, first);
                break;
            }
        }
        if (lane != 0) continue;
        if (covered || blocker < 0) {
            // (a vertex covered by a seed of this very round may also be claimed by that seed's push in part B:
            // the compare-and-swap there fails on a decided vertex, so it is listed once)
            __hip_atomic_store(&S.state[v], covered ? kCovered : kSeed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dl[atomicAdd(&S.counts[2 + cur], 1)] = v;
        } else {
            S.pa[v] = stop_a; // resume AT the blocker
            S.pb[v] = stop_b;
            int old = __hip_atomic_load(&S.whead[blocker], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                S.wnext[v] = old;
                const int seen = atomicCAS(&S.whead[blocker], old, v);
                if (seen == old) break;
                old = seen;
            }
        }
    }
}

// one round, part B: wake the vertices parked on what was decided, and let every new seed cover the later
// vertices it reaches in one or two hops (fptr, fcol = the strength graph itself, successors); GROUP lanes
// per decided vertex, one first-hop successor per lane
template <int GROUP>
__global__ __launch_bounds__(kBlock) void agg_wake_kernel(int round, const int *__restrict__ fptr,
                                                           const int *__restrict__ fcol, AggState S)
{
    const int cur = round & 1, nxt = cur ^ 1;
    const int cnt = S.counts[2 + cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) S.counts[cur] = 0; // the work list part A consumed
    const int lane = threadIdx.x % GROUP;
    const int ngroups = gridDim.x * kBlock / GROUP;
    for (int idx = (blockIdx.x * kBlock + threadIdx.x) / GROUP; idx < cnt; idx += ngroups) {
        const int j = S.dl[cur][idx];
        if (lane == 0) { // eight waiters per step: the walk is a chain of loads, the list slots come from one atomic
            int w = S.whead[j];
            while (w >= 0) {
                int buf[8], m = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    buf[q] = w;
                    if (w >= 0) {
                        ++m;
                        w = S.wnext[w];
                    }
                }
                const int at = atomicAdd(&S.counts[nxt], m);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < m) S.wl[nxt][at + q] = buf[q];
            }
        }
        if (S.state[j] != kSeed) continue;
        auto cover = [&](int x) {
            if (x > j && S.state[x] == kUndecided && atomicCAS(&S.state[x], kUndecided, kCovered) == kUndecided)
                S.dl[nxt][atomicAdd(&S.counts[2 + nxt], 1)] = x;
        };
        (void)cover; // (kept as the statement of what a claim is; the loop below does it eight at a time)
        for (int e = fptr[j] + lane; e < fptr[j + 1]; e += GROUP) {
            const int c = fcol[e];
            if (c == j) continue;
            // the sweep only goes through neighbours that are not removed; a removed vertex has no successors.
            // Eight second-hop entries per step (c itself rides in the first one): ids, states and the claiming
            // compare-and-swaps are each in flight together, and the step takes its list slots with one atomic
            const int ke = fptr[c + 1];
            bool first = true;
            for (int k = fptr[c]; k < ke || first; k += 8) {
                int xs[8], st[8];
                bool won[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xs[q] = k + q < ke ? fcol[k + q] : -1;
                if (first) { // fold the first-hop vertex into a free slot of the step, or the last one
                    first = false;
                    int spare = -1;
#pragma unroll
                    for (int q = 7; q >= 0; --q)
                        if (xs[q] < 0 || xs[q] == c) spare = q;
                    if (spare >= 0) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (q == spare) xs[q] = c;
                    } else if (c > j && S.state[c] == kUndecided &&
                               atomicCAS(&S.state[c], kUndecided, kCovered) == kUndecided) {
                        S.dl[nxt][atomicAdd(&S.counts[2 + nxt], 1)] = c;
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) st[q] = xs[q] > j ? S.state[xs[q]] : kGone;
                int m = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    won[q] = st[q] == kUndecided && atomicCAS(&S.state[xs[q]], kUndecided, kCovered) == kUndecided;
                    m += won[q] ? 1 : 0;
                }
                if (m) {
                    int at = atomicAdd(&S.counts[2 + nxt], m);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (won[q]) S.dl[nxt][at++] = xs[q];
                }
            }
        }
    }
}

// ---- the same seed set without rounds ---------------------------------------------------------------------
// A dependency round costs two kernels whose length is a chain of ~10 dependent loads (~40 us at 216^3), and the
// sweep of a 256^3 grid is ~1300 rounds deep.  Here a vertex simply WAITS for what it depends on: waves take the
// vertices in index order (a ticket counter); a lane (agg_wait_kernel) or a group of lanes (agg_wait_slots_kernel)
// walks the earlier two-hop predecessors of its vertex exactly as agg_scan_kernel does (covered ones are passed for
// good, a seed covers it, nothing left makes it a seed) and, at an undecided one, polls that vertex's state -- and its own,
// because a new seed covers the later vertices it reaches in one or two hops at once (agg_wake_kernel's push),
// which is what keeps the chains as short as the rounds' (most vertices are covered before their ticket is drawn
// and never walk anything).  A vertex only ever waits for smaller indices, and tickets are handed to RUNNING waves
// in index order, so the smallest undecided vertex is always with a running group that waits for nobody: no
// deadlock, whatever the grid size and residency.  A hop of the dependency chain costs a poll and a short walk
// instead of a round.  ctrl[0] = ticket, ctrl[1] = abort (set when the time limit passes: chain-like graphs are
// the host sweep's).
// The walk of ONE lane over the earlier two-hop predecessors of v, four first-hop entries and four entries of each
// of their lists per step: the ids, the row pointers, the list entries and the states of a step are four rounds of
// independent loads instead of five rounds PER first-hop entry (a seed must look at all of them: ~35 dependent
// rounds for a 7-point row).  (a0, b0) is the resume position as in agg_scan_kernel (b0 < 0: the first-hop vertex
// itself is still to be looked at).  Returns 0: everything is decided and nothing is a seed (v is a seed),
// 1: blocked (position and blocker updated; everything before it is passed for good), 2: a seed covers v.
__device__ __forceinline__ int agg_walk4(int v, int vb, int deg, const int *__restrict__ pptr,
                                         const int *__restrict__ pcol, const int *state, int &a0, int &b0, int &blocker)
{
    while (a0 < deg) {
        int c[4], cb[4], cl[4], sc[4], s0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = a0 + q < deg ? pcol[vb + a0 + q] : -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool valid = c[q] >= 0 && c[q] != v;
            const bool look = valid && c[q] < v && !(q == 0 && b0 >= 0);
            sc[q] = look ? __hip_atomic_load(&state[c[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kCovered;
            cb[q] = valid ? pptr[c[q]] : 0;
            cl[q] = valid ? pptr[c[q] + 1] - cb[q] : 0;
            s0[q] = (q == 0 && b0 > 0) ? b0 : 0;
        }
        int js[4][4], st[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) js[q][r] = s0[q] + r < cl[q] ? pcol[cb[q] + s0[q] + r] : INT_MAX;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[q][r] = (js[q][r] < v && js[q][r] != c[q])
                               ? __hip_atomic_load(&state[js[q][r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kCovered;
        bool seed = false; // a seed is final: wherever it stands in the step, it covers v
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            seed |= sc[q] == kSeed;
#pragma unroll
            for (int r = 0; r < 4; ++r) seed |= st[q][r] == kSeed;
        }
        if (seed) return 2;
        bool again = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (again) continue;
            if (sc[q] == kUndecided) {
                a0 += q;
                b0 = -1;
                blocker = c[q];
                return 1;
            }
            bool ended = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ended) continue;
                if (js[q][r] >= v) ended = true; // sorted rows: the earlier vertices are a prefix (INT_MAX: list over)
                else if (st[q][r] == kUndecided) {
                    a0 += q;
                    b0 = s0[q] + r;
                    blocker = js[q][r];
                    return 1;
                }
            }
            if (!ended) { // four more entries of this list
                a0 += q;
                b0 = s0[q] + 4;
                again = true;
            }
        }
        if (!again) {
            a0 += 4;
            b0 = -1;
        }
    }
    return 0;
}

__global__ __launch_bounds__(kBlock) void agg_wait_kernel(int n, const int *__restrict__ pptr,
                                                          const int *__restrict__ pcol, const int *__restrict__ fptr,
                                                          const int *__restrict__ fcol, int *__restrict__ state,
                                                          int *__restrict__ ctrl, long long limit_ticks)
{
    // one lane per vertex (narrow rows: stencil-like graphs), 64 vertices per ticket
    const int wlane = threadIdx.x & 63;
    const long long t0 = (long long)wall_clock64();
    for (;;) {
        int base = 0;
        if (wlane == 0) base = atomicAdd(&ctrl[0], 64);
        base = __shfl(base, 0);
        if (base >= n) return;
        const int v = base + wlane;
        // most vertices are covered before their ticket is drawn
        bool active = v < n && __hip_atomic_load(&state[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kUndecided;
        int vb = 0, deg = 0;
        if (active) {
            vb = pptr[v];
            deg = pptr[v + 1] - vb;
        }
        int a0 = 0, b0 = -1, blocker = -1; // resume position of the walk (as S.pa / S.pb), and whom it waits for
        unsigned spins = 0;
        while (__any(active)) {
            bool became_seed = false, moved = false;
            if (active) {
                bool go = true, covered = false;
                if (__hip_atomic_load(&state[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kUndecided) {
                    active = false; // an earlier seed has covered it meanwhile
                    go = false;
                }
                if (go && blocker >= 0) {
                    const int st = __hip_atomic_load(&state[blocker], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (st == kUndecided) go = false;
                    else if (st == kSeed) covered = true;
                    else { // passed for good
                        b0 = b0 < 0 ? 0 : b0 + 1;
                        blocker = -1;
                    }
                }
                moved = go || !active;
                if (go && !covered) {
                    const int w = agg_walk4(v, vb, deg, pptr, pcol, state, a0, b0, blocker);
                    if (w == 2) covered = true;
                    else if (w == 1) go = false;
                }
                if (go) {
                    active = false;
                    // a seed: none of its earlier one- or two-hop predecessors is one
                    __hip_atomic_store(&state[v], covered ? kCovered : kSeed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    became_seed = !covered;
                }
            }
            // the whole wave covers for a new seed -- eight first-hop successors at a time, eight lanes on the list
            // of each (one lane alone would walk ~50 entries while 63 others wait)
            unsigned long long sm = __ballot(became_seed);
            while (sm) {
                const int src = __ffsll((long long)sm) - 1;
                sm &= sm - 1;
                const int sv = __shfl(v, src);
                const int fb = fptr[sv], fe = fptr[sv + 1], row = wlane >> 3, sub = wlane & 7;
                for (int e0 = fb; e0 < fe; e0 += 8) {
                    const int e = e0 + row;
                    const int c = e < fe ? fcol[e] : -1;
                    if (c < 0 || c == sv) continue;
                    if (sub == 0 && c > sv) atomicCAS(&state[c], kUndecided, kCovered);
                    const int ke = fptr[c + 1];
                    for (int k = fptr[c] + sub; k < ke; k += 8) {
                        const int x = fcol[k];
                        if (x > sv && state[x] == kUndecided) atomicCAS(&state[x], kUndecided, kCovered);
                    }
                }
            }
            if ((++spins & 63u) == 0) {
                if ((long long)wall_clock64() - t0 > limit_ticks) ctrl[1] = 1;
                if (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
            // a wave in which nobody moved only polls: thousands of such waves would saturate the L2 with their
            // scattered loads and slow the few that make progress (216^3 level 0: 0.033 s without a nap,
            // 0.032 / 0.030 / 0.025 s with s_sleep 4 / 16 / 64; round 4: 30.6 / 28.0 / 24.8 / 22.7 / 21.7 / 23.5 / 25.9 ms
            // with 16 / 32 / 64 / 127 / 2 x 127 / 4 x 127 / 6 x 127 -- at 256^3 36.3 -> 33.0 ms; the residency of the
            // grid, 4 ... 16 workgroups per CU, makes no difference)
            if (!__any(moved)) {
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
            }
        }
    }
}

// The walk of a GROUP of lanes over the earlier two-hop predecessors of v from the resume position (a0, b0), as
// in agg_scan_kernel: lane l takes the first-hop entries a0 + l, a0 + l + GROUP, ...; the group stops at the first
// first-hop entry (in list order) behind which something is undecided, or as soon as any lane meets a seed.
// Returns 2: covered, 1: blocked (stop_a, stop_b, blk2 set), 0: nothing left (v is a seed).
constexpr int kWalkBatch = 8; // second-hop entries of a lane looked at together (their loads are independent: 4 -> 8 halves the chain)
template <int GROUP>
__device__ __forceinline__ int agg_group_walk(int v, int vb, int deg, int a0, int b0, int lane, int gbase,
                                              unsigned long long gmask, const int *__restrict__ pptr,
                                              const int *__restrict__ pcol, const int *state, int &stop_a, int &stop_b,
                                              int &blk2)
{
    for (int fb = a0; fb < deg; fb += GROUP) {
        const int a = fb + lane;
        int kind = 0, myb = -1, blk = -1; // 0 passed, 1 blocked, 2 met a seed
        if (a < deg) {
            const int c = pcol[vb + a];
            if (c != v) {
                int bb = (a == a0) ? b0 : -1;
                if (bb < 0) {
                    if (c < v) {
                        const int st = __hip_atomic_load(&state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st == kSeed) kind = 2;
                        else if (st == kUndecided) {
                            kind = 1;
                            blk = c;
                        }
                    }
                    bb = 0;
                }
                if (kind == 0) {
                    const int cb = pptr[c], clen = pptr[c + 1] - cb;
                    bool end = false;
                    for (; bb < clen && !end && kind == 0; bb += kWalkBatch) {
                        int js[kWalkBatch], st[kWalkBatch];
#pragma unroll
                        for (int q = 0; q < kWalkBatch; ++q) js[q] = bb + q < clen ? pcol[cb + bb + q] : INT_MAX;
#pragma unroll
                        for (int q = 0; q < kWalkBatch; ++q)
                            st[q] = (js[q] < v && js[q] != c)
                                        ? __hip_atomic_load(&state[js[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : kCovered;
#pragma unroll
                        for (int q = 0; q < kWalkBatch; ++q) {
                            if (end || kind) continue;
                            if (js[q] >= v) end = true; // sorted rows: the earlier vertices are a prefix
                            else if (st[q] == kSeed) kind = 2;
                            else if (st[q] == kUndecided) {
                                kind = 1;
                                blk = js[q];
                                myb = bb + q;
                            }
                        }
                    }
                }
            }
        }
        const unsigned long long seeds = __ballot(kind == 2) & gmask;
        const unsigned long long stops = __ballot(kind == 1) & gmask;
        if (seeds) return 2;
        if (stops) {
            const int first = __ffsll((long long)stops) - 1; // lowest lane = earliest entry
            stop_a = fb + (first - gbase);
            stop_b = __shfl(myb, first);
            blk2 = __shfl(blk, first);
            return 1;
        }
    }
    return 0;
}

// a new seed v covers the later vertices it reaches in one or two hops; GROUP lanes, one first-hop successor each
template <int GROUP>
__device__ __forceinline__ void agg_group_push(int v, int lane, const int *__restrict__ fptr,
                                               const int *__restrict__ fcol, int *state)
{
    for (int e = fptr[v] + lane; e < fptr[v + 1]; e += GROUP) {
        const int c = fcol[e];
        if (c == v) continue;
        if (c > v) atomicCAS(&state[c], kUndecided, kCovered);
        const int ke = fptr[c + 1];
        for (int k = fptr[c]; k < ke; k += 8) {
            int xs[8], st[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xs[q] = k + q < ke ? fcol[k + q] : -1;
#pragma unroll
            for (int q = 0; q < 8; ++q) st[q] = xs[q] > v ? state[xs[q]] : kGone;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (st[q] == kUndecided) atomicCAS(&state[xs[q]], kUndecided, kCovered);
        }
    }
}

// Several lanes per vertex (wide rows): a group of GROUP lanes that held ONE waiting vertex would keep the window of
// the sweep -- the vertices that have been drawn and can be decided as soon as their turn comes -- at a few thousand,
// and the in-order tickets would then serialize a sweep whose frontier is a whole plane of the mesh (Q1 elasticity
// blocks, 27-point graph of 10^6 nodes: 0.13 s against 0.018 s for the rounds).  Here every LANE of the group holds a
// waiting vertex (its resume position and its blocker): one round of loads polls all GROUP of them, the ones whose
// blocker has been decided are walked by the whole group, one after the other, and emptied slots draw new vertices
// from the ticket counter.  The smallest undecided vertex has always been drawn and sits in a slot whose blocker is
// decided (or which has none): no deadlock, as above.
template <int GROUP>
__global__ __launch_bounds__(kBlock) void agg_wait_slots_kernel(int n, const int *__restrict__ pptr,
                                                                const int *__restrict__ pcol,
                                                                const int *__restrict__ fptr,
                                                                const int *__restrict__ fcol, int *__restrict__ state,
                                                                int *__restrict__ ctrl, long long limit_ticks)
{
    const int wlane = threadIdx.x & 63, lane = wlane % GROUP, gbase = wlane / GROUP * GROUP;
    const unsigned long long gmask = GROUP == 64 ? ~0ull : (((1ull << GROUP) - 1ull) << gbase);
    const long long t0 = (long long)wall_clock64();
    int sv = -1, sa0 = 0, sb0 = -1, sblk = -1; // my slot: vertex, resume position, blocker
    bool drained = false;                      // (uniform over the group) the ticket counter has passed n
    unsigned spins = 0;
    for (;;) {
        bool moved = false;
        if (!drained) { // empty slots draw the next vertices
            const unsigned long long empty = __ballot(sv < 0) & gmask;
            const int cnt = __popcll(empty);
            if (cnt) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&ctrl[0], cnt);
                base = __shfl(base, gbase);
                if (base >= n) drained = true;
                else if (sv < 0) {
                    const int v = base + __popcll(empty & ((1ull << wlane) - 1ull));
                    if (v < n && __hip_atomic_load(&state[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kUndecided) {
                        sv = v;
                        sa0 = 0;
                        sb0 = -1;
                        sblk = -1;
                    }
                }
                moved = true;
            }
        }
        bool ready = false;
        if (sv >= 0) { // poll: my vertex (an earlier seed may have covered it) and its blocker
            if (__hip_atomic_load(&state[sv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kUndecided) {
                sv = -1;
                moved = true;
            } else if (sblk < 0) {
                ready = true;
            } else {
                const int st = __hip_atomic_load(&state[sblk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st == kSeed) {
                    __hip_atomic_store(&state[sv], kCovered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sv = -1;
                    moved = true;
                } else if (st != kUndecided) { // passed for good
                    sb0 = sb0 < 0 ? 0 : sb0 + 1;
                    sblk = -1;
                    ready = true;
                }
            }
        }
        unsigned long long rm = __ballot(ready) & gmask;
        while (rm) { // the group walks its ready slots one after the other
            const int src = __ffsll((long long)rm) - 1;
            rm &= rm - 1;
            const int v = __shfl(sv, src), a0 = __shfl(sa0, src), b0 = __shfl(sb0, src);
            const int vb = pptr[v], deg = pptr[v + 1] - vb;
            int stop_a = -1, stop_b = -1, blk2 = -1;
            const int w = agg_group_walk<GROUP>(v, vb, deg, a0, b0, lane, gbase, gmask, pptr, pcol, state, stop_a, stop_b, blk2);
            if (w == 1) {
                if (wlane == src) {
                    sa0 = stop_a;
                    sb0 = stop_b;
                    sblk = blk2;
                }
            } else {
                if (lane == 0)
                    __hip_atomic_store(&state[v], w == 2 ? kCovered : kSeed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w == 0) agg_group_push<GROUP>(v, lane, fptr, fcol, state);
                if (wlane == src) sv = -1;
            }
            moved = true;
        }
        if (__all(drained && sv < 0)) return;
        if ((++spins & 63u) == 0) {
            if ((long long)wall_clock64() - t0 > limit_ticks) ctrl[1] = 1;
            if (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
        if (!__any(moved)) __builtin_amdgcn_s_sleep(127); // (64 ... 2 x 127: 14.7 ... 14.3 ms on level 1 of the 216^3 hierarchy)
    }
}

__global__ __launch_bounds__(kBlock) void agg_init_state_kernel(int n, const int *__restrict__ id0, int *__restrict__ state)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock)
        state[v] = id0[v] != -1 ? kGone : kUndecided;
}

__global__ __launch_bounds__(kBlock) void agg_seed_flags_kernel(int n, const int *__restrict__ state,
                                                                 int *__restrict__ flag)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) flag[v] = state[v] == kSeed;
}

__global__ __launch_bounds__(kBlock) void agg_assign_kernel(int n, const int *__restrict__ sptr,
                                                             const int *__restrict__ scol,
                                                             const int *__restrict__ state,
                                                             const int *__restrict__ rank, int *__restrict__ id)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        const int st = state[v];
        if (st == kGone) {
            id[v] = -2;
            continue;
        }
        int best = -1;
        for (int j = sptr[v]; j < sptr[v + 1]; ++j) {
            const int c = scol[j];
            if (c != v && state[c] == kSeed) best = max(best, c); // claims overwrite: the last claiming seed wins
        }
        if (best < 0 && st == kSeed) best = v; // nobody claimed it: a seed keeps itself
        if (best < 0) {
            int first = INT_MAX; // tentative claims do not overwrite: the first seed at distance two wins
            for (int j = sptr[v]; j < sptr[v + 1]; ++j) {
                const int c = scol[j];
                if (c == v) continue;
                for (int k = sptr[c]; k < sptr[c + 1]; ++k) {
                    const int s = scol[k];
                    if (s != c && state[s] == kSeed) first = min(first, s);
                }
            }
            best = first;
        }
        id[v] = best == INT_MAX ? -1 : rank[best];
    }
}

// the same rule with G lanes per vertex (wide rows: level 1 of the 216^3 hierarchy, 961 two-hop entries per vertex, 3.7 ms with
// one lane): the lanes stride the first hop, maximum / minimum over the group -- order-independent, the same aggregates
template <int G>
__global__ __launch_bounds__(kBlock) void agg_assign_group_kernel(int n, const int *__restrict__ sptr,
                                                                   const int *__restrict__ scol,
                                                                   const int *__restrict__ state,
                                                                   const int *__restrict__ rank, int *__restrict__ id)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        const int st = state[v]; // (uniform over the group)
        if (st == kGone) {
            if (lane == 0) id[v] = -2;
            continue;
        }
        const int b = sptr[v], e = sptr[v + 1];
        int best = -1;
        for (int j = b + lane; j < e; j += G) {
            const int c = scol[j];
            if (c != v && state[c] == kSeed) best = max(best, c);
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, G));
        if (best < 0 && st == kSeed) best = v;
        if (best < 0) {
            int first = INT_MAX;
            for (int j = b; j < e; ++j) { // (the group walks the first hop together, its lanes stride each second-hop row)
                const int c = scol[j];
                if (c == v) continue;
                const int ke = sptr[c + 1];
                for (int k = sptr[c] + lane; k < ke; k += G) {
                    const int s = scol[k];
                    if (s != c && state[s] == kSeed) first = min(first, s);
                }
            }
#pragma unroll
            for (int off = G >> 1; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, G));
            best = first;
        }
        if (lane == 0) id[v] = best == INT_MAX ? -1 : rank[best];
    }
}

// The membership rule in two one-hop passes (round 5) instead of a two-hop walk per vertex (49 loads for a 7-point row, 961 on
// level 1 of the 216^3 hierarchy): pass 1 records for every vertex the largest and the smallest seed NEXT to it; pass 2: a
// vertex next to a seed takes the largest, a seed nobody claims keeps itself, everybody else takes the smallest of its
// neighbours' smallest seeds -- the first seed two hops away.  Maximum and minimum are order-independent: the same aggregates.
template <int G>
__global__ __launch_bounds__(kBlock) void agg_seed_range_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                                 const int *__restrict__ state, int *__restrict__ smax,
                                                                 int *__restrict__ smin)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        int hi = -1, lo = INT_MAX;
        if (state[v] != kGone) {
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e; j += G) {
                const int c = scol[j];
                if (c != v && state[c] == kSeed) {
                    hi = max(hi, c);
                    lo = min(lo, c);
                }
            }
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            hi = max(hi, __shfl_xor(hi, off, G));
            lo = min(lo, __shfl_xor(lo, off, G));
        }
        if (lane == 0) {
            smax[v] = hi;
            smin[v] = lo;
        }
    }
}

template <int G>
__global__ __launch_bounds__(kBlock) void agg_assign2_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                              const int *__restrict__ state, const int *__restrict__ smax,
                                                              const int *__restrict__ smin, const int *__restrict__ rank,
                                                              int *__restrict__ id)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        const int st = state[v]; // (uniform over the group)
        if (st == kGone) {
            if (lane == 0) id[v] = -2;
            continue;
        }
        int best = smax[v];
        if (best < 0 && st == kSeed) best = v;
        if (best < 0) {
            int first = INT_MAX;
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e; j += G) {
                const int c = scol[j];
                if (c != v) first = min(first, smin[c]);
            }
#pragma unroll
            for (int off = G >> 1; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, G));
            best = first;
        }
        if (lane == 0) id[v] = best == INT_MAX ? -1 : rank[best];
    }
}

// ---- "amg.aggregation" = "parallel" (round 5; oracle: parallel_aggregates_graph) ---------------------------------------
// The seeds as the distance-2 maximal independent set by hashed priorities, in synchronous rounds: an undecided vertex
// whose key is the largest among the undecided vertices within two hops becomes a seed; everything within two hops of a
// seed is covered.  A dozen rounds whatever the mesh, four one-hop passes each; integer work, the oracle's seeds exactly.
__device__ __forceinline__ unsigned agg_hash32(unsigned v)
{
    v ^= v >> 16;
    v *= 0x7feb352du;
    v ^= v >> 15;
    v *= 0x846ca68bu;
    v ^= v >> 16;
    return v;
}
// is the key of a larger than the key of b?  (-1: no vertex)
__device__ __forceinline__ bool agg_key_greater(int a, int b)
{
    if (a < 0) return false;
    if (b < 0) return true;
    const unsigned ha = agg_hash32((unsigned)a), hb = agg_hash32((unsigned)b);
    return ha > hb || (ha == hb && a > b);
}

// m1[v] = the undecided vertex of the largest key in the closed neighbourhood of v (-1: none)
// (quiet[v]: v is decided and so are all its neighbours -- nothing passes through it any more, its m1 stays -1 and its
// "next to a seed" flag is final: the later rounds, in which most vertices are quiet, skip their rows)
template <int G>
__global__ __launch_bounds__(kBlock) void mis_max1_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                           const int *__restrict__ state, int *__restrict__ m1,
                                                           unsigned char *__restrict__ quiet)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        if (quiet[v]) continue; // (uniform over the group)
        const int st = state[v];
        int m = -1;
        if (st != kGone) {
            if (st == kUndecided) m = v;
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e; j += G) {
                const int u = scol[j];
                if (u != v && state[u] == kUndecided && agg_key_greater(u, m)) m = u;
            }
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            const int o = __shfl_xor(m, off, G);
            if (agg_key_greater(o, m)) m = o;
        }
        if (lane == 0) {
            m1[v] = m;
            if (m < 0) quiet[v] = 1;
        }
    }
}

// an undecided vertex that holds the largest key within two hops becomes a seed
template <int G>
__global__ __launch_bounds__(kBlock) void mis_seed_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                           const int *__restrict__ m1, int *__restrict__ state)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        if (state[v] != kUndecided) continue; // (uniform over the group; only v's own thread group writes state[v])
        int m = m1[v];
        const int e = sptr[v + 1];
        for (int j = sptr[v] + lane; j < e; j += G) {
            const int u = scol[j];
            if (u == v) continue;
            const int o = m1[u];
            if (agg_key_greater(o, m)) m = o;
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            const int o = __shfl_xor(m, off, G);
            if (agg_key_greater(o, m)) m = o;
        }
        if (lane == 0 && m == v) state[v] = kSeed;
    }
}

// c1[v] = v is a seed or next to one
template <int G>
__global__ __launch_bounds__(kBlock) void mis_near_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                           const int *__restrict__ state, unsigned char *__restrict__ c1,
                                                           const unsigned char *__restrict__ quiet)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        if (c1[v] || quiet[v]) continue; // seeds stay seeds; a quiet vertex gets no new seed next to it
        const int st = state[v];
        int c = st == kSeed;
        if (!c && st != kGone) {
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e && !c; j += G) c = state[scol[j]] == kSeed;
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) c |= __shfl_xor(c, off, G);
        if (lane == 0) c1[v] = (unsigned char)c;
    }
}

// an undecided vertex within two hops of a seed is covered; left[0] += the vertices still undecided
template <int G>
__global__ __launch_bounds__(kBlock) void mis_cover_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                            const unsigned char *__restrict__ c1, int *__restrict__ state,
                                                            int *__restrict__ left)
{
    const int lane = threadIdx.x % G;
    int mine = 0;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        if (state[v] != kUndecided) continue;
        int c = c1[v];
        if (!c) {
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e && !c; j += G) c = c1[scol[j]];
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) c |= __shfl_xor(c, off, G);
        if (lane == 0) {
            if (c) state[v] = kCovered;
            else ++mine;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(left, mine);
}


// ---- "amg.aggregation" = "compact" (round 6; oracle: compact_aggregates_graph, host: compact_sweep) ----------------------
// One-hop aggregates around two generations of hashed-priority distance-2 independent sets (the rounds above, run twice: on
// the whole graph, then on the subgraph of the leftovers among the candidates), the rest by most connections.
__global__ __launch_bounds__(kBlock) void compact_init_owner_kernel(int n, const int *__restrict__ state, int *__restrict__ owner)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) owner[v] = state[v] == kGone ? -2 : -1;
}

// steps B / E: a seed of the current graph keeps itself, a vertex next to one joins it (the largest index if there are several)
template <int G>
__global__ __launch_bounds__(kBlock) void compact_claim_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                                const int *__restrict__ state, int *__restrict__ owner)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        const int st = state[v]; // (uniform over the group)
        if (st == kGone) continue;
        int best = -1;
        if (st == kSeed) best = v;
        else {
            const int e = sptr[v + 1];
            for (int j = sptr[v] + lane; j < e; j += G) {
                const int u = scol[j];
                if (u != v && state[u] == kSeed) best = max(best, u);
            }
#pragma unroll
            for (int off = G >> 1; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, G));
        }
        if (lane == 0 && best >= 0) owner[v] = best;
    }
}

// step C: the graph of the leftovers (everything assigned or removed leaves it) and its candidates -- a leftover whose
// leftover neighbours are at least 3/5 of its strong neighbours competes, the others only pass keys on
template <int G>
__global__ __launch_bounds__(kBlock) void compact_candidates_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                                     const int *__restrict__ owner, int *__restrict__ state)
{
    const int lane = threadIdx.x % G;
    for (int v = (blockIdx.x * kBlock + threadIdx.x) / G; v < n; v += gridDim.x * (kBlock / G)) {
        if (owner[v] != -1) { // (uniform over the group)
            if (lane == 0) state[v] = kGone;
            continue;
        }
        int deg = 0, lo = 0;
        const int e = sptr[v + 1];
        for (int j = sptr[v] + lane; j < e; j += G) {
            const int u = scol[j];
            if (u == v) continue;
            ++deg;
            lo += owner[u] == -1;
        }
#pragma unroll
        for (int off = G >> 1; off > 0; off >>= 1) {
            deg += __shfl_xor(deg, off, G);
            lo += __shfl_xor(lo, off, G);
        }
        if (lane == 0) state[v] = (lo > 0 && 5 * lo >= 3 * deg) ? kUndecided : kCovered;
    }
}

// step F, one synchronous pass: an unassigned vertex joins the aggregate it has the most strong connections to (ties: the
// smaller seed); counts[0] += vertices still unassigned, counts[1] += vertices assigned in this pass
__global__ __launch_bounds__(kBlock) void compact_join_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                               const int *__restrict__ owner, int *__restrict__ next,
                                                               int *__restrict__ counts)
{
    int left = 0, moved = 0;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        const int o0 = owner[v];
        int out = o0;
        if (o0 == -1) {
            int best = -1, bc = 0;
            const int b = sptr[v], e = sptr[v + 1];
            for (int j = b; j < e; ++j) {
                const int u = scol[j];
                if (u == v) continue;
                const int o = owner[u];
                if (o < 0 || o == best) continue;
                int c = 0;
                for (int k = b; k < e; ++k) {
                    const int w = scol[k];
                    c += (w != v && owner[w] == o);
                }
                if (c > bc || (c == bc && o < best)) {
                    bc = c;
                    best = o;
                }
            }
            if (best >= 0) {
                out = best;
                ++moved;
            } else ++left;
        }
        next[v] = out;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        left += __shfl_xor(left, off);
        moved += __shfl_xor(moved, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (left) atomicAdd(&counts[0], left);
        if (moved) atomicAdd(&counts[1], moved);
    }
}

// what is still unassigned seeds its own aggregate (unsymmetric patterns only); flag[v] = v is the seed of an aggregate
__global__ __launch_bounds__(kBlock) void compact_seed_flags_kernel(int n, int *__restrict__ owner, int *__restrict__ flag)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        int o = owner[v];
        if (o == -1) owner[v] = o = v;
        flag[v] = o == v;
    }
}

__global__ __launch_bounds__(kBlock) void compact_ids_kernel(int n, const int *__restrict__ owner, const int *__restrict__ rank,
                                                              int *__restrict__ id)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock) {
        const int o = owner[v];
        id[v] = o == -2 ? -2 : rank[o];
    }
}

// unsymmetric patterns: aggregates whose members were all claimed by later seeds disappear
__global__ __launch_bounds__(kBlock) void agg_mark_used_kernel(int n, const int *__restrict__ id, int *__restrict__ used)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock)
        if (id[v] >= 0) used[id[v]] = 1;
}

__global__ __launch_bounds__(kBlock) void agg_renumber_kernel(int n, const int *__restrict__ newid, int *__restrict__ id)
{
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < n; v += gridDim.x * kBlock)
        if (id[v] >= 0) id[v] = newid[id[v]];
}

} // namespace

// Returns the aggregate count and fills id[n] (device), or -1 when the graph does not qualify (not
// symmetric / not sorted) or the dependency chains exceed max_rounds: the caller then runs the host sweep.
int64_t device_aggregate(const Launch &L, int n, const int *sptr, const int *scol, const int *id0, int *id,
                         int max_rounds, AggregateScratch &W, SymbolicScratch &S, int *rounds_out, int mode)
{
    hipStream_t s = L.stream;
    S.counters.ensure(16);
    S.host.ensure(16);
    PS_HIP_CHECK(hipMemsetAsync(S.counters.ptr, 0, 16 * sizeof(int), s));
    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, sptr + n, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const bool wide = (double)*reinterpret_cast<const int *>(S.host.ptr) > 12.0 * (double)std::max(1, n);
    if (mode != 3 && mode != 4) { // ("parallel" / "compact" are defined on the graph as given: nothing to check)
        if (wide) hipLaunchKernelGGL((graph_check_group_kernel<16>), dim3(L.grid), dim3(kBlock), 0, s, n, sptr, scol, S.counters.ptr);
        else hipLaunchKernelGGL(graph_check_kernel, dim3(L.grid), dim3(kBlock), 0, s, n, sptr, scol, S.counters.ptr);
        PS_HIP_CHECK(hipGetLastError());
        PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, S.counters.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
    }
    const int *fptr = sptr, *fcol = scol; // successors: the strength graph as given
    // predecessor lists: the graph itself when it is symmetric with sorted rows, else its transpose
    // ("parallel", mode 3, is DEFINED on the graph as given -- the oracle's passes read the stored rows, whatever their
    // symmetry: a Galerkin operator whose entries cancel to an exact zero on one side of the diagonal only still gives
    // the oracle's aggregates)
    const bool transposed = mode != 3 && mode != 4 && *reinterpret_cast<const int *>(S.host.ptr) != 0;
    if (transposed) {
        int64_t nnz = 0;
        PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, sptr + n, sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        nnz = *reinterpret_cast<const int *>(S.host.ptr);
        device_transpose_pattern(L, n, n, sptr, scol, nnz, W.tptr, W.tcol, W.tmap, S);
        sptr = W.tptr.ptr;
        scol = W.tcol.ptr;
    }

    PS_HIP_CHECK(hipMemcpyAsync(S.host.ptr, sptr + n, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const double avg_degree = (double)*reinterpret_cast<const int *>(S.host.ptr) / std::max(1, n);
    const size_t N = (size_t)n + 1;
    W.ints.ensure(9 * N + 16);
    const dim3 g(L.grid), blk(kBlock);
    bool done = false;
    int round = 0;
    AggState A;
    A.state = W.ints.ptr;
    A.pa = A.state + N;
    A.pb = A.pa + N;
    if (mode == 3 || mode == 4) {
        // "parallel" / "compact": hashed-priority distance-2 independent set, synchronous rounds
        int *left = S.counters.ptr + 8;
        int *m1 = A.pa;
        unsigned char *c1 = reinterpret_cast<unsigned char *>(A.pb), *quiet = c1 + N;
        hipLaunchKernelGGL(agg_init_state_kernel, g, blk, 0, s, n, id0, A.state);
        int *hc = reinterpret_cast<int *>(S.host.ptr);
        const bool widem = avg_degree > 12.0;
        // the rounds on the graph A.state describes (kUndecided compete, kCovered pass keys on, kGone are not part of it)
        auto mis_rounds = [&]() {
            PS_HIP_CHECK(hipMemsetAsync(c1, 0, 2 * N, s));
            bool fin = false;
            for (int r = 0; r < 64 && !fin; ++r, ++round) {
                PS_HIP_CHECK(hipMemsetAsync(left, 0, sizeof(int), s));
                if (widem && avg_degree > 24.0) {
                    hipLaunchKernelGGL((mis_max1_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, m1, quiet);
                    hipLaunchKernelGGL((mis_seed_kernel<16>), g, blk, 0, s, n, sptr, scol, m1, A.state);
                    hipLaunchKernelGGL((mis_near_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, c1, quiet);
                    hipLaunchKernelGGL((mis_cover_kernel<16>), g, blk, 0, s, n, sptr, scol, c1, A.state, left);
                } else if (widem) {
                    hipLaunchKernelGGL((mis_max1_kernel<8>), g, blk, 0, s, n, sptr, scol, A.state, m1, quiet);
                    hipLaunchKernelGGL((mis_seed_kernel<8>), g, blk, 0, s, n, sptr, scol, m1, A.state);
                    hipLaunchKernelGGL((mis_near_kernel<8>), g, blk, 0, s, n, sptr, scol, A.state, c1, quiet);
                    hipLaunchKernelGGL((mis_cover_kernel<8>), g, blk, 0, s, n, sptr, scol, c1, A.state, left);
                } else {
                    hipLaunchKernelGGL((mis_max1_kernel<1>), g, blk, 0, s, n, sptr, scol, A.state, m1, quiet);
                    hipLaunchKernelGGL((mis_seed_kernel<1>), g, blk, 0, s, n, sptr, scol, m1, A.state);
                    hipLaunchKernelGGL((mis_near_kernel<1>), g, blk, 0, s, n, sptr, scol, A.state, c1, quiet);
                    hipLaunchKernelGGL((mis_cover_kernel<1>), g, blk, 0, s, n, sptr, scol, c1, A.state, left);
                }
                PS_HIP_CHECK(hipGetLastError());
                PS_HIP_CHECK(hipMemcpyAsync(hc, left, sizeof(int), hipMemcpyDeviceToHost, s));
                PS_HIP_CHECK(hipStreamSynchronize(s));
                fin = hc[0] == 0;
            }
            return fin;
        };
        done = mis_rounds();
        if (rounds_out) *rounds_out = round;
        if (!done) return -1;
        if (mode == 4) {
            // "compact": one-hop aggregates, a second generation of seeds among the leftovers, the rest by most connections
            int *owner = A.pb + N, *next = owner + N, *flag = next + N;
            auto claim = [&]() {
                if (widem) hipLaunchKernelGGL((compact_claim_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, owner);
                else hipLaunchKernelGGL((compact_claim_kernel<1>), g, blk, 0, s, n, sptr, scol, A.state, owner);
            };
            hipLaunchKernelGGL(compact_init_owner_kernel, g, blk, 0, s, n, A.state, owner);
            claim();
            if (widem) hipLaunchKernelGGL((compact_candidates_kernel<16>), g, blk, 0, s, n, sptr, scol, owner, A.state);
            else hipLaunchKernelGGL((compact_candidates_kernel<1>), g, blk, 0, s, n, sptr, scol, owner, A.state);
            PS_HIP_CHECK(hipGetLastError());
            done = mis_rounds();
            if (rounds_out) *rounds_out = round;
            if (!done) return -1;
            claim();
            for (int pass = 0; pass < 8; ++pass) {
                PS_HIP_CHECK(hipMemsetAsync(left, 0, 2 * sizeof(int), s));
                hipLaunchKernelGGL(compact_join_kernel, g, blk, 0, s, n, sptr, scol, owner, next, left);
                PS_HIP_CHECK(hipGetLastError());
                PS_HIP_CHECK(hipMemcpyAsync(hc, left, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
                PS_HIP_CHECK(hipStreamSynchronize(s));
                std::swap(owner, next);
                if (hc[0] == 0 || hc[1] == 0) break;
            }
            hipLaunchKernelGGL(compact_seed_flags_kernel, g, blk, 0, s, n, owner, flag);
            PS_HIP_CHECK(hipGetLastError());
            const int64_t nagg = device_exclusive_scan(L, flag, n, S);
            hipLaunchKernelGGL(compact_ids_kernel, g, blk, 0, s, n, owner, flag, id);
            PS_HIP_CHECK(hipGetLastError());
            return nagg;
        }
    } else if (mode == 2) {
        // no rounds: every vertex waits for the earlier vertices it depends on (agg_wait_kernel)
        int *ctrl = S.counters.ptr + 8;
        PS_HIP_CHECK(hipMemsetAsync(ctrl, 0, 8 * sizeof(int), s));
        hipLaunchKernelGGL(agg_init_state_kernel, g, blk, 0, s, n, id0, A.state);
        // time limit in the place of the round budget: 10 us per allowed round (100 MHz counter)
        // (+ 10 ns per vertex: the sheer volume of a very large level is not a long chain)
        const long long limit_ticks = (long long)max_rounds * 1000ll + (long long)n;
        const int lanes = avg_degree <= 8.0 ? 1 : (avg_degree <= 16.0 ? 8 : 32);
        // resident workgroups per CU: every resident wave that is not at the frontier of the sweep only polls
        // (216^3 level 0, one lane per vertex: 0.040 / 0.031 / 0.026 / 0.025 / 0.026 s with 1 / 2 / 4 / 8 / 16;
        // level 1, 32 lanes per vertex and a slot per lane: 0.019 / 0.039 s with 2 / 8; Q1 elasticity blocks,
        // 10^6 nodes: 0.014 / 0.020 s)
        const int wgs_per_cu = lanes == 1 ? 8 : 2;
        const int nwg = std::max(8, std::min((n + kBlock - 1) / kBlock, L.num_cus * wgs_per_cu));
        if (lanes == 32)
            hipLaunchKernelGGL((agg_wait_slots_kernel<32>), dim3(nwg), blk, 0, s, n, sptr, scol, fptr, fcol, A.state, ctrl, limit_ticks);
        else if (lanes == 8)
            hipLaunchKernelGGL((agg_wait_slots_kernel<8>), dim3(nwg), blk, 0, s, n, sptr, scol, fptr, fcol, A.state, ctrl, limit_ticks);
        else
            hipLaunchKernelGGL(agg_wait_kernel, dim3(nwg), blk, 0, s, n, sptr, scol, fptr, fcol, A.state, ctrl, limit_ticks);
        PS_HIP_CHECK(hipGetLastError());
        int *hc = reinterpret_cast<int *>(S.host.ptr);
        PS_HIP_CHECK(hipMemcpyAsync(hc, ctrl, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        done = hc[1] == 0;
        if (rounds_out) *rounds_out = 0;
        if (!done) return -1;
    } else {
    A.whead = A.pb + N;
    A.wnext = A.whead + N;
    A.wl[0] = A.wnext + N;
    A.wl[1] = A.wl[0] + N;
    A.dl[0] = A.wl[1] + N;
    A.dl[1] = A.dl[0] + N;
    A.counts = S.counters.ptr + 8;
    PS_HIP_CHECK(hipMemsetAsync(A.counts, 0, 8 * sizeof(int), s));
    hipLaunchKernelGGL(agg_init_kernel, g, blk, 0, s, n, id0, A);
    PS_HIP_CHECK(hipGetLastError());
    int *hc = reinterpret_cast<int *>(S.host.ptr);
    PS_HIP_CHECK(hipMemcpyAsync(hc, A.counts, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    const int active = hc[0];
    // lanes per vertex: 1 for stencil-like graphs (7 entries per row), 8 / 32 for wider rows
    const int lanes = avg_degree <= 8.0 ? 1 : (avg_degree <= 16.0 ? 8 : 32);
    done = active == 0;
    const dim3 gr(std::min(L.grid, 1024));
    while (!done && round < max_rounds) {
        const int batch = round < 64 ? 16 : 256; // early rounds are the big ones; afterwards a check is a sync
        // (more lanes per vertex on the short lists of the late rounds were tried and are slower: one lane stops at
        // its first blocker, eight lanes read all the first-hop lists -- 216^3 level 0: 0.081 s against 0.043 s)
        const int lanes_now = lanes;
        for (int k = 0; k < batch; ++k, ++round) {
            if (lanes_now == 32) {
                hipLaunchKernelGGL(agg_scan_kernel<32>, gr, blk, 0, s, round, sptr, scol, A);
                hipLaunchKernelGGL(agg_wake_kernel<32>, gr, blk, 0, s, round, fptr, fcol, A);
            } else if (lanes_now == 8) {
                hipLaunchKernelGGL(agg_scan_kernel<8>, gr, blk, 0, s, round, sptr, scol, A);
                hipLaunchKernelGGL(agg_wake_kernel<8>, gr, blk, 0, s, round, fptr, fcol, A);
            } else {
                hipLaunchKernelGGL(agg_scan_kernel<1>, gr, blk, 0, s, round, sptr, scol, A);
                hipLaunchKernelGGL(agg_wake_kernel<1>, gr, blk, 0, s, round, fptr, fcol, A);
            }
        }
        PS_HIP_CHECK(hipGetLastError());
        PS_HIP_CHECK(hipMemcpyAsync(hc, A.counts, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
        const long long decided = (long long)hc[4] + hc[2] + hc[3]; // [4] lags by the round not folded yet
        done = decided >= active;
        // progress check: at this pace, would the rounds exceed the budget?  (a 1-D chain decides ~3 vertices
        // per round: give up after a few hundred rounds instead of burning the whole budget)
        if (!done && round >= 256 && (double)round * (double)active > 1.5 * (double)max_rounds * (double)std::max(1ll, decided))
            break;
    }
    if (rounds_out) *rounds_out = round;
    if (!done) return -1;
    } // (mode)
    // aggregate numbers = rank among the seeds; then the membership rule
    int *rank = A.pa; // the scan state is no longer needed
    hipLaunchKernelGGL(agg_seed_flags_kernel, g, blk, 0, s, n, A.state, rank);
    PS_HIP_CHECK(hipGetLastError());
    int64_t nagg = device_exclusive_scan(L, rank, n, S);
    if (L.lab.agg_two_pass_assign) {
        // (scratch behind the scan state: smax / smin live where the wait lists of the round-based variant would)
        int *smax = A.pb + N, *smin = smax + N;
        if (avg_degree > 12.0) {
            hipLaunchKernelGGL((agg_seed_range_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, smax, smin);
            hipLaunchKernelGGL((agg_assign2_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, smax, smin, rank, id);
        } else {
            hipLaunchKernelGGL((agg_seed_range_kernel<1>), g, blk, 0, s, n, sptr, scol, A.state, smax, smin);
            hipLaunchKernelGGL((agg_assign2_kernel<1>), g, blk, 0, s, n, sptr, scol, A.state, smax, smin, rank, id);
        }
    } else if (avg_degree > 12.0) hipLaunchKernelGGL((agg_assign_group_kernel<16>), g, blk, 0, s, n, sptr, scol, A.state, rank, id);
    else hipLaunchKernelGGL(agg_assign_kernel, g, blk, 0, s, n, sptr, scol, A.state, rank, id);
    PS_HIP_CHECK(hipGetLastError());
    if (transposed && nagg > 0) {
        int *used = A.pb;
        PS_HIP_CHECK(hipMemsetAsync(used, 0, ((size_t)nagg + 1) * sizeof(int), s));
        hipLaunchKernelGGL(agg_mark_used_kernel, g, blk, 0, s, n, id, used);
        PS_HIP_CHECK(hipGetLastError());
        const int64_t kept = device_exclusive_scan(L, used, nagg, S);
        if (kept != nagg) {
            hipLaunchKernelGGL(agg_renumber_kernel, g, blk, 0, s, n, used, id);
            PS_HIP_CHECK(hipGetLastError());
            nagg = kept;
        }
    }
    return nagg;
}

} // namespace psolve
