// amg_setup.cpp -- host-side smoothed-aggregation hierarchy (see amg_setup.hpp).
#include "amg_setup.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <utility>

#include "common.hpp"
#include "solver.hpp"

namespace psolve {

namespace {

int host_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    return (int)std::min<unsigned>(hw, 32u);
}

// fn(thread_index, begin, end) over [0, n) in contiguous chunks
void parallel_chunks(int64_t n, const std::function<void(int, int64_t, int64_t)> &fn)
{
    int T = host_threads();
    if (n < 20000) T = 1;
    if (T <= 1) {
        fn(0, 0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const int64_t b = t * chunk, e = std::min<int64_t>(n, b + chunk);
        if (b >= e) break;
        th.emplace_back(fn, t, b, e);
    }
    for (auto &x : th) x.join();
}

void exclusive_scan_rows(std::vector<int32_t> &ptr)
{
    // ptr[i+1] holds the count of row i on entry
    int64_t run = 0;
    for (size_t i = 1; i < ptr.size(); ++i) {
        run += ptr[i];
        PS_REQUIRE(run < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
        ptr[i] = (int32_t)run;
    }
}

// The greedy sweep of amgcl/coarsening/plain_aggregates.hpp.  Order-dependent, hence sequential (as in
// AMGCL).  id[] holds kUndefined / kRemoved on entry, aggregate numbers (or kRemoved) on exit.
template <class Graph>
int64_t greedy_sweep(int64_t n, const Graph &G, std::vector<int32_t> &id)
{
    constexpr int32_t kUndefined = -1, kRemoved = -2;
    std::vector<int32_t> neib;
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (id[i] != kUndefined) continue;
        const int32_t cur = (int32_t)count++;
        id[i] = cur;
        neib.clear();
        for (int32_t j = G.begin(i); j < G.end(i); ++j) {
            const int32_t c = G.col(j);
            if (G.is_strong(i, j) && id[c] != kRemoved) { // also steals members of earlier aggregates
                id[c] = cur;
                neib.push_back(c);
            }
        }
        for (int32_t c : neib)
            for (int32_t j = G.begin(c); j < G.end(c); ++j) {
                const int32_t cc = G.col(j);
                if (G.is_strong(c, j) && id[cc] == kUndefined) id[cc] = cur;
            }
    }
    if (count == 0) return 0;
    // aggregates emptied by later seeds disappear: renumber
    std::vector<int32_t> cnt((size_t)count, 0);
    for (int64_t i = 0; i < n; ++i)
        if (id[i] >= 0) cnt[id[i]] = 1;
    for (int64_t k = 1; k < count; ++k) cnt[k] += cnt[k - 1];
    if (count > cnt[count - 1]) {
        for (int64_t i = 0; i < n; ++i)
            if (id[i] >= 0) id[i] = cnt[id[i]] - 1;
        count = cnt[count - 1];
    }
    return count;
}

// "amg.aggregation" = "parallel" on the host (round 5; the oracle's parallel_aggregates_graph, the device's mis_* kernels):
// seeds = the distance-2 maximal independent set by hashed priorities in synchronous rounds, membership by the sweep's
// closed form (next to a seed: the largest such seed; else the smallest seed two hops away), aggregates in seed order.
inline uint32_t agg_hash32(uint32_t v)
{
    v ^= v >> 16;
    v *= 0x7feb352du;
    v ^= v >> 15;
    v *= 0x846ca68bu;
    v ^= v >> 16;
    return v;
}
inline uint64_t agg_key(int64_t v) { return ((uint64_t)agg_hash32((uint32_t)v) << 32) | (uint32_t)v; }

template <class Graph>
int64_t parallel_sweep(int64_t n, const Graph &G, std::vector<int32_t> &id)
{
    constexpr int32_t kUndefined = -1, kRemoved = -2;
    enum : char { U = 0, S = 1, C = 2, Gn = 3 };
    std::vector<char> st((size_t)n), c1((size_t)n);
    std::vector<uint64_t> m1((size_t)n), m2((size_t)n);
    int64_t undecided = 0;
    for (int64_t i = 0; i < n; ++i) {
        st[i] = id[i] == kUndefined ? U : Gn;
        undecided += st[i] == U;
    }
    while (undecided > 0) {
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                uint64_t m = st[v] == U ? agg_key(v) : 0;
                for (int32_t j = G.begin(v); j < G.end(v); ++j)
                    if (G.is_strong(v, j) && st[G.col(j)] == U) m = std::max(m, agg_key(G.col(j)));
                m1[v] = m;
            }
        });
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                uint64_t m = m1[v];
                for (int32_t j = G.begin(v); j < G.end(v); ++j)
                    if (G.is_strong(v, j)) m = std::max(m, m1[G.col(j)]);
                m2[v] = m;
            }
        });
        for (int64_t v = 0; v < n; ++v)
            if (st[v] == U && m2[v] == agg_key(v)) st[v] = S;
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                char c = st[v] == S;
                for (int32_t j = G.begin(v); j < G.end(v) && !c; ++j)
                    if (G.is_strong(v, j) && st[G.col(j)] == S) c = 1;
                c1[v] = c;
            }
        });
        int64_t left = 0;
        for (int64_t v = 0; v < n; ++v) {
            if (st[v] != U) continue;
            char c = c1[v];
            for (int32_t j = G.begin(v); j < G.end(v) && !c; ++j)
                if (G.is_strong(v, j) && c1[G.col(j)]) c = 1;
            if (c) st[v] = C;
            else ++left;
        }
        undecided = left;
    }
    std::vector<int32_t> rank((size_t)n, -1);
    int64_t count = 0;
    for (int64_t v = 0; v < n; ++v)
        if (st[v] == S) rank[v] = (int32_t)count++;
    parallel_chunks(n, [&](int, int64_t b, int64_t e) {
        for (int64_t v = b; v < e; ++v) {
            if (st[v] == Gn) {
                id[v] = kRemoved;
                continue;
            }
            int64_t best = -1;
            for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                const int32_t c = G.col(j);
                if (G.is_strong(v, j) && c != v && st[c] == S) best = std::max<int64_t>(best, c);
            }
            if (best < 0 && st[v] == S) best = v;
            if (best < 0) {
                int64_t first = INT64_MAX;
                for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                    const int32_t c = G.col(j);
                    if (!G.is_strong(v, j) || c == v) continue;
                    for (int32_t k = G.begin(c); k < G.end(c); ++k) {
                        const int32_t s2 = G.col(k);
                        if (G.is_strong(c, k) && s2 != c && st[s2] == S) first = std::min<int64_t>(first, s2);
                    }
                }
                best = first;
            }
            id[v] = best == INT64_MAX ? kUndefined : rank[best];
        }
    });
    return count;
}


// "amg.aggregation" = "compact" on the host (round 6; the oracle's compact_aggregates_graph, the device's compact_* kernels):
// one-hop aggregates around two generations of hashed-priority distance-2 independent sets -- the second generation on the
// subgraph of the leftovers, among those whose leftover neighbours are at least 3/5 of their strong neighbours -- and every
// remaining vertex to the aggregate it has the most strong connections to (ties: the smaller seed); aggregates in the order of
// their seeds' indices.  Integer work: device = host = oracle bit for bit.
template <class Graph>
void mis2_rounds(int64_t n, const Graph &G, std::vector<char> &st)
{
    enum : char { U = 0, S = 1, C = 2, Gn = 3 };
    std::vector<char> c1((size_t)n);
    std::vector<uint64_t> m1((size_t)n), m2((size_t)n);
    int64_t undecided = 0;
    for (int64_t i = 0; i < n; ++i) undecided += st[i] == U;
    while (undecided > 0) {
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                uint64_t m = 0;
                if (st[v] != Gn) {
                    if (st[v] == U) m = agg_key(v);
                    for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                        const int32_t u = G.col(j);
                        if (G.is_strong(v, j) && u != v && st[u] == U) m = std::max(m, agg_key(u));
                    }
                }
                m1[v] = m;
            }
        });
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                uint64_t m = m1[v];
                if (st[v] == U)
                    for (int32_t j = G.begin(v); j < G.end(v); ++j)
                        if (G.is_strong(v, j) && G.col(j) != v) m = std::max(m, m1[G.col(j)]);
                m2[v] = m;
            }
        });
        for (int64_t v = 0; v < n; ++v)
            if (st[v] == U && m2[v] == agg_key(v)) st[v] = S;
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                char c = st[v] == S;
                if (!c && st[v] != Gn)
                    for (int32_t j = G.begin(v); j < G.end(v) && !c; ++j)
                        if (G.is_strong(v, j) && st[G.col(j)] == S) c = 1;
                c1[v] = c;
            }
        });
        int64_t left = 0;
        for (int64_t v = 0; v < n; ++v) {
            if (st[v] != U) continue;
            char c = c1[v];
            for (int32_t j = G.begin(v); j < G.end(v) && !c; ++j)
                if (G.is_strong(v, j) && c1[G.col(j)]) c = 1;
            if (c) st[v] = C;
            else ++left;
        }
        undecided = left;
    }
}

template <class Graph>
int64_t compact_sweep(int64_t n, const Graph &G, std::vector<int32_t> &id)
{
    constexpr int32_t kUndefined = -1, kRemoved = -2;
    enum : char { U = 0, S = 1, C = 2, Gn = 3 };
    std::vector<char> st((size_t)n);
    std::vector<int64_t> owner((size_t)n), nw((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        st[i] = id[i] == kUndefined ? U : Gn;
        owner[i] = id[i] == kUndefined ? -1 : -2;
    }
    auto claim = [&]() { // a vertex of the current graph that is a seed keeps itself, one next to a seed joins it
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t v = b; v < e; ++v) {
                if (st[v] == Gn) continue;
                if (st[v] == S) {
                    owner[v] = v;
                    continue;
                }
                int64_t best = -1;
                for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                    const int32_t u = G.col(j);
                    if (G.is_strong(v, j) && u != v && st[u] == S) best = std::max<int64_t>(best, u);
                }
                if (best >= 0) owner[v] = best;
            }
        });
    };
    mis2_rounds(n, G, st);
    claim();
    parallel_chunks(n, [&](int, int64_t b, int64_t e) { // the graph of the leftovers and its candidates
        for (int64_t v = b; v < e; ++v) {
            if (owner[v] != -1) {
                nw[v] = Gn;
                continue;
            }
            int64_t deg = 0, lo = 0;
            for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                const int32_t u = G.col(j);
                if (!G.is_strong(v, j) || u == v) continue;
                ++deg;
                lo += owner[u] == -1;
            }
            nw[v] = (lo > 0 && 5 * lo >= 3 * deg) ? U : C;
        }
    });
    for (int64_t v = 0; v < n; ++v) st[v] = (char)nw[v];
    mis2_rounds(n, G, st);
    claim();
    for (int pass = 0; pass < 8; ++pass) {
        std::vector<int64_t> left_t(64, 0), moved_t(64, 0);
        parallel_chunks(n, [&](int t, int64_t b, int64_t e) {
            int64_t left = 0, moved = 0;
            for (int64_t v = b; v < e; ++v) {
                nw[v] = owner[v];
                if (owner[v] != -1) continue;
                int64_t best = -1, bc = 0;
                for (int32_t j = G.begin(v); j < G.end(v); ++j) {
                    const int32_t u = G.col(j);
                    if (!G.is_strong(v, j) || u == v || owner[u] < 0) continue;
                    const int64_t o = owner[u];
                    if (o == best) continue;
                    int64_t c = 0;
                    for (int32_t k = G.begin(v); k < G.end(v); ++k)
                        if (G.is_strong(v, k) && G.col(k) != v && owner[G.col(k)] == o) ++c;
                    if (c > bc || (c == bc && o < best)) {
                        bc = c;
                        best = o;
                    }
                }
                if (best >= 0) {
                    nw[v] = best;
                    ++moved;
                } else ++left;
            }
            left_t[(size_t)t % 64] += left;
            moved_t[(size_t)t % 64] += moved;
        });
        owner.swap(nw);
        int64_t left = 0, moved = 0;
        for (size_t t = 0; t < 64; ++t) {
            left += left_t[t];
            moved += moved_t[t];
        }
        if (left == 0 || moved == 0) break;
    }
    std::vector<int32_t> rank((size_t)n, -1);
    int64_t count = 0;
    for (int64_t v = 0; v < n; ++v) {
        if (owner[v] == -1) owner[v] = v; // (unsymmetric patterns only: nothing assigned in reach)
        if (owner[v] == v) rank[v] = (int32_t)count++;
    }
    for (int64_t v = 0; v < n; ++v) id[v] = owner[v] == -2 ? kRemoved : rank[owner[v]];
    return count;
}

} // namespace

double gershgorin_scaled(const HostCsr &A)
{
    // amgcl: `dia` carries over from the previous row when a row has no diagonal entry
    double radius = 0.0, dia = 1.0;
    for (int64_t i = 0; i < A.nrows; ++i) {
        double s = 0.0;
        for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j) {
            s += std::fabs(A.val[j]);
            if (A.col[j] == i) dia = A.val[j];
        }
        s *= std::fabs(1.0 / dia);
        radius = std::max(radius, s);
    }
    return radius;
}

int64_t plain_aggregates(const HostCsr &A, double eps_strong, std::vector<int32_t> &id, std::vector<char> &strong, int mode)
{
    constexpr int32_t kUndefined = -1, kRemoved = -2;
    const int64_t n = A.nrows;
    const double eps2 = eps_strong * eps_strong;
    std::vector<double> dia((size_t)n, 0.0);
    parallel_chunks(n, [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i)
            for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j)
                if (A.col[j] == i) {
                    dia[i] = A.val[j];
                    break;
                }
    });
    strong.assign((size_t)A.nnz(), 0);
    id.assign((size_t)n, kRemoved);
    parallel_chunks(n, [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const double eps_dia_i = eps2 * dia[i];
            bool any = false;
            for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j) {
                const int32_t c = A.col[j];
                const double v = A.val[j];
                const bool s = (c != i) && (eps_dia_i * dia[c] < v * v);
                strong[j] = s;
                any = any || s;
            }
            id[i] = any ? kUndefined : kRemoved; // lonely nodes are removed
        }
    });
    struct FlagGraph {
        const HostCsr &A;
        const std::vector<char> &strong;
        int32_t begin(int64_t i) const { return A.ptr[i]; }
        int32_t end(int64_t i) const { return A.ptr[i + 1]; }
        int32_t col(int32_t j) const { return A.col[j]; }
        bool is_strong(int64_t, int32_t j) const { return strong[j] != 0; }
    };
    return mode == 2 ? compact_sweep(n, FlagGraph{A, strong}, id)
                     : (mode == 1 ? parallel_sweep(n, FlagGraph{A, strong}, id) : greedy_sweep(n, FlagGraph{A, strong}, id));
}

int64_t aggregate_strength_graph(int64_t n, const int32_t *sptr, const int32_t *scol, std::vector<int32_t> &id,
                                 bool id_initialised, int mode)
{
    constexpr int32_t kUndefined = -1, kRemoved = -2;
    if (!id_initialised) {
        id.assign((size_t)n, kRemoved);
        parallel_chunks(n, [&](int, int64_t b, int64_t e) {
            for (int64_t i = b; i < e; ++i) {
                bool any = false;
                for (int32_t j = sptr[i]; j < sptr[i + 1]; ++j) any = any || scol[j] != i;
                id[i] = any ? kUndefined : kRemoved;
            }
        });
    }
    struct CompactGraph {
        const int32_t *sptr, *scol;
        int32_t begin(int64_t i) const { return sptr[i]; }
        int32_t end(int64_t i) const { return sptr[i + 1]; }
        int32_t col(int32_t j) const { return scol[j]; }
        bool is_strong(int64_t i, int32_t j) const { return scol[j] != i; } // the graph also holds the diagonal
    };
    return mode == 2 ? compact_sweep(n, CompactGraph{sptr, scol}, id)
                     : (mode == 1 ? parallel_sweep(n, CompactGraph{sptr, scol}, id) : greedy_sweep(n, CompactGraph{sptr, scol}, id));
}

HostCsr smoothed_prolongation(const HostCsr &A, const std::vector<char> &strong, const std::vector<int32_t> &id,
                              int64_t nagg, double omega)
{
    const int64_t n = A.nrows;
    HostCsr P;
    P.nrows = n;
    P.ncols = nagg;
    P.ptr.assign((size_t)n + 1, 0);
    // rows touch a handful of aggregates: collect (aggregate, value) pairs, sort, merge
    auto row_entries = [&](int64_t i, std::vector<std::pair<int32_t, double>> &ent) {
        ent.clear();
        double dia = 0.0; // filtered diagonal: diagonal plus the weak connections
        for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j)
            if (A.col[j] == i || !strong[j]) dia += A.val[j];
        dia = -omega * (1.0 / dia);
        for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j) {
            const int32_t ca = A.col[j];
            if (ca != i && !strong[j]) continue;
            const int32_t cp = id[ca];
            if (cp < 0) continue; // P_tent row of a removed node is empty
            const double va = (ca == i) ? (1.0 - omega) : dia * A.val[j];
            ent.emplace_back(cp, va);
        }
        std::stable_sort(ent.begin(), ent.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t r = 0; r < ent.size(); ++r) {
            if (w > 0 && ent[w - 1].first == ent[r].first) ent[w - 1].second += ent[r].second;
            else ent[w++] = ent[r];
        }
        ent.resize(w);
    };
    parallel_chunks(n, [&](int, int64_t b, int64_t e) {
        std::vector<std::pair<int32_t, double>> ent;
        for (int64_t i = b; i < e; ++i) {
            row_entries(i, ent);
            P.ptr[i + 1] = (int32_t)ent.size();
        }
    });
    exclusive_scan_rows(P.ptr);
    P.col.resize((size_t)P.nnz());
    P.val.resize((size_t)P.nnz());
    parallel_chunks(n, [&](int, int64_t b, int64_t e) {
        std::vector<std::pair<int32_t, double>> ent;
        for (int64_t i = b; i < e; ++i) {
            row_entries(i, ent);
            int32_t p = P.ptr[i];
            for (const auto &kv : ent) {
                P.col[p] = kv.first;
                P.val[p++] = kv.second;
            }
        }
    });
    return P;
}

HostCsr transpose(const HostCsr &A, std::vector<int32_t> *entry_map)
{
    HostCsr T;
    T.nrows = A.ncols;
    T.ncols = A.nrows;
    T.ptr.assign((size_t)T.nrows + 1, 0);
    const int64_t nnz = A.nnz();
    for (int64_t j = 0; j < nnz; ++j) ++T.ptr[A.col[j] + 1];
    exclusive_scan_rows(T.ptr);
    T.col.resize((size_t)nnz);
    T.val.resize((size_t)nnz);
    if (entry_map) entry_map->resize((size_t)nnz);
    std::vector<int32_t> head(T.ptr.begin(), T.ptr.end() - 1);
    for (int64_t i = 0; i < A.nrows; ++i)
        for (int32_t j = A.ptr[i]; j < A.ptr[i + 1]; ++j) {
            const int32_t h = head[A.col[j]]++;
            T.col[h] = (int32_t)i; // rows visited in order => sorted columns
            T.val[h] = A.val[j];
            if (entry_map) (*entry_map)[h] = j;
        }
    return T;
}

HostCsr multiply(const HostCsr &A, const HostCsr &B)
{
    PS_REQUIRE(A.ncols == B.nrows, PSOLVE_HIP_EINVAL, "multiply: shape mismatch");
    HostCsr C;
    C.nrows = A.nrows;
    C.ncols = B.ncols;
    C.ptr.assign((size_t)C.nrows + 1, 0);
    const int64_t m = B.ncols;
    // pass 1: row sizes
    parallel_chunks(A.nrows, [&](int, int64_t b, int64_t e) {
        std::vector<int64_t> marker((size_t)m, -1);
        for (int64_t i = b; i < e; ++i) {
            int32_t cnt = 0;
            for (int32_t ja = A.ptr[i]; ja < A.ptr[i + 1]; ++ja) {
                const int32_t ca = A.col[ja];
                for (int32_t jb = B.ptr[ca]; jb < B.ptr[ca + 1]; ++jb) {
                    const int32_t cb = B.col[jb];
                    if (marker[cb] != i) {
                        marker[cb] = i;
                        ++cnt;
                    }
                }
            }
            C.ptr[i + 1] = cnt;
        }
    });
    exclusive_scan_rows(C.ptr);
    C.col.resize((size_t)C.nnz());
    C.val.resize((size_t)C.nnz());
    // pass 2: values; columns sorted per row
    parallel_chunks(A.nrows, [&](int, int64_t b, int64_t e) {
        std::vector<int32_t> marker((size_t)m, -1); // position of column in the current row
        std::vector<std::pair<int32_t, double>> ent;
        for (int64_t i = b; i < e; ++i) {
            const int32_t row_beg = C.ptr[i];
            int32_t row_end = row_beg;
            for (int32_t ja = A.ptr[i]; ja < A.ptr[i + 1]; ++ja) {
                const int32_t ca = A.col[ja];
                const double va = A.val[ja];
                for (int32_t jb = B.ptr[ca]; jb < B.ptr[ca + 1]; ++jb) {
                    const int32_t cb = B.col[jb];
                    const double v = va * B.val[jb];
                    if (marker[cb] < 0) {
                        marker[cb] = row_end;
                        C.col[row_end] = cb;
                        C.val[row_end] = v;
                        ++row_end;
                    } else {
                        C.val[marker[cb]] += v;
                    }
                }
            }
            // sort the row by column (rows are short)
            const int32_t len = row_end - row_beg;
            ent.resize((size_t)len);
            for (int32_t k = 0; k < len; ++k) ent[k] = {C.col[row_beg + k], C.val[row_beg + k]};
            std::sort(ent.begin(), ent.end(), [](const auto &a, const auto &b2) { return a.first < b2.first; });
            for (int32_t k = 0; k < len; ++k) {
                C.col[row_beg + k] = ent[k].first;
                C.val[row_beg + k] = ent[k].second;
                marker[ent[k].first] = -1;
            }
        }
    });
    return C;
}


// ---------------------------------------------------------------------------------------------
// block value types (AMGCL_Block<N>): see the long comment in oracle/amg_oracle.c
// ---------------------------------------------------------------------------------------------
HostBcsr to_blocks(const HostCsr &A, int b)
{
    PS_REQUIRE(b >= 2 && b <= 4 && A.nrows % b == 0 && A.ncols % b == 0, PSOLVE_HIP_EINVAL,
               "block_size does not divide the matrix size");
    HostBcsr B;
    B.nb = A.nrows / b;
    B.b = b;
    const int64_t ncb = A.ncols / b;
    const int bb = b * b;
    B.ptr.assign((size_t)B.nb + 1, 0);
    parallel_chunks(B.nb, [&](int, int64_t lo, int64_t hi) {
        std::vector<int32_t> cols;
        for (int64_t ib = lo; ib < hi; ++ib) {
            cols.clear();
            for (int r = 0; r < b; ++r)
                for (int32_t j = A.ptr[ib * b + r]; j < A.ptr[ib * b + r + 1]; ++j) cols.push_back(A.col[j] / b);
            std::sort(cols.begin(), cols.end());
            B.ptr[ib + 1] = (int32_t)(std::unique(cols.begin(), cols.end()) - cols.begin());
        }
    });
    exclusive_scan_rows(B.ptr);
    const int64_t nnzb = B.ptr[B.nb];
    B.col.resize((size_t)nnzb);
    B.val.assign((size_t)nnzb * bb, 0.0);
    (void)ncb;
    parallel_chunks(B.nb, [&](int, int64_t lo, int64_t hi) {
        std::vector<int32_t> cols;
        for (int64_t ib = lo; ib < hi; ++ib) {
            cols.clear();
            for (int r = 0; r < b; ++r)
                for (int32_t j = A.ptr[ib * b + r]; j < A.ptr[ib * b + r + 1]; ++j) cols.push_back(A.col[j] / b);
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            const int32_t beg = B.ptr[ib];
            for (size_t k = 0; k < cols.size(); ++k) B.col[beg + k] = cols[k];
            for (int r = 0; r < b; ++r)
                for (int32_t j = A.ptr[ib * b + r]; j < A.ptr[ib * b + r + 1]; ++j) {
                    const int32_t cb = A.col[j] / b, cc = A.col[j] % b;
                    const size_t slot = (size_t)(std::lower_bound(cols.begin(), cols.end(), cb) - cols.begin());
                    B.val[((size_t)beg + slot) * bb + r * b + cc] += A.val[j];
                }
        }
    });
    return B;
}

void invert_block(int b, const double *X, double *Y)
{
    double a[16], inv[16];
    for (int i = 0; i < b * b; ++i) {
        a[i] = X[i];
        inv[i] = 0.0;
    }
    for (int i = 0; i < b; ++i) inv[i * b + i] = 1.0;
    for (int c = 0; c < b; ++c) {
        int piv = c;
        for (int r = c + 1; r < b; ++r)
            if (std::fabs(a[r * b + c]) > std::fabs(a[piv * b + c])) piv = r;
        if (piv != c)
            for (int k = 0; k < b; ++k) {
                std::swap(a[c * b + k], a[piv * b + k]);
                std::swap(inv[c * b + k], inv[piv * b + k]);
            }
        const double d = 1.0 / a[c * b + c];
        for (int k = 0; k < b; ++k) {
            a[c * b + k] *= d;
            inv[c * b + k] *= d;
        }
        for (int r = 0; r < b; ++r) {
            if (r == c) continue;
            const double f = a[r * b + c];
            if (f == 0.0) continue;
            for (int k = 0; k < b; ++k) {
                a[r * b + k] -= f * a[c * b + k];
                inv[r * b + k] -= f * inv[c * b + k];
            }
        }
    }
    for (int i = 0; i < b * b; ++i) Y[i] = inv[i];
}

namespace {

void blk_mul(int b, const double *X, const double *Y, double *Z)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < b; ++j) {
            double s = 0.0;
            for (int k = 0; k < b; ++k) s += X[i * b + k] * Y[k * b + j];
            Z[i * b + j] = s;
        }
}
double blk_trace(int b, const double *X)
{
    double t = 0.0;
    for (int i = 0; i < b; ++i) t += X[i * b + i];
    return t;
}
double blk_fro(int b, const double *X)
{
    double s = 0.0;
    for (int i = 0; i < b * b; ++i) s += X[i] * X[i];
    return std::sqrt(s);
}
const double *blk_diag(const HostBcsr &B, int64_t ib)
{
    const int32_t *beg = B.col.data() + B.ptr[ib], *end = B.col.data() + B.ptr[ib + 1];
    const int32_t *it = std::lower_bound(beg, end, (int32_t)ib);
    if (it == end || *it != ib) return nullptr;
    return B.val.data() + (size_t)(it - B.col.data()) * B.b * B.b;
}

// strength of connection on the block graph + the (sequential) greedy sweep on it
int64_t block_aggregates(const HostBcsr &B, double eps_strong, std::vector<int32_t> &id, std::vector<char> &strong, int mode)
{
    const int b = B.b, bb = b * b;
    const double eps2 = eps_strong * eps_strong;
    strong.assign((size_t)B.ptr[B.nb], 0);
    const double zero[16] = {0};
    parallel_chunks(B.nb, [&](int, int64_t lo, int64_t hi) {
        double t1[16], t2[16];
        for (int64_t i = lo; i < hi; ++i) {
            const double *di = blk_diag(B, i);
            for (int32_t j = B.ptr[i]; j < B.ptr[i + 1]; ++j) {
                const int32_t c = B.col[j];
                const double *v = B.val.data() + (size_t)j * bb;
                const double *dc = blk_diag(B, c);
                blk_mul(b, v, v, t1);
                blk_mul(b, di ? di : zero, dc ? dc : zero, t2);
                strong[j] = (c != i) && (eps2 * blk_trace(b, t2) < blk_trace(b, t1));
            }
        }
    });
    // reuse the scalar sweep: a pattern-only CSR whose "values" make exactly the strong entries strong
    HostCsr G;
    G.nrows = G.ncols = B.nb;
    G.ptr = B.ptr;
    G.col = B.col;
    G.val.resize(B.col.size());
    for (size_t j = 0; j < G.val.size(); ++j) G.val[j] = strong[j] ? 1.0 : 0.0;
    for (int64_t i = 0; i < B.nb; ++i)
        for (int32_t j = G.ptr[i]; j < G.ptr[i + 1]; ++j)
            if (G.col[j] == i) G.val[j] = 1.0;
    std::vector<char> s2;
    // eps = 0 on G: strong <=> off-diagonal and value^2 > 0
    return plain_aggregates(G, 0.0, id, s2, mode);
}

double block_gershgorin(const HostBcsr &B)
{
    const int b = B.b, bb = b * b;
    double radius = 0.0, dia[16], inv[16];
    for (int i = 0; i < bb; ++i) dia[i] = (i % (b + 1) == 0) ? 1.0 : 0.0;
    for (int64_t i = 0; i < B.nb; ++i) {
        double s = 0.0;
        for (int32_t j = B.ptr[i]; j < B.ptr[i + 1]; ++j) {
            s += blk_fro(b, B.val.data() + (size_t)j * bb);
            if (B.col[j] == i) std::copy_n(B.val.data() + (size_t)j * bb, bb, dia);
        }
        invert_block(b, dia, inv);
        s *= blk_fro(b, inv);
        radius = std::max(radius, s);
    }
    return radius;
}

HostCsr block_smoothed_prolongation(const HostBcsr &B, const std::vector<char> &strong, const std::vector<int32_t> &id,
                                    int64_t nagg, double omega)
{
    const int b = B.b, bb = b * b;
    const int64_t nb = B.nb;
    struct Ent {
        int32_t agg;
        double v[16];
    };
    auto row_entries = [&](int64_t i, std::vector<Ent> &ent) {
        ent.clear();
        double dia[16], dinv[16];
        for (int k = 0; k < bb; ++k) dia[k] = 0.0;
        for (int32_t j = B.ptr[i]; j < B.ptr[i + 1]; ++j)
            if (B.col[j] == i || !strong[j])
                for (int k = 0; k < bb; ++k) dia[k] += B.val[(size_t)j * bb + k];
        invert_block(b, dia, dinv);
        for (int k = 0; k < bb; ++k) dinv[k] *= -omega;
        for (int32_t j = B.ptr[i]; j < B.ptr[i + 1]; ++j) {
            const int32_t ca = B.col[j];
            if (ca != i && !strong[j]) continue;
            const int32_t cp = id[ca];
            if (cp < 0) continue;
            Ent e;
            e.agg = cp;
            if (ca == i) {
                for (int k = 0; k < bb; ++k) e.v[k] = (k % (b + 1) == 0) ? (1.0 - omega) : 0.0;
            } else {
                blk_mul(b, dinv, B.val.data() + (size_t)j * bb, e.v);
            }
            ent.push_back(e);
        }
        std::stable_sort(ent.begin(), ent.end(), [](const Ent &x, const Ent &y) { return x.agg < y.agg; });
        size_t w = 0;
        for (size_t r = 0; r < ent.size(); ++r) {
            if (w > 0 && ent[w - 1].agg == ent[r].agg) {
                for (int k = 0; k < bb; ++k) ent[w - 1].v[k] += ent[r].v[k];
            } else {
                ent[w++] = ent[r];
            }
        }
        ent.resize(w);
    };
    std::vector<int32_t> bptr((size_t)nb + 1, 0);
    parallel_chunks(nb, [&](int, int64_t lo, int64_t hi) {
        std::vector<Ent> ent;
        for (int64_t i = lo; i < hi; ++i) {
            row_entries(i, ent);
            bptr[i + 1] = (int32_t)ent.size();
        }
    });
    exclusive_scan_rows(bptr);
    HostCsr P;
    P.nrows = nb * b;
    P.ncols = nagg * b;
    P.ptr.assign((size_t)P.nrows + 1, 0);
    for (int64_t i = 0; i < nb; ++i)
        for (int r = 0; r < b; ++r) P.ptr[i * b + r + 1] = (bptr[i + 1] - bptr[i]) * b;
    exclusive_scan_rows(P.ptr);
    P.col.resize((size_t)P.nnz());
    P.val.resize((size_t)P.nnz());
    parallel_chunks(nb, [&](int, int64_t lo, int64_t hi) {
        std::vector<Ent> ent;
        for (int64_t i = lo; i < hi; ++i) {
            row_entries(i, ent);
            for (int r = 0; r < b; ++r) {
                int32_t p = P.ptr[i * b + r];
                for (const Ent &e : ent)
                    for (int c = 0; c < b; ++c) {
                        P.col[p] = e.agg * b + c;
                        P.val[p++] = e.v[r * b + c]; // full blocks, explicit zeros kept
                    }
            }
        }
    });
    return P;
}

} // namespace

// amgcl/amg.hpp do_init(): coarsen while rows > coarse_enough and levels < max_levels; the coarsest
// level is relaxed, not factorised (direct_coarse = false in AMGCL.cpp:46).
// amgcl/coarsening/tentative_prolongation.hpp without near-nullspace vectors: P(i, id[i]) = 1 (block value types: the
// identity block); rows of removed nodes are empty.  "amg.coarsening" = "aggregation" (amgcl/coarsening/aggregation.hpp).
HostCsr tentative_prolongation(int64_t n_nodes, const std::vector<int32_t> &id, int64_t nagg, int bs)
{
    HostCsr P;
    P.nrows = n_nodes * bs;
    P.ncols = nagg * bs;
    P.ptr.assign((size_t)P.nrows + 1, 0);
    for (int64_t i = 0; i < n_nodes; ++i)
        for (int r = 0; r < bs; ++r) P.ptr[(size_t)(i * bs + r) + 1] = id[(size_t)i] >= 0 ? bs : 0;
    for (size_t i = 1; i < P.ptr.size(); ++i) P.ptr[i] += P.ptr[i - 1];
    P.col.resize((size_t)P.ptr.back());
    P.val.resize((size_t)P.ptr.back());
    for (int64_t i = 0; i < n_nodes; ++i) {
        if (id[(size_t)i] < 0) continue;
        for (int r = 0; r < bs; ++r) {
            const int32_t b = P.ptr[(size_t)(i * bs + r)];
            for (int c = 0; c < bs; ++c) {
                P.col[(size_t)b + c] = id[(size_t)i] * bs + c;
                P.val[(size_t)b + c] = r == c ? 1.0 : 0.0;
            }
        }
    }
    return P;
}

// the factor amgcl's aggregation coarsening scales the Galerkin operator by: 1 / over_interp, computed in single precision
// (detail::scaled_galerkin takes a float)
double over_interp_scale(double over_interp, int bs)
{
    const float oi = over_interp > 0 ? (float)over_interp : (bs == 1 ? 1.5f : 2.0f);
    const float sf = 1 / oi;
    return (double)sf;
}

std::vector<HostLevel> build_hierarchy(HostCsr &&fine, const AmgParams &prm)
{
    const bool timing = std::getenv("PSOLVE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char *what, int64_t rows) {
        if (!timing) return;
        const double t1 = now();
        std::fprintf(stderr, "[psolve timing] amg host %-22s rows=%lld %.3f s\n", what, (long long)rows, t1 - t0);
        t0 = t1;
    };
    std::vector<HostLevel> levels;
    HostCsr A = std::move(fine);
    double eps = prm.eps_strong;
    bool have_A = true;
    while (A.nrows > prm.coarse_enough) {
        levels.emplace_back();
        HostLevel &L = levels.back();
        L.A = std::move(A);
        if ((int)levels.size() >= prm.max_levels) {
            have_A = false;
            break;
        }
        std::vector<int32_t> id;
        std::vector<char> strong;
        // SA's own power_iters defaults to 0 in AMGCL (polysolve does not set it): Gershgorin
        PS_REQUIRE(prm.sa_power_iters == 0, PSOLVE_HIP_EINVAL,
                   "amg.sa_power_iters > 0 is not supported (AMGCL's default 0 = Gershgorin is)");
        double omega = prm.sa_relax;
        int64_t nagg = 0;
        if (prm.block_size > 1) {
            const HostBcsr B = to_blocks(L.A, prm.block_size);
            nagg = block_aggregates(B, eps, id, strong, prm.aggregation);
            eps *= 0.5;
            if (nagg == 0) {
                have_A = false;
                break;
            }
            if (prm.coarsening == 1) {
                L.P = tentative_prolongation(B.nb, id, nagg, prm.block_size);
            } else {
                omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / block_gershgorin(B) : 2.0 / 3.0;
                L.P = block_smoothed_prolongation(B, strong, id, nagg, omega);
            }
        } else {
            nagg = plain_aggregates(L.A, eps, id, strong, prm.aggregation);
            lap("aggregates", L.A.nrows);
            eps *= 0.5;
            if (nagg == 0) { // amgcl error::empty_level: the level is (block-)diagonal
                have_A = false;
                break;
            }
            if (prm.coarsening == 1) {
                L.P = tentative_prolongation(L.A.nrows, id, nagg, 1);
            } else {
                omega *= prm.estimate_spectral_radius ? (4.0 / 3.0) / gershgorin_scaled(L.A) : 2.0 / 3.0;
                lap("gershgorin", L.A.nrows);
                L.P = smoothed_prolongation(L.A, strong, id, nagg, omega);
            }
            lap("prolongation", L.A.nrows);
        }
        L.omega = omega;
        L.naggregates = nagg;
        L.R = transpose(L.P, &L.r_from_p);
        lap("transpose", L.A.nrows);
        HostCsr AP = multiply(L.A, L.P);
        lap("A*P", L.A.nrows);
        A = multiply(L.R, AP);
        if (prm.coarsening == 1) {
            const double sc = over_interp_scale(prm.over_interp, prm.block_size);
            for (double &v : A.val) v = sc * v;
        }
        lap("R*(AP)", L.A.nrows);
        if (prm.block_size <= 1) { // symbolic data for the device-side numeric refresh (scalar path)
            L.id = std::move(id);
            L.AP = std::move(AP);
        }
    }
    if (have_A) {
        levels.emplace_back();
        levels.back().A = std::move(A);
    }
    return levels;
}

} // namespace psolve
