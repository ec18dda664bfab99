/*
 * amg_oracle.c -- CPU restatement of the AMGCL preconditioner the reference configures in
 * src/polysolve/linear/AMGCL.cpp:32-65 (smoothed-aggregation AMG + Chebyshev relaxation).
 * TEST INFRASTRUCTURE ONLY (see psolve_oracle.c header).  PARITY UNPINNED: AMGCL 1.4.3 is not
 * vendored in /root/reference (cmake/recipes/amgcl.cmake:47) and not present in this image; the
 * functions below restate its published algorithms, each citing the upstream header it follows:
 *     amgcl/amg.hpp                              -- hierarchy construction, apply(), cycle()
 *     amgcl/coarsening/plain_aggregates.hpp      -- strength of connection + greedy aggregation
 *     amgcl/coarsening/tentative_prolongation.hpp
 *     amgcl/coarsening/smoothed_aggregation.hpp  -- P = (I - w D_f^-1 A_f) P_tent, R = P^T
 *     amgcl/coarsening/detail/galerkin.hpp       -- A_c = R A P
 *     amgcl/relaxation/chebyshev.hpp             -- Chebyshev polynomial smoother
 *     amgcl/backend/builtin.hpp                  -- spectral_radius<scale>() (power iteration /
 *                                                   Gershgorin), single-threaded variant (tid 0)
 * Scalar value type only (block_size 1, AMGCL.cpp:148-184).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef int32_t idx_t;

typedef struct {
    int64_t nrows, ncols;
    idx_t *ptr, *col;
    double *val;
} csr_t;

static csr_t *csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz)
{
    csr_t *A = (csr_t *)calloc(1, sizeof(csr_t));
    A->nrows = nrows;
    A->ncols = ncols;
    A->ptr = (idx_t *)calloc((size_t)nrows + 1, sizeof(idx_t));
    A->col = (idx_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(idx_t));
    A->val = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    return A;
}

static void csr_free(csr_t *A)
{
    if (!A) return;
    free(A->ptr); free(A->col); free(A->val); free(A);
}

static void csr_spmv(double alpha, const csr_t *A, const double *x, double beta, double *y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < A->nrows; ++i) {
        double s = 0.0;
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) s += A->val[j] * x[A->col[j]];
        y[i] = (beta != 0.0) ? alpha * s + beta * y[i] : alpha * s;
    }
}

static void csr_residual(const double *f, const csr_t *A, const double *x, double *r)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < A->nrows; ++i) {
        double s = f[i];
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) s -= A->val[j] * x[A->col[j]];
        r[i] = s;
    }
}

/* ---- std::mt19937 + libstdc++ uniform_real_distribution<double>(-1,1), as used by
 *      amgcl::backend::spectral_radius for the power-iteration start vector (seed = thread id). */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t s)
{
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* std::generate_canonical<double,53>(mt19937): two 32-bit draws, (lo + hi*2^32) / 2^64 */
static double mt_uniform_pm1(mt19937_t *g)
{
    double lo = (double)mt_next(g);
    double hi = (double)mt_next(g);
    double c = (lo + hi * 4294967296.0) / 18446744073709551616.0;
    if (c >= 1.0) c = nextafter(1.0, 0.0);
    return 2.0 * c - 1.0; /* (b - a) * c + a with a=-1, b=1 */
}

/* amgcl/backend/builtin.hpp: spectral_radius<scale>(A, power_iters).  scale => rho(D^-1 A). */
static double spectral_radius(const csr_t *A, int scale, int power_iters)
{
    const int64_t n = A->nrows;
    double radius;
    if (power_iters <= 0) {
        /* Gershgorin disc bound */
        radius = 0.0;
        double dia = 1.0;
        for (int64_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
                s += fabs(A->val[j]);
                if (scale && A->col[j] == i) dia = A->val[j];
            }
            if (scale) s *= fabs(1.0 / dia);
            if (s > radius) radius = s;
        }
    } else {
        double *b0 = (double *)malloc((size_t)n * 8), *b1 = (double *)malloc((size_t)n * 8);
        mt19937_t rng;
        mt_seed(&rng, 0u);
        double b0_norm = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double v = mt_uniform_pm1(&rng);
            b0[i] = v;
            b0_norm += v * v;
        }
        b0_norm = 1.0 / sqrt(b0_norm);
        for (int64_t i = 0; i < n; ++i) b0[i] = b0_norm * b0[i];
        radius = 0.0;
        /* the diagonal a row scales with (a row without one inherits the previous row's, as the sequential loop did) */
        double *rdia = (double *)malloc((size_t)n * 8);
        {
            double dia = 1.0;
            for (int64_t i = 0; i < n; ++i) {
                for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j)
                    if (scale && A->col[j] == i) dia = A->val[j];
                rdia[i] = dia;
            }
        }
        for (int iter = 0; iter < power_iters;) {
            double b1_norm = 0.0;
            radius = 0.0;
            /* rows in parallel (AMGCL's loop is an OpenMP one) ... */
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) {
                double s = 0.0;
                for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) s += A->val[j] * b0[A->col[j]];
                if (scale) s = (1.0 / rdia[i]) * s;
                b1[i] = s;
            }
            /* ... the two sums in row order, so that the estimate does not depend on the thread count */
            for (int64_t i = 0; i < n; ++i) {
                b1_norm += b1[i] * b1[i];
                radius += fabs(b1[i] * b0[i]);
            }
            if (++iter < power_iters) {
                b1_norm = 1.0 / sqrt(b1_norm);
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; ++i) b0[i] = b1_norm * b1[i];
            }
        }
        free(b0); free(b1); free(rdia);
    }
    return radius < 0 ? 2.0 : radius;
}

/* ---- amgcl/coarsening/plain_aggregates.hpp ------------------------------------------------ */
#define AGG_UNDEFINED (-1)
#define AGG_REMOVED (-2)

static int64_t parallel_aggregates_graph(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, idx_t *id,
                                         int *rounds_out);
static int64_t compact_aggregates_graph(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, idx_t *id,
                                        int *rounds_out);

/* returns aggregate count; fills strong[nnz] and id[n] (id < 0 => removed).  mode 0: amgcl's sweep; 1: "parallel" */
static int64_t plain_aggregates_mode(const csr_t *A, double eps_strong, char *strong, idx_t *id, int mode);
static int64_t plain_aggregates(const csr_t *A, double eps_strong, char *strong, idx_t *id)
{
    return plain_aggregates_mode(A, eps_strong, strong, id, 0);
}
static int64_t plain_aggregates_mode(const csr_t *A, double eps_strong, char *strong, idx_t *id, int mode)
{
    const int64_t n = A->nrows;
    const double eps2 = eps_strong * eps_strong;
    double *dia = (double *)malloc((size_t)n * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double d = 0.0;
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j)
            if (A->col[j] == i) { d = A->val[j]; break; }
        dia[i] = d;
    }
#pragma omp parallel for schedule(static) /* (amgcl: the strength test is an OpenMP loop, the sweep below is not) */
    for (int64_t i = 0; i < n; ++i) {
        double eps_dia_i = eps2 * dia[i];
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
            idx_t c = A->col[j];
            double v = A->val[j];
            strong[j] = (c != i) && (eps_dia_i * dia[c] < v * v);
        }
    }
    free(dia);
    if (mode == 1) return parallel_aggregates_graph(n, A->ptr, A->col, strong, id, NULL);
    if (mode == 2) return compact_aggregates_graph(n, A->ptr, A->col, strong, id, NULL);

    int64_t max_neib = 0;
    for (int64_t i = 0; i < n; ++i) {
        idx_t j = A->ptr[i], e = A->ptr[i + 1];
        if (e - j > max_neib) max_neib = e - j;
        idx_t state = AGG_REMOVED;
        for (; j < e; ++j)
            if (strong[j]) { state = AGG_UNDEFINED; break; }
        id[i] = state;
    }
    idx_t *neib = (idx_t *)malloc((size_t)(max_neib + 1) * sizeof(idx_t));
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (id[i] != AGG_UNDEFINED) continue;
        idx_t cur = (idx_t)count++;
        id[i] = cur;
        int64_t nn = 0;
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
            idx_t c = A->col[j];
            if (strong[j] && id[c] != AGG_REMOVED) {
                id[c] = cur;
                neib[nn++] = c;
            }
        }
        for (int64_t q = 0; q < nn; ++q) {
            idx_t c = neib[q];
            for (idx_t j = A->ptr[c]; j < A->ptr[c + 1]; ++j) {
                idx_t cc = A->col[j];
                if (strong[j] && id[cc] == AGG_UNDEFINED) id[cc] = cur;
            }
        }
    }
    free(neib);
    if (count == 0) return 0;
    /* aggregates may vanish when a later seed steals all members: renumber */
    idx_t *cnt = (idx_t *)calloc((size_t)count, sizeof(idx_t));
    for (int64_t i = 0; i < n; ++i)
        if (id[i] >= 0) cnt[id[i]] = 1;
    for (int64_t k = 1; k < count; ++k) cnt[k] += cnt[k - 1];
    if (count > cnt[count - 1]) {
        int64_t newcount = cnt[count - 1];
        for (int64_t i = 0; i < n; ++i)
            if (id[i] >= 0) id[i] = cnt[id[i]] - 1;
        count = newcount;
    }
    free(cnt);
    return count;
}

/* ---- "amg.aggregation" = "parallel" (round 5): THIS REPOSITORY'S opt-in alternative to the sequential sweep ---------
 * Not AMGCL: amgcl::coarsening::plain_aggregates is the loop above, and stays the default (AMGCL.cpp:32-65).  The sweep's
 * seeds are the lexicographically first distance-2 maximal independent set of the strength graph -- a chain of decisions
 * as deep as the mesh is long; here the seeds are the distance-2 maximal independent set by HASHED priorities (Luby-style
 * synchronous rounds: an undecided vertex whose key is the largest among the undecided vertices within two hops becomes a
 * seed, everything within two hops of a seed is covered), which takes a dozen rounds whatever the mesh.  The membership
 * rule is the sweep's closed form (see amg_aggregate.hip): a vertex next to a seed belongs to the largest such seed,
 * otherwise to the smallest seed two hops away; aggregates are numbered in seed order.  Symmetric strength patterns only
 * (the SPD case); integer work, so device and oracle agree bit for bit. */
static uint32_t agg_hash32(uint32_t v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
static uint64_t agg_key(int64_t v) { return ((uint64_t)agg_hash32((uint32_t)v) << 32) | (uint32_t)v; }

static int64_t parallel_aggregates_graph(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, idx_t *id,
                                         int *rounds_out)
{
    enum { ST_U = 0, ST_S = 1, ST_C = 2, ST_G = 3 };
    char *st = (char *)malloc((size_t)n + 1);
    uint64_t *m1 = (uint64_t *)malloc((size_t)n * 8 + 8), *m2 = (uint64_t *)malloc((size_t)n * 8 + 8);
    char *c1 = (char *)malloc((size_t)n + 1);
    int64_t undecided = 0;
    for (int64_t i = 0; i < n; ++i) {
        int any = 0;
        for (idx_t j = ptr[i]; j < ptr[i + 1]; ++j)
            if (strong[j]) { any = 1; break; }
        st[i] = any ? ST_U : ST_G;
        undecided += any;
    }
    int rounds = 0;
    uint64_t *key = (uint64_t *)malloc((size_t)n * 8 + 8);
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) key[v] = agg_key(v);
    while (undecided > 0) {
        ++rounds;
        /* largest undecided key in the closed neighbourhood, then once more: within two hops */
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            uint64_t m = st[v] == ST_U ? key[v] : 0;
            for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
                idx_t u = col[j];
                if (strong[j] && st[u] == ST_U) { uint64_t k = key[u]; if (k > m) m = k; }
            }
            m1[v] = m;
        }
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            uint64_t m = m1[v];
            for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j)
                if (strong[j] && m1[col[j]] > m) m = m1[col[j]];
            m2[v] = m;
        }
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v)
            if (st[v] == ST_U && m2[v] == key[v]) st[v] = ST_S;
        /* everything within two hops of a seed is covered */
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            char c = st[v] == ST_S;
            for (idx_t j = ptr[v]; j < ptr[v + 1] && !c; ++j)
                if (strong[j] && st[col[j]] == ST_S) c = 1;
            c1[v] = c;
        }
        int64_t left = 0;
#pragma omp parallel for schedule(static) reduction(+ : left)
        for (int64_t v = 0; v < n; ++v) {
            if (st[v] != ST_U) continue;
            char c = c1[v];
            for (idx_t j = ptr[v]; j < ptr[v + 1] && !c; ++j)
                if (strong[j] && c1[col[j]]) c = 1;
            if (c) st[v] = ST_C; else ++left;
        }
        undecided = left;
    }
    if (rounds_out) *rounds_out = rounds;
    /* aggregates in seed order; membership by the sweep's closed form */
    idx_t *rank = (idx_t *)malloc((size_t)n * sizeof(idx_t) + 8);
    int64_t count = 0;
    for (int64_t v = 0; v < n; ++v) rank[v] = st[v] == ST_S ? (idx_t)count++ : -1;
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) {
        if (st[v] == ST_G) { id[v] = AGG_REMOVED; continue; }
        int64_t best = -1;
        for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
            idx_t c = col[j];
            if (strong[j] && c != v && st[c] == ST_S && c > best) best = c;
        }
        if (best < 0 && st[v] == ST_S) best = v;
        if (best < 0) {
            int64_t first = INT64_MAX;
            for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
                idx_t c = col[j];
                if (!strong[j] || c == v) continue;
                for (idx_t k = ptr[c]; k < ptr[c + 1]; ++k) {
                    idx_t s2 = col[k];
                    if (strong[k] && s2 != c && st[s2] == ST_S && s2 < first) first = s2;
                }
            }
            best = first;
        }
        id[v] = best == INT64_MAX ? AGG_UNDEFINED : rank[best];
    }
    free(st); free(m1); free(m2); free(c1); free(rank); free(key);
    return count;
}


/* ---- "amg.aggregation" = "compact" (round 6): THIS REPOSITORY'S second opt-in alternative to the sequential sweep --------
 * "parallel" above takes everything within TWO hops of a seed of a random-priority independent set: a random packing, 46
 * nodes per aggregate on a 27-point node graph where the sweep's lattice packing gives 26 -- Q1 elasticity, 3 M DOF: 37 -> 62
 * iterations.  "compact" keeps aggregates at ONE hop and fills the gaps with a second generation of seeds:
 *   A  seeds = the distance-2 maximal independent set by hashed priorities (the rounds of "parallel");
 *   B  a vertex next to a seed joins it (seeds are three hops apart: at most one; the larger index if a pattern is unsymmetric);
 *   C  a leftover vertex (two hops from every seed) is a CANDIDATE if it has leftover neighbours and they are at least 3/5
 *      of its strong neighbours;
 *   D  second-generation seeds = the distance-2 maximal independent set, by the same priorities, of the subgraph induced by
 *      the leftovers (paths through assigned vertices do not count), taken among the candidates only;
 *   E  a leftover next to a second-generation seed joins it;
 *   F  every remaining vertex joins the aggregate it has the most strong connections to (ties: the smaller seed) -- one
 *      synchronous pass on a symmetric pattern (a leftover always touches a first-generation aggregate), repeated on an
 *      unsymmetric one until nothing changes, whatever is left then seeds its own aggregate;
 *   aggregates are numbered in the order of their seeds' indices.
 * Integer work on the stored strength rows: device (amg_aggregate.hip), host (amg_setup.cpp) and this file agree bit for bit.
 * Not AMGCL: amgcl's plain_aggregates is the sweep and stays the default (AMGCL.cpp:32-65). */
enum { CST_U = 0, CST_S = 1, CST_C = 2, CST_G = 3 };

/* the rounds of parallel_aggregates_graph on a state array: CST_U vertices compete, CST_C vertices only pass keys on and get
 * covered, CST_G vertices are not part of the graph.  Returns the rounds taken. */
static int mis2_rounds(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, char *st, const uint64_t *key)
{
    uint64_t *m1 = (uint64_t *)malloc((size_t)n * 8 + 8), *m2 = (uint64_t *)malloc((size_t)n * 8 + 8);
    char *c1 = (char *)malloc((size_t)n + 1);
    int64_t undecided = 0;
    for (int64_t v = 0; v < n; ++v) undecided += st[v] == CST_U;
    int rounds = 0;
    while (undecided > 0) {
        ++rounds;
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            uint64_t m = 0;
            if (st[v] != CST_G) {
                if (st[v] == CST_U) m = key[v];
                for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
                    idx_t u = col[j];
                    if (strong[j] && u != v && st[u] == CST_U && key[u] > m) m = key[u];
                }
            }
            m1[v] = m;
        }
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            uint64_t m = m1[v];
            if (st[v] == CST_U)
                for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j)
                    if (strong[j] && col[j] != v && m1[col[j]] > m) m = m1[col[j]];
            m2[v] = m;
        }
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v)
            if (st[v] == CST_U && m2[v] == key[v]) st[v] = CST_S;
#pragma omp parallel for schedule(static)
        for (int64_t v = 0; v < n; ++v) {
            char c = st[v] == CST_S;
            if (!c && st[v] != CST_G)
                for (idx_t j = ptr[v]; j < ptr[v + 1] && !c; ++j)
                    if (strong[j] && st[col[j]] == CST_S) c = 1;
            c1[v] = c;
        }
        int64_t left = 0;
#pragma omp parallel for schedule(static) reduction(+ : left)
        for (int64_t v = 0; v < n; ++v) {
            if (st[v] != CST_U) continue;
            char c = c1[v];
            for (idx_t j = ptr[v]; j < ptr[v + 1] && !c; ++j)
                if (strong[j] && c1[col[j]]) c = 1;
            if (c) st[v] = CST_C; else ++left;
        }
        undecided = left;
    }
    free(m1); free(m2); free(c1);
    return rounds;
}

/* steps B / E: owner[v] for the vertices of the current graph (st != CST_G) that are seeds or next to one */
static void compact_claim(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, const char *st, int64_t *owner)
{
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) {
        if (st[v] == CST_G) continue;
        if (st[v] == CST_S) { owner[v] = v; continue; }
        int64_t best = -1;
        for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
            idx_t u = col[j];
            if (strong[j] && u != v && st[u] == CST_S && u > best) best = u;
        }
        if (best >= 0) owner[v] = best;
    }
}

static int64_t compact_aggregates_graph(int64_t n, const idx_t *ptr, const idx_t *col, const char *strong, idx_t *id,
                                        int *rounds_out)
{
    char *st = (char *)malloc((size_t)n + 1);
    uint64_t *key = (uint64_t *)malloc((size_t)n * 8 + 8);
    int64_t *owner = (int64_t *)malloc((size_t)n * 8 + 8), *nw = (int64_t *)malloc((size_t)n * 8 + 8);
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) {
        int any = 0;
        for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j)
            if (strong[j]) { any = 1; break; }
        st[v] = any ? CST_U : CST_G;
        owner[v] = any ? -1 : -2;
        key[v] = agg_key(v);
    }
    int rounds = mis2_rounds(n, ptr, col, strong, st, key);        /* A */
    compact_claim(n, ptr, col, strong, st, owner);                  /* B */
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) {                               /* C: the graph of the leftovers and its candidates */
        if (owner[v] != -1) { nw[v] = CST_G; continue; }
        int64_t deg = 0, lo = 0;
        for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
            idx_t u = col[j];
            if (!strong[j] || u == v) continue;
            ++deg;
            lo += owner[u] == -1;
        }
        nw[v] = (lo > 0 && 5 * lo >= 3 * deg) ? CST_U : CST_C;
    }
    for (int64_t v = 0; v < n; ++v) st[v] = (char)nw[v];
    rounds += mis2_rounds(n, ptr, col, strong, st, key);           /* D */
    compact_claim(n, ptr, col, strong, st, owner);                  /* E */
    for (int pass = 0; pass < 8; ++pass) {                          /* F */
        int64_t left = 0, moved = 0;
#pragma omp parallel for schedule(static) reduction(+ : left, moved)
        for (int64_t v = 0; v < n; ++v) {
            nw[v] = owner[v];
            if (owner[v] != -1) continue;
            int64_t best = -1, bc = 0;
            for (idx_t j = ptr[v]; j < ptr[v + 1]; ++j) {
                idx_t u = col[j];
                if (!strong[j] || u == v || owner[u] < 0) continue;
                const int64_t o = owner[u];
                if (o == best) continue;
                int64_t c = 0;
                for (idx_t k = ptr[v]; k < ptr[v + 1]; ++k)
                    if (strong[k] && col[k] != v && owner[col[k]] == o) ++c;
                if (c > bc || (c == bc && o < best)) { bc = c; best = o; }
            }
            if (best >= 0) { nw[v] = best; ++moved; } else ++left;
        }
        for (int64_t v = 0; v < n; ++v) owner[v] = nw[v];
        if (left == 0 || moved == 0) break;
    }
    for (int64_t v = 0; v < n; ++v)
        if (owner[v] == -1) owner[v] = v; /* (unsymmetric patterns only: nothing assigned in reach) */
    idx_t *rank = (idx_t *)malloc((size_t)n * sizeof(idx_t) + 8);
    int64_t count = 0;
    for (int64_t v = 0; v < n; ++v) rank[v] = owner[v] == v ? (idx_t)count++ : -1;
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n; ++v) id[v] = owner[v] == -2 ? AGG_REMOVED : rank[owner[v]];
    if (rounds_out) *rounds_out = rounds;
    free(st); free(key); free(owner); free(nw); free(rank);
    return count;
}

/* ---- amgcl/coarsening/smoothed_aggregation.hpp: transfer_operators ------------------------- */
/* With the default (no near-nullspace) tentative prolongation P_tent(i, id[i]) = 1. */
static csr_t *smoothed_prolongation(const csr_t *A, const char *strong, const idx_t *id, int64_t nagg, double omega)
{
    /* rows in parallel, one marker array per thread over contiguous row ranges (amgcl: an OpenMP loop): the rows
     * come out as from the sequential loop */
    const int64_t n = A->nrows;
    csr_t *P = (csr_t *)calloc(1, sizeof(csr_t));
    P->nrows = n;
    P->ncols = nagg;
    P->ptr = (idx_t *)calloc((size_t)n + 1, sizeof(idx_t));
#pragma omp parallel
    {
        int64_t *marker = (int64_t *)malloc((size_t)(nagg > 0 ? nagg : 1) * sizeof(int64_t));
        for (int64_t k = 0; k < nagg; ++k) marker[k] = -1;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            idx_t cnt = 0;
            for (idx_t ja = A->ptr[i]; ja < A->ptr[i + 1]; ++ja) {
                idx_t ca = A->col[ja];
                if (ca != i && !strong[ja]) continue;
                idx_t cp = id[ca];
                if (cp < 0) continue; /* empty P_tent row */
                if (marker[cp] != i) {
                    marker[cp] = i;
                    ++cnt;
                }
            }
            P->ptr[i + 1] = cnt;
        }
        free(marker);
    }
    for (int64_t i = 0; i < n; ++i) P->ptr[i + 1] += P->ptr[i];
    int64_t nnz = P->ptr[n];
    P->col = (idx_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(idx_t));
    P->val = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
#pragma omp parallel
    {
        int64_t *marker = (int64_t *)malloc((size_t)(nagg > 0 ? nagg : 1) * sizeof(int64_t));
        for (int64_t k = 0; k < nagg; ++k) marker[k] = -1;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            /* diagonal of the filtered matrix = diagonal minus (i.e. plus the values of) weak links */
            double dia = 0.0;
            for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j)
                if (A->col[j] == i || !strong[j]) dia += A->val[j];
            dia = -omega * (1.0 / dia);
            idx_t row_beg = P->ptr[i], row_end = row_beg;
            for (idx_t ja = A->ptr[i]; ja < A->ptr[i + 1]; ++ja) {
                idx_t ca = A->col[ja];
                if (ca != i && !strong[ja]) continue;
                double va = (ca == i) ? (1.0 - omega) : dia * A->val[ja];
                idx_t cp = id[ca];
                if (cp < 0) continue;
                if (marker[cp] < row_beg) {
                    marker[cp] = row_end;
                    P->col[row_end] = cp;
                    P->val[row_end] = va; /* va * 1.0 */
                    ++row_end;
                } else {
                    P->val[marker[cp]] += va;
                }
            }
        }
        free(marker);
    }
    return P;
}

static csr_t *csr_transpose(const csr_t *A)
{
    const int64_t n = A->nrows, m = A->ncols, nnz = A->ptr[n];
    csr_t *T = csr_alloc(m, n, nnz);
    for (int64_t j = 0; j < nnz; ++j) ++T->ptr[A->col[j] + 1];
    for (int64_t i = 0; i < m; ++i) T->ptr[i + 1] += T->ptr[i];
    idx_t *head = (idx_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(idx_t));
    memcpy(head, T->ptr, (size_t)m * sizeof(idx_t));
    for (int64_t i = 0; i < n; ++i)
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
            idx_t h = head[A->col[j]]++;
            T->col[h] = (idx_t)i;
            T->val[h] = A->val[j];
        }
    free(head);
    return T;
}

/* C = A * B, Gustavson row-by-row (amgcl/backend/detail/spgemm.hpp, `saad` variant). Column
 * order inside a row is first-touch order, as upstream when sort == false. */
static csr_t *csr_product(const csr_t *A, const csr_t *B)
{
    /* Row-parallel (amgcl's spgemm is an OpenMP loop over the rows of A with one marker array per thread): a thread
     * takes a contiguous range of rows, so its marker sees increasing row starts exactly like the sequential loop, and
     * every row comes out with the same columns in the same first-touch order and the same sums. */
    const int64_t n = A->nrows, m = B->ncols;
    csr_t *C = (csr_t *)calloc(1, sizeof(csr_t));
    C->nrows = n;
    C->ncols = m;
    C->ptr = (idx_t *)calloc((size_t)n + 1, sizeof(idx_t));
#pragma omp parallel
    {
        int64_t *marker = (int64_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
        for (int64_t k = 0; k < m; ++k) marker[k] = -1;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            idx_t cnt = 0;
            for (idx_t ja = A->ptr[i]; ja < A->ptr[i + 1]; ++ja) {
                idx_t ca = A->col[ja];
                for (idx_t jb = B->ptr[ca]; jb < B->ptr[ca + 1]; ++jb) {
                    idx_t cb = B->col[jb];
                    if (marker[cb] != i) { marker[cb] = i; ++cnt; }
                }
            }
            C->ptr[i + 1] = cnt;
        }
        free(marker);
    }
    for (int64_t i = 0; i < n; ++i) C->ptr[i + 1] += C->ptr[i];
    int64_t nnz = C->ptr[n];
    C->col = (idx_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(idx_t));
    C->val = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
#pragma omp parallel
    {
        int64_t *marker = (int64_t *)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
        for (int64_t k = 0; k < m; ++k) marker[k] = -1;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            idx_t row_beg = C->ptr[i], row_end = row_beg;
            for (idx_t ja = A->ptr[i]; ja < A->ptr[i + 1]; ++ja) {
                idx_t ca = A->col[ja];
                double va = A->val[ja];
                for (idx_t jb = B->ptr[ca]; jb < B->ptr[ca + 1]; ++jb) {
                    idx_t cb = B->col[jb];
                    double vb = B->val[jb];
                    if (marker[cb] < row_beg) {
                        marker[cb] = row_end;
                        C->col[row_end] = cb;
                        C->val[row_end] = va * vb;
                        ++row_end;
                    } else {
                        C->val[marker[cb]] += va * vb;
                    }
                }
            }
        }
        free(marker);
    }
    return C;
}


/* ==== block value types: polysolve::linear::AMGCL_Block<N> (AMGCL.cpp:243-302) ==================
 * amgcl::backend::builtin<static_matrix<double,N,N>> run through the same templates; restated here
 * on the scalar CSR with block size b:
 *   - aggregation works on the block graph (plain_aggregates on a matrix whose values are b x b
 *     blocks); "eps_dia_i * dia_c < v * v" compares static matrices, which AMGCL orders by trace
 *     [upstream, recalled: value_type/static_matrix.hpp]; with the reference's eps_strong = 0 this is
 *     "trace(v v) > 0";
 *   - the tentative prolongation is the b x b identity per node; smoothing uses the inverse of the
 *     (filtered) diagonal BLOCK; omega comes from the block Gershgorin bound with Frobenius norms;
 *   - chebyshev scales the residual with the inverted diagonal blocks; its power iteration starts
 *     from a block-constant random vector and sums |<s_i, b_i>| per block.
 * Transfer operators are stored with full b x b blocks (explicit zeros), as block arithmetic does. */
typedef struct {
    int64_t nb;      /* block rows */
    int b;
    idx_t *ptr, *col;
    double *val;     /* b*b per block, row-major */
} bcsr_t;

static void bcsr_free(bcsr_t *B)
{
    if (!B) return;
    free(B->ptr); free(B->col); free(B->val); free(B);
}

/* zero-filled block view of a scalar CSR whose size is a multiple of b; block columns sorted */
static bcsr_t *to_blocks(const csr_t *A, int b)
{
    const int64_t nb = A->nrows / b, ncb = A->ncols / b;
    bcsr_t *B = (bcsr_t *)calloc(1, sizeof(bcsr_t));
    B->nb = nb; B->b = b;
    B->ptr = (idx_t *)calloc((size_t)nb + 1, sizeof(idx_t));
    int64_t *marker = (int64_t *)malloc((size_t)(ncb > 0 ? ncb : 1) * sizeof(int64_t));
    for (int64_t k = 0; k < ncb; ++k) marker[k] = -1;
    for (int64_t ib = 0; ib < nb; ++ib) {
        idx_t cnt = 0;
        for (int r = 0; r < b; ++r)
            for (idx_t j = A->ptr[ib * b + r]; j < A->ptr[ib * b + r + 1]; ++j) {
                idx_t cb = A->col[j] / b;
                if (marker[cb] != ib) { marker[cb] = ib; ++cnt; }
            }
        B->ptr[ib + 1] = B->ptr[ib] + cnt;
    }
    const int64_t nnzb = B->ptr[nb];
    B->col = (idx_t *)malloc((size_t)(nnzb > 0 ? nnzb : 1) * sizeof(idx_t));
    B->val = (double *)calloc((size_t)(nnzb > 0 ? nnzb : 1) * b * b, sizeof(double));
    for (int64_t k = 0; k < ncb; ++k) marker[k] = -1;
    for (int64_t ib = 0; ib < nb; ++ib) {
        idx_t beg = B->ptr[ib], end = beg;
        for (int r = 0; r < b; ++r)
            for (idx_t j = A->ptr[ib * b + r]; j < A->ptr[ib * b + r + 1]; ++j) {
                idx_t cb = A->col[j] / b;
                if (marker[cb] < beg) { marker[cb] = end; B->col[end++] = cb; }
            }
        /* sort the block columns of this row (insertion sort: rows are short) */
        for (idx_t a = beg + 1; a < end; ++a) {
            idx_t v = B->col[a], k = a;
            while (k > beg && B->col[k - 1] > v) { B->col[k] = B->col[k - 1]; --k; }
            B->col[k] = v;
        }
        for (idx_t a = beg; a < end; ++a) marker[B->col[a]] = a;
        for (int r = 0; r < b; ++r)
            for (idx_t j = A->ptr[ib * b + r]; j < A->ptr[ib * b + r + 1]; ++j) {
                idx_t cb = A->col[j] / b, cc = A->col[j] % b;
                B->val[(size_t)marker[cb] * b * b + r * b + cc] += A->val[j];
            }
    }
    free(marker);
    return B;
}

static void blk_mul(int b, const double *X, const double *Y, double *Z) /* Z = X Y */
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < b; ++j) {
            double s = 0.0;
            for (int k = 0; k < b; ++k) s += X[i * b + k] * Y[k * b + j];
            Z[i * b + j] = s;
        }
}

static double blk_trace(int b, const double *X)
{
    double t = 0.0;
    for (int i = 0; i < b; ++i) t += X[i * b + i];
    return t;
}

static double blk_fro(int b, const double *X)
{
    double s = 0.0;
    for (int i = 0; i < b * b; ++i) s += X[i] * X[i];
    return sqrt(s);
}

/* inverse by Gauss-Jordan with partial pivoting (b <= 4) */
static void blk_inv(int b, const double *X, double *Y)
{
    double a[16], inv[16];
    for (int i = 0; i < b * b; ++i) { a[i] = X[i]; inv[i] = 0.0; }
    for (int i = 0; i < b; ++i) inv[i * b + i] = 1.0;
    for (int c = 0; c < b; ++c) {
        int piv = c;
        for (int r = c + 1; r < b; ++r)
            if (fabs(a[r * b + c]) > fabs(a[piv * b + c])) piv = r;
        if (piv != c)
            for (int k = 0; k < b; ++k) {
                double t = a[c * b + k]; a[c * b + k] = a[piv * b + k]; a[piv * b + k] = t;
                t = inv[c * b + k]; inv[c * b + k] = inv[piv * b + k]; inv[piv * b + k] = t;
            }
        const double d = 1.0 / a[c * b + c];
        for (int k = 0; k < b; ++k) { a[c * b + k] *= d; inv[c * b + k] *= d; }
        for (int r = 0; r < b; ++r) {
            if (r == c) continue;
            const double f = a[r * b + c];
            if (f == 0.0) continue;
            for (int k = 0; k < b; ++k) { a[r * b + k] -= f * a[c * b + k]; inv[r * b + k] -= f * inv[c * b + k]; }
        }
    }
    for (int i = 0; i < b * b; ++i) Y[i] = inv[i];
}

static const double *blk_diag(const bcsr_t *B, int64_t ib)
{
    for (idx_t j = B->ptr[ib]; j < B->ptr[ib + 1]; ++j)
        if (B->col[j] == ib) return B->val + (size_t)j * B->b * B->b;
    return NULL;
}

/* plain_aggregates on the block graph; id[nb] and strong[nnzb] */
static int64_t block_aggregates_mode(const bcsr_t *B, double eps_strong, char *strong, idx_t *id, int mode);
static int64_t __attribute__((unused)) block_aggregates(const bcsr_t *B, double eps_strong, char *strong, idx_t *id)
{
    return block_aggregates_mode(B, eps_strong, strong, id, 0);
}
static int64_t block_aggregates_mode(const bcsr_t *B, double eps_strong, char *strong, idx_t *id, int mode)
{
    const int64_t nb = B->nb;
    const int b = B->b, bb = b * b;
    const double eps2 = eps_strong * eps_strong;
    double tmp[16], tmp2[16], zero[16] = {0};
    for (int64_t i = 0; i < nb; ++i) {
        const double *di = blk_diag(B, i);
        for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j) {
            idx_t c = B->col[j];
            const double *v = B->val + (size_t)j * bb;
            const double *dc = blk_diag(B, c);
            blk_mul(b, v, v, tmp);                                   /* v * v */
            blk_mul(b, di ? di : zero, dc ? dc : zero, tmp2);        /* dia_i * dia_c */
            strong[j] = (c != i) && (eps2 * blk_trace(b, tmp2) < blk_trace(b, tmp));
        }
    }
    if (mode == 1) return parallel_aggregates_graph(nb, B->ptr, B->col, strong, id, NULL);
    if (mode == 2) return compact_aggregates_graph(nb, B->ptr, B->col, strong, id, NULL);
    /* the greedy sweep is the scalar one, on the block graph */
    csr_t G = {nb, nb, B->ptr, B->col, NULL};
    int64_t max_neib = 0;
    for (int64_t i = 0; i < nb; ++i) {
        idx_t j = G.ptr[i], e = G.ptr[i + 1];
        if (e - j > max_neib) max_neib = e - j;
        idx_t state = AGG_REMOVED;
        for (; j < e; ++j)
            if (strong[j]) { state = AGG_UNDEFINED; break; }
        id[i] = state;
    }
    idx_t *neib = (idx_t *)malloc((size_t)(max_neib + 1) * sizeof(idx_t));
    int64_t count = 0;
    for (int64_t i = 0; i < nb; ++i) {
        if (id[i] != AGG_UNDEFINED) continue;
        idx_t cur = (idx_t)count++;
        id[i] = cur;
        int64_t nn = 0;
        for (idx_t j = G.ptr[i]; j < G.ptr[i + 1]; ++j) {
            idx_t c = G.col[j];
            if (strong[j] && id[c] != AGG_REMOVED) { id[c] = cur; neib[nn++] = c; }
        }
        for (int64_t q = 0; q < nn; ++q) {
            idx_t c = neib[q];
            for (idx_t j = G.ptr[c]; j < G.ptr[c + 1]; ++j) {
                idx_t cc = G.col[j];
                if (strong[j] && id[cc] == AGG_UNDEFINED) id[cc] = cur;
            }
        }
    }
    free(neib);
    if (count == 0) return 0;
    idx_t *cnt = (idx_t *)calloc((size_t)count, sizeof(idx_t));
    for (int64_t i = 0; i < nb; ++i)
        if (id[i] >= 0) cnt[id[i]] = 1;
    for (int64_t k = 1; k < count; ++k) cnt[k] += cnt[k - 1];
    if (count > cnt[count - 1]) {
        int64_t newcount = cnt[count - 1];
        for (int64_t i = 0; i < nb; ++i)
            if (id[i] >= 0) id[i] = cnt[id[i]] - 1;
        count = newcount;
    }
    free(cnt);
    return count;
}

/* block Gershgorin bound of rho(D^-1 A): max_i (sum_j ||A_ij||_F) ||inv(D_i)||_F */
static double block_gershgorin(const bcsr_t *B)
{
    const int b = B->b, bb = b * b;
    double radius = 0.0, dia[16], inv[16];
    for (int i = 0; i < bb; ++i) dia[i] = (i % (b + 1) == 0) ? 1.0 : 0.0;
    for (int64_t i = 0; i < B->nb; ++i) {
        double s = 0.0;
        for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j) {
            s += blk_fro(b, B->val + (size_t)j * bb);
            if (B->col[j] == i) memcpy(dia, B->val + (size_t)j * bb, sizeof(double) * bb);
        }
        blk_inv(b, dia, inv);
        s *= blk_fro(b, inv);
        if (s > radius) radius = s;
    }
    return radius;
}

/* P = (I - omega D^-1 A_f) P_tent with b x b blocks; returned as scalar CSR with full blocks */
static csr_t *block_smoothed_prolongation(const bcsr_t *B, const char *strong, const idx_t *id, int64_t nagg,
                                          double omega)
{
    const int64_t nb = B->nb;
    const int b = B->b, bb = b * b;
    idx_t *bptr = (idx_t *)calloc((size_t)nb + 1, sizeof(idx_t));
    int64_t *marker = (int64_t *)malloc((size_t)(nagg > 0 ? nagg : 1) * sizeof(int64_t));
    for (int64_t k = 0; k < nagg; ++k) marker[k] = -1;
    for (int64_t i = 0; i < nb; ++i)
        for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j) {
            idx_t ca = B->col[j];
            if (ca != i && !strong[j]) continue;
            idx_t cp = id[ca];
            if (cp < 0) continue;
            if (marker[cp] != i) { marker[cp] = i; ++bptr[i + 1]; }
        }
    for (int64_t i = 0; i < nb; ++i) bptr[i + 1] += bptr[i];
    const int64_t nnzb = bptr[nb];
    idx_t *bcol = (idx_t *)malloc((size_t)(nnzb > 0 ? nnzb : 1) * sizeof(idx_t));
    double *bval = (double *)calloc((size_t)(nnzb > 0 ? nnzb : 1) * bb, sizeof(double));
    for (int64_t k = 0; k < nagg; ++k) marker[k] = -1;
    double dia[16], dinv[16], va[16];
    for (int64_t i = 0; i < nb; ++i) {
        for (int k = 0; k < bb; ++k) dia[k] = 0.0;
        for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j)
            if (B->col[j] == i || !strong[j])
                for (int k = 0; k < bb; ++k) dia[k] += B->val[(size_t)j * bb + k];
        blk_inv(b, dia, dinv);
        for (int k = 0; k < bb; ++k) dinv[k] *= -omega;
        idx_t row_beg = bptr[i], row_end = row_beg;
        for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j) {
            idx_t ca = B->col[j];
            if (ca != i && !strong[j]) continue;
            if (ca == i) {
                for (int k = 0; k < bb; ++k) va[k] = (k % (b + 1) == 0) ? (1.0 - omega) : 0.0;
            } else {
                blk_mul(b, dinv, B->val + (size_t)j * bb, va);
            }
            idx_t cp = id[ca];
            if (cp < 0) continue;
            if (marker[cp] < row_beg) {
                marker[cp] = row_end;
                bcol[row_end] = cp;
                memcpy(bval + (size_t)row_end * bb, va, sizeof(double) * bb);
                ++row_end;
            } else {
                for (int k = 0; k < bb; ++k) bval[(size_t)marker[cp] * bb + k] += va[k];
            }
        }
    }
    free(marker);
    /* expand to scalar CSR, full blocks */
    csr_t *P = csr_alloc(nb * b, nagg * b, nnzb * bb);
    int64_t p = 0;
    for (int64_t i = 0; i < nb; ++i)
        for (int r = 0; r < b; ++r) {
            for (idx_t j = bptr[i]; j < bptr[i + 1]; ++j)
                for (int c = 0; c < b; ++c) {
                    P->col[p] = bcol[j] * b + c;
                    P->val[p++] = bval[(size_t)j * bb + r * b + c];
                }
            P->ptr[i * b + r + 1] = (idx_t)p;
        }
    free(bptr); free(bcol); free(bval);
    return P;
}

/* block power iteration: rho(D^-1 A), D = diagonal blocks */
static double block_spectral_radius(const csr_t *A, int b, const double *Minv, int power_iters)
{
    const int64_t n = A->nrows, nb = n / b;
    double *b0 = (double *)malloc((size_t)n * 8), *b1 = (double *)malloc((size_t)n * 8), *t = (double *)malloc((size_t)n * 8);
    mt19937_t rng;
    mt_seed(&rng, 0u);
    double b0_norm = 0.0;
    for (int64_t i = 0; i < nb; ++i) {
        double v = mt_uniform_pm1(&rng); /* math::constant<rhs_type>(rnd(rng)) */
        for (int k = 0; k < b; ++k) b0[i * b + k] = v;
        b0_norm += b * v * v;
    }
    b0_norm = 1.0 / sqrt(b0_norm);
    for (int64_t i = 0; i < n; ++i) b0[i] = b0_norm * b0[i];
    double radius = 0.0;
    for (int iter = 0; iter < power_iters;) {
        double b1_norm = 0.0;
        radius = 0.0;
        csr_spmv(1.0, A, b0, 0.0, t);
        for (int64_t i = 0; i < nb; ++i) {
            double dotsb = 0.0;
            for (int r = 0; r < b; ++r) {
                double s = 0.0;
                for (int c = 0; c < b; ++c) s += Minv[(size_t)i * b * b + r * b + c] * t[i * b + c];
                b1[i * b + r] = s;
                b1_norm += s * s;
                dotsb += s * b0[i * b + r];
            }
            radius += fabs(dotsb);
        }
        if (++iter < power_iters) {
            b1_norm = 1.0 / sqrt(b1_norm);
            for (int64_t i = 0; i < n; ++i) b0[i] = b1_norm * b1[i];
        }
    }
    free(b0); free(b1); free(t);
    return radius < 0 ? 2.0 : radius;
}

/* ---- amgcl/relaxation/chebyshev.hpp ------------------------------------------------------- */
typedef struct {
    int degree, scale;
    double d, c;     /* centre / semi-axis of the eigenvalue interval */
    double *M;       /* inverted diagonal (scale == true) */
    double *p, *r;   /* work vectors */
    double rho;      /* the estimated spectral radius, for inspection */
    int bs;          /* block size (1 = scalar) */
    double *Mb;      /* bs > 1: inverted diagonal blocks */
    int type;        /* 0 chebyshev; 1 damped_jacobi, 2 spai0: M / Mb is the whole scaling (damping included), one step
                        x += M (rhs - A x) per application; 3 gauss_seidel, 4 ilu0: sweeps over S */
    bcsr_t *S;       /* types 3, 4: the level's matrix in b x b blocks (b = 1: one scalar per block), sorted rows; ilu0: the
                        factors in place (strictly lower part L without its unit diagonal, strictly upper part U) */
    double *Dinv;    /* types 3, 4: inverted diagonal blocks (ilu0: of the factored diagonal) */
    double damping;  /* ilu0: x += damping * (LU)^-1 (rhs - A x) */
} cheby_t;

/* ---- amgcl/relaxation/damped_jacobi.hpp, spai0.hpp (round 5; amgcl::runtime::relaxation, reached from the reference by
 * "/AMGCL/precond/relax/type", linear-solver-spec.json:393-397, AMGCL.cpp:67-92) ---------------------------------------
 * damped_jacobi: dia = inverted diagonal (blocks); apply: tmp = rhs - A x; x = damping * dia * tmp + x
 *                (backend::vmul: z = a * x * y + b * z, evaluated left to right: the scaling a * dia first).
 * spai0:         M_i = (1 / sum_j |a_ij|^2) a_ii  (blocks: Frobenius norms, the diagonal block); apply: x = M tmp + x.
 * Both are stored here as ONE scaling M (damping folded in), applied like the scaled Chebyshev residual. */
static cheby_t *jacobi_like_create(const csr_t *A, int type, double damping, int bs)
{
    cheby_t *C = (cheby_t *)calloc(1, sizeof(cheby_t));
    const int64_t n = A->nrows;
    C->type = type; C->bs = bs; C->degree = 1; C->scale = 1;
    C->p = (double *)calloc((size_t)n, 8);
    C->r = (double *)calloc((size_t)n, 8);
    if (bs > 1) {
        bcsr_t *B = to_blocks(A, bs);
        const int bb = bs * bs;
        C->Mb = (double *)calloc((size_t)B->nb * bb, 8);
        for (int64_t i = 0; i < B->nb; ++i) {
            double *M = C->Mb + (size_t)i * bb;
            const double *d = blk_diag(B, i);
            if (type == 1) {
                if (d) { blk_inv(bs, d, M); for (int k = 0; k < bb; ++k) M[k] = damping * M[k]; }
            } else {
                double den = 0.0;
                for (idx_t j = B->ptr[i]; j < B->ptr[i + 1]; ++j) {
                    const double *v = B->val + (size_t)j * bb;
                    double s2 = 0.0;
                    for (int k = 0; k < bb; ++k) s2 += v[k] * v[k];
                    double nv = sqrt(s2);
                    den += nv * nv;
                }
                double inv = 1.0 / den;
                if (d) for (int k = 0; k < bb; ++k) M[k] = inv * d[k];
            }
        }
        bcsr_free(B);
        return C;
    }
    C->M = (double *)calloc((size_t)n, 8);
    for (int64_t i = 0; i < n; ++i) {
        double num = 0.0, den = 0.0;
        int has = 0;
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
            double v = A->val[j], nv = fabs(v);
            den += nv * nv;
            if (A->col[j] == i) { num += v; has = 1; }
        }
        if (type == 1) C->M[i] = has ? damping * (1.0 / num) : 0.0;
        else C->M[i] = (1.0 / den) * num;
    }
    return C;
}

/* ---- amgcl/relaxation/gauss_seidel.hpp, ilu0.hpp + detail/ilu_solve.hpp (round 6; "/AMGCL/precond/relax/type",
 * linear-solver-spec.json:393-397) -- restated for the builtin backend's SERIAL order, which its parallel sweeps reproduce
 * (they only schedule rows by their dependency levels).
 * gauss_seidel: apply_pre = one forward sweep, apply_post = one backward sweep; a row walks its entries in storage order:
 *               X = rhs_i; D = I; for j: c == i ? D = a_ij : X -= a_ij x_c;  x_i = inverse(D) X.
 * ilu0:         IKJ elimination on A's pattern (rows sorted): for the lower entries c of row i in order, tl = a_ic D_c
 *               (D_c already inverted), a_ic = tl, and every entry k of row c's upper part that row i also stores gets
 *               a_ik -= tl u_ck; then D_i = inverse(a_ii).  solve: forward x_i -= l_ic x_c, backward x_i -= u_ic x_c,
 *               x_i = D_i x_i.  apply_pre = apply_post: t = rhs - A x; solve(t); x = damping t + x. */
static void blk_matvec_sub(int b, const double *V, const double *x, double *X) /* X -= V x */
{
    for (int r = 0; r < b; ++r) {
        double s = 0.0;
        for (int q = 0; q < b; ++q) s += V[r * b + q] * x[q];
        X[r] -= s;
    }
}

static void blk_matvec(int b, const double *V, const double *x, double *y) /* y = V x (y != x) */
{
    for (int r = 0; r < b; ++r) {
        double s = 0.0;
        for (int q = 0; q < b; ++q) s += V[r * b + q] * x[q];
        y[r] = s;
    }
}

static void gs_sweep(const bcsr_t *S, const double *Dinv, const double *rhs, double *x, int backward)
{
    const int b = S->b, bb = b * b;
    for (int64_t t = 0; t < S->nb; ++t) {
        const int64_t i = backward ? S->nb - 1 - t : t;
        double X[4], y[4];
        for (int r = 0; r < b; ++r) X[r] = rhs[i * b + r];
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1]; ++j) {
            const idx_t c = S->col[j];
            if (c == i) continue;
            blk_matvec_sub(b, S->val + (size_t)j * bb, x + (size_t)c * b, X);
        }
        blk_matvec(b, Dinv + (size_t)i * bb, X, y);
        for (int r = 0; r < b; ++r) x[i * b + r] = y[r];
    }
}

/* 0: ok, 1: a row without its diagonal (amgcl: "No diagonal value in system matrix") */
static int ilu0_factor(bcsr_t *S, double *Dinv)
{
    const int b = S->b, bb = b * b;
    double **work = (double **)calloc((size_t)S->nb + 1, sizeof(double *));
    int bad = 0;
    for (int64_t i = 0; i < S->nb && !bad; ++i) {
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1]; ++j) work[S->col[j]] = S->val + (size_t)j * bb;
        int met = 0;
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1]; ++j) {
            const idx_t c = S->col[j];
            if (c >= i) {
                if (c == i) { blk_inv(b, S->val + (size_t)j * bb, Dinv + (size_t)i * bb); met = 1; }
                break;
            }
            double tl[16], prod[16];
            blk_mul(b, work[c], Dinv + (size_t)c * bb, tl);
            memcpy(work[c], tl, (size_t)bb * 8);
            for (idx_t k = S->ptr[c]; k < S->ptr[c + 1]; ++k) {
                if (S->col[k] <= c) continue; /* row c's upper part */
                double *w = work[S->col[k]];
                if (!w) continue;
                blk_mul(b, tl, S->val + (size_t)k * bb, prod);
                for (int e = 0; e < bb; ++e) w[e] -= prod[e];
            }
        }
        if (!met) bad = 1;
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1]; ++j) work[S->col[j]] = NULL;
    }
    free(work);
    return bad;
}

static void ilu0_solve(const bcsr_t *S, const double *Dinv, double *x)
{
    const int b = S->b, bb = b * b;
    for (int64_t i = 0; i < S->nb; ++i)
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1] && S->col[j] < i; ++j)
            blk_matvec_sub(b, S->val + (size_t)j * bb, x + (size_t)S->col[j] * b, x + (size_t)i * b);
    for (int64_t i = S->nb - 1; i >= 0; --i) {
        double y[4];
        for (idx_t j = S->ptr[i]; j < S->ptr[i + 1]; ++j)
            if (S->col[j] > i) blk_matvec_sub(b, S->val + (size_t)j * bb, x + (size_t)S->col[j] * b, x + (size_t)i * b);
        blk_matvec(b, Dinv + (size_t)i * bb, x + (size_t)i * b, y);
        for (int r = 0; r < b; ++r) x[i * b + r] = y[r];
    }
}

static cheby_t *sweep_create(const csr_t *A, int type, double damping, int bs)
{
    cheby_t *C = (cheby_t *)calloc(1, sizeof(cheby_t));
    const int64_t n = A->nrows;
    C->type = type; C->bs = bs; C->degree = 1; C->scale = 1; C->damping = damping;
    C->p = (double *)calloc((size_t)n, 8);
    C->r = (double *)calloc((size_t)n, 8);
    C->S = to_blocks(A, bs);
    const int bb = bs * bs;
    C->Dinv = (double *)calloc((size_t)C->S->nb * bb, 8);
    if (type == 4) {
        if (ilu0_factor(C->S, C->Dinv)) fprintf(stderr, "[oracle] ilu0: no diagonal value in system matrix\n");
        return C;
    }
    for (int64_t i = 0; i < C->S->nb; ++i) {
        const double *d = blk_diag(C->S, i);
        double ident[16];
        for (int k = 0; k < bb; ++k) ident[k] = (k % (bs + 1) == 0) ? 1.0 : 0.0;
        blk_inv(bs, d ? d : ident, C->Dinv + (size_t)i * bb);
    }
    return C;
}

static cheby_t *cheby_create_bs(const csr_t *A, int degree, int power_iters, double higher, double lower, int scale,
                                int bs);
static cheby_t *cheby_create(const csr_t *A, int degree, int power_iters, double higher, double lower, int scale)
{
    return cheby_create_bs(A, degree, power_iters, higher, lower, scale, 1);
}

static cheby_t *cheby_create_bs(const csr_t *A, int degree, int power_iters, double higher, double lower, int scale,
                                int bs)
{
    cheby_t *C = (cheby_t *)calloc(1, sizeof(cheby_t));
    const int64_t n = A->nrows;
    C->degree = degree;
    C->scale = scale;
    C->bs = bs;
    C->p = (double *)calloc((size_t)n, 8);
    C->r = (double *)calloc((size_t)n, 8);
    if (bs > 1) {
        bcsr_t *B = to_blocks(A, bs);
        const int bb = bs * bs;
        C->Mb = (double *)malloc((size_t)B->nb * bb * 8);
        for (int64_t i = 0; i < B->nb; ++i) {
            const double *d = blk_diag(B, i);
            double ident[16];
            for (int k = 0; k < bb; ++k) ident[k] = (k % (bs + 1) == 0) ? 1.0 : 0.0;
            if (scale) blk_inv(bs, d ? d : ident, C->Mb + (size_t)i * bb);
            else memcpy(C->Mb + (size_t)i * bb, ident, (size_t)bb * 8); /* chebyshev.scale = false: no scaling (1.0 * r = r) */
        }
        double hi = power_iters > 0 ? block_spectral_radius(A, bs, C->Mb, power_iters) : block_gershgorin(B);
        bcsr_free(B);
        C->rho = hi;
        double lo = hi * lower;
        hi *= higher;
        C->d = 0.5 * (hi + lo);
        C->c = 0.5 * (hi - lo);
        return C;
    }
    if (scale) {
        C->M = (double *)malloc((size_t)n * 8);
        for (int64_t i = 0; i < n; ++i) {
            double d = 1.0;
            for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j)
                if (A->col[j] == i) { d = A->val[j]; break; }
            C->M[i] = 1.0 / d;
        }
    }
    double hi = spectral_radius(A, scale, power_iters);
    C->rho = hi;
    double lo = hi * lower;
    hi *= higher;
    C->d = 0.5 * (hi + lo);
    C->c = 0.5 * (hi - lo);
    return C;
}

static void cheby_free(cheby_t *C)
{
    if (!C) return;
    free(C->M); free(C->Mb); free(C->p); free(C->r); bcsr_free(C->S); free(C->Dinv); free(C);
}

/* chebyshev::solve -- apply_pre and apply_post both call it (post: the sweeps' apply_post). */
static void relax_apply(cheby_t *C, const csr_t *A, const double *rhs, double *x, int post)
{
    const int64_t n = A->nrows;
    double alpha = 0.0, beta = 0.0;
    const double d = C->d, c = C->c;
    if (C->type == 3) { gs_sweep(C->S, C->Dinv, rhs, x, post); return; }
    if (C->type == 4) {
        csr_residual(rhs, A, x, C->r);
        ilu0_solve(C->S, C->Dinv, C->r);
        for (int64_t i = 0; i < n; ++i) x[i] = C->damping * C->r[i] + x[i];
        return;
    }
    if (C->type != 0) { /* damped_jacobi / spai0: x = M (rhs - A x) + x */
        csr_residual(rhs, A, x, C->r);
        if (C->bs > 1) {
            const int b = C->bs;
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n / b; ++i)
                for (int r = 0; r < b; ++r) {
                    double sacc = 0.0;
                    for (int q = 0; q < b; ++q) sacc += C->Mb[(size_t)i * b * b + r * b + q] * C->r[i * b + q];
                    x[i * b + r] = sacc + x[i * b + r];
                }
        } else {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) x[i] = C->M[i] * C->r[i] + x[i];
        }
        return;
    }
    for (int k = 0; k < C->degree; ++k) {
        csr_residual(rhs, A, x, C->r);
        if (C->bs > 1) {
            const int b = C->bs;
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n / b; ++i) {
                double t[4];
                for (int r = 0; r < b; ++r) {
                    double sacc = 0.0;
                    for (int c = 0; c < b; ++c) sacc += C->Mb[(size_t)i * b * b + r * b + c] * C->r[i * b + c];
                    t[r] = sacc;
                }
                for (int r = 0; r < b; ++r) C->r[i * b + r] = t[r];
            }
        } else if (C->scale) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) C->r[i] = C->M[i] * C->r[i];
        }
        if (k == 0) {
            alpha = 1.0 / d;
            beta = 0.0;
        } else if (k == 1) {
            alpha = 2 * d * (1.0 / (2 * d * d - c * c));
            beta = alpha * d - 1.0;
        } else {
            alpha = 1.0 / (d - 0.25 * alpha * c * c);
            beta = alpha * d - 1.0;
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            /* axpby(alpha, r, beta, p): beta == 0 assigns (no 0 * NaN) as AMGCL's backend does */
            C->p[i] = (beta != 0.0) ? alpha * C->r[i] + beta * C->p[i] : alpha * C->r[i];
            x[i] += C->p[i];
        }
    }
}

/* ---- amgcl/amg.hpp ------------------------------------------------------------------------ */
typedef struct {
    csr_t *A, *P, *R;
    cheby_t *relax;
    double *f, *u, *t;
    int64_t nagg;   /* aggregates produced when coarsening this level (0 on the coarsest) */
    double omega;
    double *chol;   /* direct_coarse: dense Cholesky factor of the coarsest operator (lower triangle, row-major n x n) */
} level_t;

/* ---- direct_coarse = true (amgcl/amg.hpp: the coarsest level gets a direct solver; builtin backend: skyline_lu.hpp) ------
 * restated as a DENSE Cholesky factorization of the (SPD, at most coarse_enough rows) coarsest operator: the same solution
 * up to rounding, not skyline_lu's elimination order. */
static double *dense_cholesky(const csr_t *A)
{
    const int64_t n = A->nrows;
    double *L = (double *)calloc((size_t)n * n, 8);
    for (int64_t i = 0; i < n; ++i)
        for (idx_t j = A->ptr[i]; j < A->ptr[i + 1]; ++j)
            if (A->col[j] <= i) L[i * n + A->col[j]] += A->val[j];
    for (int64_t j = 0; j < n; ++j) {
        double d = L[j * n + j];
        for (int64_t k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        d = sqrt(d);
        L[j * n + j] = d;
#pragma omp parallel for schedule(static)
        for (int64_t i = j + 1; i < n; ++i) {
            double v = L[i * n + j];
            for (int64_t k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / d;
        }
    }
    return L;
}
static void dense_cholesky_solve(const double *L, int64_t n, const double *rhs, double *x)
{
    for (int64_t i = 0; i < n; ++i) {
        double v = rhs[i];
        for (int64_t k = 0; k < i; ++k) v -= L[i * n + k] * x[k];
        x[i] = v / L[i * n + i];
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int64_t k = i + 1; k < n; ++k) v -= L[k * n + i] * x[k];
        x[i] = v / L[i * n + i];
    }
}

struct orc_amg {
    int nlevels;
    level_t lv[64];
    int ncycle, npre, npost, pre_cycles;
    int relax_only; /* "/AMGCL/precond/class" = "relaxation": amgcl::relaxation::as_preconditioner, one level, relax.apply */
};

/* params mirror AMGCL.cpp:32-65 + amgcl defaults (coarse_enough 3000 for a scalar skyline_lu,
 * npre = npost = 1, pre_cycles = 1).  direct_coarse is false in the reference configuration:
 * the coarsest level is relaxed, not factorised. */
struct orc_amg *orc_amg_create_bs(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int max_levels,
                                  int coarse_enough, int ncycle, int npre, int npost, double eps_strong,
                                  double sa_relax, int estimate_spectral_radius, int sa_power_iters, int cheb_degree,
                                  int cheb_power_iters, double cheb_higher, double cheb_lower, int cheb_scale,
                                  int block_size);

struct orc_amg *orc_amg_create(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int max_levels,
                               int coarse_enough, int ncycle, int npre, int npost, double eps_strong,
                               double sa_relax, int estimate_spectral_radius, int sa_power_iters, int cheb_degree,
                               int cheb_power_iters, double cheb_higher, double cheb_lower, int cheb_scale)
{
    return orc_amg_create_bs(n, rowptr, col, val, max_levels, coarse_enough, ncycle, npre, npost, eps_strong, sa_relax,
                             estimate_spectral_radius, sa_power_iters, cheb_degree, cheb_power_iters, cheb_higher,
                             cheb_lower, cheb_scale, 1);
}

/* Options of orc_amg_create_ex, by index of a double array (ctypes-friendly; missing trailing entries keep their defaults).
 * 0-14: the arguments of orc_amg_create_bs in order.  Round 5: 15 aggregation (0 amgcl's sweep, 1 "parallel", 2 "compact"),
 * 16 coarsening (0 smoothed_aggregation, 1 aggregation: P = P_tent, A_c scaled by 1 / over_interp), 17 over_interp
 * (0: amgcl's default, 1.5f scalar / 2.0f block value types), 18 relax type (0 chebyshev, 1 damped_jacobi, 2 spai0),
 * 19 damping (damped_jacobi; amgcl's default 0.72), 20 direct_coarse. */
enum { ORC_AMG_NOPTS = 23 };

/* tentative prolongation without near-nullspace vectors (amgcl/coarsening/tentative_prolongation.hpp): P(i, id[i]) = 1;
 * block value types: the identity block */
static csr_t *tentative_prolongation(int64_t n_nodes, const idx_t *id, int64_t nagg, int bs)
{
    int64_t nnz = 0;
    for (int64_t i = 0; i < n_nodes; ++i) nnz += id[i] >= 0 ? 1 : 0;
    csr_t *P = csr_alloc(n_nodes * bs, nagg * bs, nnz * bs * bs);
    int64_t k = 0;
    P->ptr[0] = 0;
    for (int64_t i = 0; i < n_nodes; ++i)
        for (int r = 0; r < bs; ++r) {
            if (id[i] >= 0)
                for (int c = 0; c < bs; ++c) {
                    P->col[k] = id[i] * bs + c;
                    P->val[k] = r == c ? 1.0 : 0.0;
                    ++k;
                }
            P->ptr[i * bs + r + 1] = (idx_t)k;
        }
    return P;
}

static cheby_t *relax_create(const csr_t *A, const double *o, int bs)
{
    const int type = (int)o[18];
    if (type == 3 || type == 4) return sweep_create(A, type, o[21], bs);
    if (type != 0) return jacobi_like_create(A, type, o[19], bs);
    return cheby_create_bs(A, (int)o[9], (int)o[10], o[11], o[12], (int)o[13], bs);
}

struct orc_amg *orc_amg_create_ex(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *opts,
                                  int nopts)
{
    double o[ORC_AMG_NOPTS] = {6, 3000, 2, 1, 1, 0.0, 1.0, 1, 0, 16, 100, 2.0, 1.0 / 120, 1, 1, 0, 0, 0, 0, 0.72, 0, 1.0, 0};
    for (int k = 0; k < nopts && k < ORC_AMG_NOPTS; ++k) o[k] = opts[k];
    const int max_levels = (int)o[0], coarse_enough = (int)o[1];
    const double sa_relax = o[6];
    const int estimate_spectral_radius = (int)o[7], sa_power_iters = (int)o[8];
    const int bs = (int)o[14] > 1 ? (int)o[14] : 1;
    const int agg_mode = (int)o[15], coarsening = (int)o[16], direct_coarse = (int)o[20];
    const float over_interp = o[17] > 0 ? (float)o[17] : (bs == 1 ? 1.5f : 2.0f);
    struct orc_amg *h = (struct orc_amg *)calloc(1, sizeof(struct orc_amg));
    h->ncycle = (int)o[2]; h->npre = (int)o[3]; h->npost = (int)o[4]; h->pre_cycles = 1;
    int64_t nnz = rowptr[n];
    csr_t *A = csr_alloc(n, n, nnz);
    memcpy(A->ptr, rowptr, (size_t)(n + 1) * sizeof(idx_t));
    memcpy(A->col, col, (size_t)nnz * sizeof(idx_t));
    memcpy(A->val, val, (size_t)nnz * 8);

    double eps = o[5];
    if ((int)o[22] == 1) { /* class = relaxation: the smoother of the system matrix is the whole preconditioner */
        level_t *L = &h->lv[h->nlevels++];
        h->relax_only = 1;
        L->A = A;
        L->t = (double *)calloc((size_t)A->nrows, 8);
        L->relax = relax_create(A, o, bs);
        return h;
    }
    while (A->nrows > coarse_enough) {
        level_t *L = &h->lv[h->nlevels++];
        L->A = A;
        L->t = (double *)calloc((size_t)A->nrows, 8);
        if (h->nlevels > 1) {
            L->f = (double *)calloc((size_t)A->nrows, 8);
            L->u = (double *)calloc((size_t)A->nrows, 8);
        }
        if (h->nlevels >= max_levels) { A = NULL; break; }
        L->relax = relax_create(A, o, bs);
        /* step_down: transfer operators + Galerkin product */
        double omega = sa_relax;
        int64_t nagg;
        if (bs > 1) {
            bcsr_t *B = to_blocks(A, bs);
            char *strong = (char *)malloc((size_t)B->ptr[B->nb] + 1);
            idx_t *id = (idx_t *)malloc((size_t)B->nb * sizeof(idx_t));
            nagg = block_aggregates_mode(B, eps, strong, id, agg_mode);
            eps *= 0.5;
            if (nagg == 0) { free(strong); free(id); bcsr_free(B); A = NULL; break; }
            if (coarsening == 1) {
                L->P = tentative_prolongation(B->nb, id, nagg, bs);
            } else {
                if (estimate_spectral_radius) omega *= (4.0 / 3.0) / block_gershgorin(B);
                else omega *= 2.0 / 3.0;
                L->P = block_smoothed_prolongation(B, strong, id, nagg, omega);
            }
            free(strong); free(id); bcsr_free(B);
        } else {
            char *strong = (char *)malloc((size_t)A->ptr[A->nrows] + 1);
            idx_t *id = (idx_t *)malloc((size_t)A->nrows * sizeof(idx_t));
            nagg = plain_aggregates_mode(A, eps, strong, id, agg_mode);
            eps *= 0.5;
            if (nagg == 0) { free(strong); free(id); A = NULL; break; } /* error::empty_level */
            if (coarsening == 1) {
                L->P = tentative_prolongation(A->nrows, id, nagg, 1);
            } else {
                if (estimate_spectral_radius)
                    omega *= (4.0 / 3.0) / spectral_radius(A, 1, sa_power_iters);
                else
                    omega *= 2.0 / 3.0;
                L->P = smoothed_prolongation(A, strong, id, nagg, omega);
            }
            free(strong); free(id);
        }
        L->nagg = nagg;
        L->omega = omega;
        L->R = csr_transpose(L->P);
        csr_t *AP = csr_product(A, L->P);
        csr_t *Ac = csr_product(L->R, AP);
        csr_free(AP);
        if (coarsening == 1) { /* amgcl/coarsening/aggregation.hpp: detail::scaled_galerkin(A, P, R, 1 / over_interp), a float */
            const float sf = 1 / over_interp;
            const double sd = (double)sf;
            for (int64_t k = 0; k < Ac->ptr[Ac->nrows]; ++k) Ac->val[k] = sd * Ac->val[k];
        }
        A = Ac;
    }
    /* (the level that stopped at max_levels got no smoother above) */
    if (h->nlevels > 0 && !h->lv[h->nlevels - 1].relax && !A) {
        level_t *L = &h->lv[h->nlevels - 1];
        if (direct_coarse) L->chol = dense_cholesky(L->A);
        else L->relax = relax_create(L->A, o, bs);
    }
    if (A) {
        /* coarsest level: direct_coarse == false => smoother only (the reference's configuration, AMGCL.cpp:46) */
        level_t *L = &h->lv[h->nlevels++];
        L->A = A;
        if (direct_coarse) L->chol = dense_cholesky(A);
        else L->relax = relax_create(A, o, bs);
        L->t = (double *)calloc((size_t)A->nrows, 8);
        if (h->nlevels > 1) {
            L->f = (double *)calloc((size_t)A->nrows, 8);
            L->u = (double *)calloc((size_t)A->nrows, 8);
        }
    }
    return h;
}

/* block_size > 1: AMGCL_Block<N> (AMGCL.cpp:243-302); coarse_enough stays in SCALAR rows (the
 * skyline_lu limit is 3000 / N block rows, amgcl/solver/skyline_lu.hpp) */
struct orc_amg *orc_amg_create_bs(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int max_levels,
                                  int coarse_enough, int ncycle, int npre, int npost, double eps_strong,
                                  double sa_relax, int estimate_spectral_radius, int sa_power_iters, int cheb_degree,
                                  int cheb_power_iters, double cheb_higher, double cheb_lower, int cheb_scale,
                                  int block_size)
{
    const double o[15] = {max_levels, coarse_enough, ncycle, npre, npost, eps_strong, sa_relax, estimate_spectral_radius,
                          sa_power_iters, cheb_degree, cheb_power_iters, cheb_higher, cheb_lower, cheb_scale, block_size};
    return orc_amg_create_ex(n, rowptr, col, val, o, 15);
}

void orc_amg_destroy(struct orc_amg *h)
{
    if (!h) return;
    for (int l = 0; l < h->nlevels; ++l) {
        level_t *L = &h->lv[l];
        csr_free(L->A); csr_free(L->P); csr_free(L->R);
        cheby_free(L->relax);
        free(L->f); free(L->u); free(L->t); free(L->chol);
    }
    free(h);
}

static void amg_cycle(struct orc_amg *h, int l, const double *rhs, double *x)
{
    level_t *L = &h->lv[l];
    if (l + 1 == h->nlevels) {
        if (L->chol) { dense_cholesky_solve(L->chol, L->A->nrows, rhs, x); return; }
        for (int i = 0; i < h->npre; ++i) relax_apply(L->relax, L->A, rhs, x, 0);
        for (int i = 0; i < h->npost; ++i) relax_apply(L->relax, L->A, rhs, x, 1);
        return;
    }
    level_t *N = &h->lv[l + 1];
    for (int j = 0; j < h->ncycle; ++j) {
        for (int i = 0; i < h->npre; ++i) relax_apply(L->relax, L->A, rhs, x, 0);
        csr_residual(rhs, L->A, x, L->t);
        csr_spmv(1.0, L->R, L->t, 0.0, N->f);
        memset(N->u, 0, (size_t)N->A->nrows * 8);
        amg_cycle(h, l + 1, N->f, N->u);
        csr_spmv(1.0, L->P, N->u, 1.0, x);
        for (int i = 0; i < h->npost; ++i) relax_apply(L->relax, L->A, rhs, x, 1);
    }
}

/* amg::apply(rhs, x): x = 0; pre_cycles (=1) cycles. */
void orc_amg_apply(struct orc_amg *h, const double *rhs, double *x)
{
    if (h->relax_only) { /* relaxation::as_preconditioner::apply -> relax.apply(A, rhs, x) */
        level_t *L = &h->lv[0];
        const int64_t n = L->A->nrows;
        if (L->relax->type == 4) { /* ilu0::apply: copy(rhs, x); ilu->solve(x) */
            memcpy(x, rhs, (size_t)n * 8);
            ilu0_solve(L->relax->S, L->relax->Dinv, x);
            return;
        }
        memset(x, 0, (size_t)n * 8);
        relax_apply(L->relax, L->A, rhs, x, 0); /* chebyshev: clear + solve; damped_jacobi / spai0: M rhs; gauss_seidel: forward */
        if (L->relax->type == 3) relax_apply(L->relax, L->A, rhs, x, 1); /* ... then backward */
        return;
    }
    memset(x, 0, (size_t)h->lv[0].A->nrows * 8);
    for (int i = 0; i < h->pre_cycles; ++i) amg_cycle(h, 0, rhs, x);
}

int orc_amg_num_levels(const struct orc_amg *h) { return h->nlevels; }

/* what: 0 = A, 1 = P, 2 = R.  out[0..2] = nrows, ncols, nnz; returns 0 if absent. */
int orc_amg_level_shape(const struct orc_amg *h, int l, int what, int64_t *out)
{
    const level_t *L = &h->lv[l];
    const csr_t *M = what == 0 ? L->A : what == 1 ? L->P : L->R;
    if (!M) return 0;
    out[0] = M->nrows; out[1] = M->ncols; out[2] = M->ptr[M->nrows];
    return 1;
}

int orc_amg_level_copy(const struct orc_amg *h, int l, int what, idx_t *ptr, idx_t *col, double *val)
{
    const level_t *L = &h->lv[l];
    const csr_t *M = what == 0 ? L->A : what == 1 ? L->P : L->R;
    if (!M) return 0;
    memcpy(ptr, M->ptr, (size_t)(M->nrows + 1) * sizeof(idx_t));
    memcpy(col, M->col, (size_t)M->ptr[M->nrows] * sizeof(idx_t));
    memcpy(val, M->val, (size_t)M->ptr[M->nrows] * 8);
    return 1;
}

/* out[0..3] = chebyshev rho, d, c ; SA omega of level l */
void orc_amg_level_scalars(const struct orc_amg *h, int l, double *out)
{
    const level_t *L = &h->lv[l];
    out[0] = L->relax ? L->relax->rho : 0.0; out[1] = L->relax ? L->relax->d : 0.0; out[2] = L->relax ? L->relax->c : 0.0;
    out[3] = L->omega;
}

/* stand-alone pieces, exposed so the tests can pin them one by one */
int64_t orc_plain_aggregates(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, double eps_strong,
                             idx_t *id)
{
    csr_t A = {n, n, (idx_t *)rowptr, (idx_t *)col, (double *)val};
    char *strong = (char *)malloc((size_t)rowptr[n] + 1);
    int64_t c = plain_aggregates(&A, eps_strong, strong, id);
    free(strong);
    return c;
}

/* "amg.aggregation" = "parallel" on the strength graph of a scalar matrix; *rounds = synchronous rounds it took */
int64_t orc_compact_aggregates(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, double eps_strong,
                               idx_t *id, int *rounds)
{
    csr_t A = {n, n, (idx_t *)rowptr, (idx_t *)col, (double *)val};
    char *strong = (char *)malloc((size_t)rowptr[n] + 1);
    idx_t *tmp = (idx_t *)malloc((size_t)n * sizeof(idx_t) + 8);
    plain_aggregates_mode(&A, eps_strong, strong, tmp, 0); /* (the strength flags as plain_aggregates computes them) */
    free(tmp);
    int64_t c = compact_aggregates_graph(n, rowptr, col, strong, id, rounds);
    free(strong);
    return c;
}

int64_t orc_parallel_aggregates(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, double eps_strong,
                                idx_t *id, int *rounds)
{
    csr_t A = {n, n, (idx_t *)rowptr, (idx_t *)col, (double *)val};
    char *strong = (char *)malloc((size_t)rowptr[n] + 1);
    /* (the strength flags as plain_aggregates computes them) */
    idx_t *tmp = (idx_t *)malloc((size_t)n * sizeof(idx_t) + 8);
    plain_aggregates_mode(&A, eps_strong, strong, tmp, 0);
    free(tmp);
    int64_t c = parallel_aggregates_graph(n, rowptr, col, strong, id, rounds);
    free(strong);
    return c;
}

double orc_spectral_radius(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int scale,
                           int power_iters)
{
    csr_t A = {n, n, (idx_t *)rowptr, (idx_t *)col, (double *)val};
    return spectral_radius(&A, scale, power_iters);
}

/* x <- chebyshev smoothing of (A, rhs) starting from x; rho given (no estimation). */
void orc_chebyshev(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *rhs, double *x,
                   int degree, double rho, double higher, double lower)
{
    csr_t A = {n, n, (idx_t *)rowptr, (idx_t *)col, (double *)val};
    cheby_t *C = cheby_create(&A, degree, 0, higher, lower, 1);
    double hi = rho, lo = hi * lower;
    hi *= higher;
    C->d = 0.5 * (hi + lo);
    C->c = 0.5 * (hi - lo);
    relax_apply(C, &A, rhs, x, 0);
    cheby_free(C);
}

void orc_mt19937_uniform(uint32_t seed, int64_t n, double *out)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (int64_t i = 0; i < n; ++i) out[i] = mt_uniform_pm1(&g);
}
