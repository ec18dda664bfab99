/*
 * psolve_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (polysolve_amd/, libpsolve_hip.so) never links, imports or
 * calls it.
 *
 * PARITY UNPINNED: the arithmetic of the reference path lives in two un-vendored third-party
 * dependencies that are absent from /root/reference and from this image:
 *     Eigen  5.0.1  (cmake/recipes/eigen.cmake:26)  -- Eigen::ConjugateGradient,
 *                                                       Eigen::DiagonalPreconditioner
 *     AMGCL  1.4.3  (cmake/recipes/amgcl.cmake:47)  -- amgcl::solver::cg, amgcl::amg,
 *                    coarsening::smoothed_aggregation, relaxation::chebyshev
 * and the reference's own tests hold no golden vectors for it (tolerance-only assertions on
 * fixtures that are downloaded at configure time).  This file restates the published
 * algorithms of those two libraries; it is anchored on the reference's call sites
 *     src/polysolve/linear/Solver.cpp:433-436      (ConjugateGradient<.., Lower|Upper, DiagonalPreconditioner>)
 *     src/polysolve/linear/EigenSolver.tpp:68-114  (setTolerance/setMaxIterations/factorize/solveWithGuess)
 *     src/polysolve/linear/AMGCL.cpp:32-65         (default_params: cg + SA-AMG + chebyshev)
 *     src/polysolve/linear/AMGCL.cpp:148-212       (factorize / solve)
 * and on the reference tests' inequalities (tests/test_linear_solver.cpp:103-164, 241-307,
 * 400-455, 541-665), and is cross-checked against scipy (tests/test_oracle.py).
 *
 * Plain C99 + optional OpenMP.  CSR: int32 row pointers / column ids, fp64 values.  A symmetric
 * matrix handed over as Eigen ColMajor (CSC) arrays is the same bytes (AMGCL.hpp:36-43).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int32_t idx_t;

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic inputs: 7-point Poisson, SplitMix64 vectors                       */
/* ------------------------------------------------------------------------------------------ */

/* nnz of the rows with z in [z0, z1) of the nx*ny*nz 7-point Laplacian (Dirichlet truncation). */
int64_t orc_poisson7_nnz(int nx, int ny, int nz, int z0, int z1)
{
    int64_t total = 0;
    for (int k = z0; k < z1; ++k) {
        for (int j = 0; j < ny; ++j) {
            int per_row_base = 1 + (k > 0) + (k < nz - 1) + (j > 0) + (j < ny - 1);
            /* i-neighbours: interior rows have 2, the two ends have 1 (or 0 when nx == 1) */
            int64_t irow = (int64_t)nx * per_row_base + (nx > 1 ? 2 * (int64_t)(nx - 1) : 0);
            total += irow;
        }
    }
    return total;
}

/* Rows z0..z1 of the matrix: diagonal 6, -1 to each in-grid neighbour, sorted columns, GLOBAL
 * column ids; rowptr is local (rowptr[0] = 0) with (z1-z0)*nx*ny + 1 entries. */
void orc_poisson7_fill(int nx, int ny, int nz, int z0, int z1, idx_t *rowptr, idx_t *col, double *val)
{
    const int64_t plane = (int64_t)nx * ny;
    rowptr[0] = 0;
    /* planes are independent once their offsets are known: filled in parallel so that, on a NUMA
     * host, the pages are first touched by the threads that will stream them in orc_spmv */
#pragma omp parallel for schedule(static)
    for (int k = z0; k < z1; ++k) {
        int64_t p = orc_poisson7_nnz(nx, ny, nz, z0, k);
        int64_t lr = (int64_t)(k - z0) * plane;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                int64_t r = i + (int64_t)nx * (j + (int64_t)ny * k);
                if (k > 0) { col[p] = (idx_t)(r - plane); val[p++] = -1.0; }
                if (j > 0) { col[p] = (idx_t)(r - nx); val[p++] = -1.0; }
                if (i > 0) { col[p] = (idx_t)(r - 1); val[p++] = -1.0; }
                col[p] = (idx_t)r; val[p++] = 6.0;
                if (i < nx - 1) { col[p] = (idx_t)(r + 1); val[p++] = -1.0; }
                if (j < ny - 1) { col[p] = (idx_t)(r + nx); val[p++] = -1.0; }
                if (k < nz - 1) { col[p] = (idx_t)(r + plane); val[p++] = -1.0; }
                rowptr[++lr] = (idx_t)p;
            }
    }
}

static inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* x[i] = U(-1,1) from SplitMix64(seed + start + i): stateless per index, identical on any shard. */
/* STREAM-like triad a = b + s c over three arrays of n doubles, allocated here and first-touched by the threads that
 * stream them (schedule(static), as every loop of this file): best of `reps` passes, in GB/s (24 n bytes per pass).
 * bench.py prints it next to the port's own rate, so that a reader sees how far the CPU baseline is from what the socket
 * streams.  Test infrastructure like the rest of oracle/. */
double orc_stream_triad(int64_t n, int reps)
{
    double *a = (double *)malloc((size_t)n * 8), *b = (double *)malloc((size_t)n * 8), *c = (double *)malloc((size_t)n * 8);
    if (!a || !b || !c) { free(a); free(b); free(c); return 0.0; }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    double best = 0.0;
    for (int r = 0; r < reps; ++r) {
        const double s = 3.0 + r;
        const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) a[i] = b[i] + s * c[i];
        const double dt = omp_get_wtime() - t0;
        const double gbs = 24.0 * (double)n / dt / 1e9;
        if (gbs > best) best = gbs;
    }
    volatile double sink = a[n / 2];
    (void)sink;
    free(a); free(b); free(c);
    return best;
}

void orc_splitmix_fill(double *x, int64_t start, int64_t n, uint64_t seed)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t u = splitmix64(seed + (uint64_t)(start + i));
        x[i] = (double)(u >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* BLAS-1 / SpMV                                                                                */
/* ------------------------------------------------------------------------------------------ */

void orc_spmv(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (idx_t j = rowptr[i]; j < rowptr[i + 1]; ++j) s += val[j] * x[col[j]];
        y[i] = s;
    }
}

/* r = b - A x */
void orc_residual(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *b,
                  const double *x, double *r)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = b[i];
        for (idx_t j = rowptr[i]; j < rowptr[i + 1]; ++j) s -= val[j] * x[col[j]];
        r[i] = s;
    }
}

/* Deterministic for ANY thread count: fixed 32768-element chunks, each summed left to right, then
 * the chunk partials summed left to right. */
#define ORC_DOT_CHUNK 32768
double orc_dot(int64_t n, const double *a, const double *b)
{
    const int64_t nchunks = (n + ORC_DOT_CHUNK - 1) / ORC_DOT_CHUNK;
    if (nchunks <= 1) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
        return s;
    }
    double *part = (double *)malloc((size_t)nchunks * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t lo = c * ORC_DOT_CHUNK, hi = (lo + ORC_DOT_CHUNK < n) ? lo + ORC_DOT_CHUNK : n;
        double s = 0.0;
        for (int64_t i = lo; i < hi; ++i) s += a[i] * b[i];
        part[c] = s;
    }
    double s = 0.0;
    for (int64_t c = 0; c < nchunks; ++c) s += part[c];
    free(part);
    return s;
}

/* Eigen::DiagonalPreconditioner::factorize: invdiag[j] = A(j,j) != 0 ? 1/A(j,j) : 1
 * (duplicated diagonal entries are summed, as Eigen's InnerIterator loop does). */
void orc_jacobi_setup(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, double *invdiag)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double d = 0.0;
        for (idx_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
            if (col[j] == i) d += val[j];
        invdiag[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Preconditioner dispatch used by both CG restatements                                         */
/* ------------------------------------------------------------------------------------------ */

struct orc_amg;
void orc_amg_apply(struct orc_amg *h, const double *rhs, double *x);
struct orc_schwarz;
void orc_schwarz_apply(struct orc_schwarz *S, const double *r, double *z);
void orc_ic_apply(void *h, const double *r, double *z); /* ic_oracle.c */

typedef struct {
    int kind;               /* 0 identity, 1 jacobi (invdiag), 2 amg, 3 schwarz, 4 incomplete Cholesky (handles passed in `amg`) */
    const double *invdiag;  /* kind 1 */
    struct orc_amg *amg;    /* kind 2 */
} orc_precond;

static void precond_apply(const orc_precond *P, int64_t n, const double *r, double *z)
{
    if (P->kind == 1) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) z[i] = P->invdiag[i] * r[i];
    } else if (P->kind == 2) {
        orc_amg_apply(P->amg, r, z);
    } else if (P->kind == 3) {
        orc_schwarz_apply((struct orc_schwarz *)P->amg, r, z);
    } else if (P->kind == 4) {
        orc_ic_apply((void *)P->amg, r, z);
    } else {
        memcpy(z, r, (size_t)n * sizeof(double));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Eigen::internal::conjugate_gradient   [Eigen 5.0.1, IterativeLinearSolvers/ConjugateGradient.h] */
/* reached from EigenSolver.tpp:109-114 (x = m_Solver.solveWithGuess(b, x)).                    */
/* ------------------------------------------------------------------------------------------ */
/*
 * precond_kind: 0 = IdentityPreconditioner, 1 = DiagonalPreconditioner (invdiag given), 2 = AMG.
 * x is the initial guess on entry.  hist (optional, max_iter+1 doubles) receives ||r||^2 after
 * each in-loop update (hist[0] = initial).  Returns via *iters / *err what
 * ConjugateGradient::iterations() / error() report (EigenSolver.tpp:88-89).
 * Stopping rule: RECURRENCE residual, relative to ||b|| (not ||r0||); `break` precedes `i++`.
 */
void orc_cg_eigen(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *b, double *x,
                  int precond_kind, const double *invdiag, struct orc_amg *amg, double tol, int64_t max_iter,
                  int64_t *iters, double *err, double *hist)
{
    orc_precond P = {precond_kind, invdiag, amg};
    double *r = (double *)malloc((size_t)n * 8), *p = (double *)malloc((size_t)n * 8);
    double *z = (double *)malloc((size_t)n * 8), *tmp = (double *)malloc((size_t)n * 8);

    orc_residual(n, rowptr, col, val, b, x, r);
    double rhsNorm2 = orc_dot(n, b, b);
    if (rhsNorm2 == 0.0) {
        memset(x, 0, (size_t)n * 8);
        *iters = 0;
        *err = 0.0;
        goto done;
    }
    {
        double threshold = tol * tol * rhsNorm2;
        if (threshold < DBL_MIN) threshold = DBL_MIN;
        double rn2 = orc_dot(n, r, r);
        if (hist) hist[0] = rn2;
        if (rn2 < threshold) {
            *iters = 0;
            *err = sqrt(rn2 / rhsNorm2);
            goto done;
        }
        precond_apply(&P, n, r, p);
        double absNew = orc_dot(n, r, p);
        int64_t i = 0;
        while (i < max_iter) {
            orc_spmv(n, rowptr, col, val, p, tmp);
            double alpha = absNew / orc_dot(n, p, tmp);
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < n; ++k) {
                x[k] += alpha * p[k];
                r[k] -= alpha * tmp[k];
            }
            rn2 = orc_dot(n, r, r);
            if (hist) hist[i + 1] = rn2;
            if (rn2 < threshold) break;
            precond_apply(&P, n, r, z);
            double absOld = absNew;
            absNew = orc_dot(n, r, z);
            double beta = absNew / absOld;
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < n; ++k) p[k] = z[k] + beta * p[k];
            i++;
        }
        *err = sqrt(rn2 / rhsNorm2);
        *iters = i;
    }
done:
    free(r); free(p); free(z); free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* amgcl::solver::cg::operator()   [AMGCL 1.4.3, amgcl/solver/cg.hpp]                           */
/* reached from AMGCL.cpp:209 ((*solver_)(rhs, x)); tol/maxiter from AMGCL.cpp:57-61.           */
/* ------------------------------------------------------------------------------------------ */
/* Returns *iters and *err = ||r||/||b|| exactly as the (iterations_, residual_error_) tuple of
 * AMGCL.cpp:209 / get_info AMGCL.cpp:142-143.  abstol <= 0 selects AMGCL's default (DBL_MIN). */
void orc_cg_amgcl(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *b, double *x,
                  int precond_kind, const double *invdiag, struct orc_amg *amg, double tol, double abstol,
                  int64_t max_iter, int64_t *iters, double *err)
{
    orc_precond P = {precond_kind, invdiag, amg};
    double norm_rhs = sqrt(orc_dot(n, b, b));
    if (norm_rhs < DBL_EPSILON) { /* amgcl::detail::eps<double>(1) */
        memset(x, 0, (size_t)n * 8);
        *iters = 0;
        *err = norm_rhs;
        return;
    }
    if (abstol <= 0.0) abstol = DBL_MIN;
    double eps = tol * norm_rhs;
    if (eps < abstol) eps = abstol;

    double *r = (double *)malloc((size_t)n * 8), *s = (double *)malloc((size_t)n * 8);
    double *p = (double *)malloc((size_t)n * 8), *q = (double *)malloc((size_t)n * 8);
    double rho1 = 2 * eps, rho2 = 0.0;
    orc_residual(n, rowptr, col, val, b, x, r);
    double res = sqrt(orc_dot(n, r, r));
    int64_t iter = 0;
    for (; iter < max_iter && res > eps; ++iter) {
        precond_apply(&P, n, r, s);
        rho2 = rho1;
        rho1 = orc_dot(n, r, s);
        if (iter) {
            double beta = rho1 / rho2;
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < n; ++k) p[k] = s[k] + beta * p[k];
        } else {
            memcpy(p, s, (size_t)n * 8);
        }
        orc_spmv(n, rowptr, col, val, p, q);
        double alpha = rho1 / orc_dot(n, q, p);
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < n; ++k) {
            x[k] += alpha * p[k];
            r[k] -= alpha * q[k];
        }
        res = sqrt(orc_dot(n, r, r));
    }
    *iters = iter;
    *err = res / norm_rhs;
    free(r); free(s); free(p); free(q);
}
