/*
 * cpu_tuned.c -- the CPU baseline's TUNED leg.  TEST / BENCH INFRASTRUCTURE ONLY (see psolve_oracle.c header): only
 * bench.py's cpu_baseline child and tests/ load it; the product never does.
 *
 * orc_cg_jacobi_tuned runs the same Jacobi-preconditioned CG recurrence as orc_cg_eigen (psolve_oracle.c: the
 * restatement of Eigen::ConjugateGradient + DiagonalPreconditioner, EigenSolver.tpp:109-114 / Solver.cpp:433-436) -- same
 * threshold, same stopping rule on the recurrence residual relative to ||b||, break before the count -- the way a tuned
 * CPU code would run it rather than the way Eigen's expression templates do:
 *   - ONE parallel region for the whole solve; every thread owns a fixed, nnz-balanced row range;
 *   - the matrix, the right-hand side and the five vectors are private copies FIRST-TOUCHED by the thread that streams
 *     them (the caller's numpy arrays may sit on one NUMA node);
 *   - three fused passes per iteration (12 nnz + 100 n bytes instead of Eigen's 12 nnz + 156 n):
 *       q = A p, p.q | r -= alpha q, r.r, r.(D^-1 r) | x += alpha p, p = D^-1 r + beta p
 *   - reductions by per-thread partials (padded to a cache line) summed in thread order: deterministic for a given
 *     thread count.
 * *loop_seconds (optional) receives the wall time of the iteration loop alone (thread 0): the private copies are what a
 * tuned code does once per factorize, not once per solve, and bench.py's bounded sample would otherwise be dominated by them.
 * Sums are in a different order than orc_cg_eigen's, so iteration counts may differ by one and x in the last digits.
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

#define PAD 8 /* doubles per partial slot: one cache line */

static double sum_partials(const double *part, int nt)
{
    double s = 0.0;
    for (int t = 0; t < nt; ++t) s += part[(size_t)t * PAD];
    return s;
}

void orc_cg_jacobi_tuned(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, const double *b, double *x,
                         double tol, int64_t max_iter, int64_t *iters, double *err, double *loop_seconds)
{
#ifdef _OPENMP
    const int nt = omp_get_max_threads();
#else
    const int nt = 1;
#endif
    const int64_t nnz = rowptr[n];
    int64_t *lo = (int64_t *)malloc((size_t)(nt + 1) * sizeof(int64_t));
    /* nnz-balanced row ranges: thread t owns rows [lo[t], lo[t+1]) */
    lo[0] = 0;
    for (int t = 1; t < nt; ++t) {
        const int64_t target = nnz / nt * t;
        int64_t a = lo[t - 1], c = n;
        while (a < c) {
            int64_t m = (a + c) / 2;
            if (rowptr[m] < target) a = m + 1; else c = m;
        }
        lo[t] = a;
    }
    lo[nt] = n;

    idx_t *rp = (idx_t *)malloc((size_t)(n + 1) * sizeof(idx_t)), *cj = (idx_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(idx_t));
    double *av = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * 8);
    double *bb = (double *)malloc((size_t)n * 8), *xx = (double *)malloc((size_t)n * 8), *r = (double *)malloc((size_t)n * 8);
    double *p = (double *)malloc((size_t)n * 8), *q = (double *)malloc((size_t)n * 8), *dinv = (double *)malloc((size_t)n * 8);
    double *part = (double *)calloc((size_t)nt * PAD * 3, sizeof(double));
    double *pa = part, *pb = part + (size_t)nt * PAD, *pc = part + (size_t)nt * PAD * 2;

    /* shared scalars of the recurrence (written by one thread between barriers) */
    double rhsNorm2 = 0.0, threshold = 0.0, rn2 = 0.0, absNew = 0.0, alpha = 0.0, beta = 0.0;
    int64_t it = 0;
    int stop = 0; /* 1: converged (break before the count), 2: trivial exit */
    double t_loop = 0.0; /* wall time of the iteration loop alone: the private copies are a per-factorize cost */

#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const int64_t r0 = lo[t], r1 = lo[t + 1];
        /* first touch: private copies of everything this thread streams */
        for (int64_t i = r0; i < r1; ++i) {
            rp[i] = rowptr[i];
            double d = 0.0;
            for (idx_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
                cj[j] = col[j];
                av[j] = val[j];
                if (col[j] == i) d += val[j];
            }
            dinv[i] = (d != 0.0) ? 1.0 / d : 1.0; /* Eigen::DiagonalPreconditioner::factorize */
            bb[i] = b[i];
            xx[i] = x[i];
            p[i] = 0.0;
            q[i] = 0.0;
            r[i] = 0.0;
        }
        if (t == nt - 1) rp[n] = rowptr[n];
#pragma omp barrier
        /* r = b - A x0, ||b||^2, ||r||^2 */
        {
            double sb = 0.0, sr = 0.0;
            for (int64_t i = r0; i < r1; ++i) {
                double s = bb[i];
                for (idx_t j = rp[i]; j < rp[i + 1]; ++j) s -= av[j] * xx[cj[j]];
                r[i] = s;
                sb += bb[i] * bb[i];
                sr += s * s;
            }
            pa[(size_t)t * PAD] = sb;
            pb[(size_t)t * PAD] = sr;
        }
#pragma omp barrier
#pragma omp single
        {
            rhsNorm2 = sum_partials(pa, nt);
            rn2 = sum_partials(pb, nt);
            threshold = tol * tol * rhsNorm2;
            if (threshold < DBL_MIN) threshold = DBL_MIN;
            if (rhsNorm2 == 0.0 || rn2 < threshold) stop = 2;
        } /* (implicit barrier) */
        if (stop == 0) {
            /* p = D^-1 r, absNew = r.p */
            double s = 0.0;
            for (int64_t i = r0; i < r1; ++i) {
                p[i] = dinv[i] * r[i];
                s += r[i] * p[i];
            }
            pa[(size_t)t * PAD] = s;
#pragma omp barrier
#pragma omp single
            absNew = sum_partials(pa, nt);
#ifdef _OPENMP
            const double t_begin = omp_get_wtime();
#endif
            while (1) {
                if (it >= max_iter) break; /* (`it` only changes inside the single below, behind barriers) */
                /* pass 1: q = A p, p.q */
                double spq = 0.0;
                for (int64_t i = r0; i < r1; ++i) {
                    double acc = 0.0;
                    for (idx_t j = rp[i]; j < rp[i + 1]; ++j) acc += av[j] * p[cj[j]];
                    q[i] = acc;
                    spq += p[i] * acc;
                }
                pa[(size_t)t * PAD] = spq;
#pragma omp barrier
                const double al = absNew / sum_partials(pa, nt); /* every thread sums the same partials in the same order */
                /* pass 2: r -= alpha q, r.r, r.z */
                double srr = 0.0, srz = 0.0;
#pragma omp simd reduction(+ : srr, srz)
                for (int64_t i = r0; i < r1; ++i) {
                    const double ri = r[i] - al * q[i];
                    r[i] = ri;
                    srr += ri * ri;
                    srz += ri * (dinv[i] * ri);
                }
                pb[(size_t)t * PAD] = srr;
                pc[(size_t)t * PAD] = srz;
#pragma omp barrier
#pragma omp single
                {
                    alpha = al;
                    rn2 = sum_partials(pb, nt);
                    if (rn2 < threshold) {
                        stop = 1;
                    } else {
                        const double absOld = absNew;
                        absNew = sum_partials(pc, nt);
                        beta = absNew / absOld;
                        it++;
                    }
                } /* (implicit barrier) */
                if (stop) {
                    for (int64_t i = r0; i < r1; ++i) xx[i] += alpha * p[i];
                    break;
                }
                /* pass 3: x += alpha p, p = D^-1 r + beta p */
#pragma omp simd
                for (int64_t i = r0; i < r1; ++i) {
                    const double pi = p[i];
                    xx[i] += alpha * pi;
                    p[i] = dinv[i] * r[i] + beta * pi;
                }
#pragma omp barrier
            }
#ifdef _OPENMP
            if (t == 0) t_loop = omp_get_wtime() - t_begin;
#endif
        }
#pragma omp barrier
        if (stop == 2 && rhsNorm2 == 0.0)
            for (int64_t i = r0; i < r1; ++i) xx[i] = 0.0;
        for (int64_t i = r0; i < r1; ++i) x[i] = xx[i];
    }
    *iters = it;
    if (loop_seconds) *loop_seconds = t_loop;
    *err = (rhsNorm2 == 0.0) ? 0.0 : sqrt(rn2 / rhsNorm2);
    free(lo); free(rp); free(cj); free(av); free(bb); free(xx); free(r); free(p); free(q); free(dinv); free(part);
}
