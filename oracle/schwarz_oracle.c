/*
 * schwarz_oracle.c -- CPU restatement of the multilevel additive Schwarz preconditioner of
 * polysolve_amd/csrc/schwarz.hip (precond = "schwarz").  TEST INFRASTRUCTURE ONLY.
 *
 * What it restates is THIS repository's wave64 re-think of the reference's MAS preconditioner
 * (/root/reference/src/polysolve/linear/mas_utils/MASPreconditioner.cu:58-457), not MAS itself: domains of 64
 * consecutive unknowns, level l+1 = one unknown per level-l domain (index >> 6), B_l = the 64 x 64 diagonal blocks
 * of P_l^T A P_l, z = sum_l P_l B_l^-1 P_l^T r.  The reference has no golden vectors for MAS either; parity here is
 * device kernel vs this file (same order of additions in the block sums, same Gauss-Jordan elimination).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t idx_t;
#define DOM 64

struct orc_schwarz {
    int levels, bs;
    int64_t n[8], nblk[8];
    double *inv[8];  /* nblk x 64 x 64 */
    double *r[8], *z[8];
};

static void invert_block(double *M, int64_t first, int64_t n_l)
{
    /* rows/columns of unknowns beyond n_l are identity; Gauss-Jordan, no pivoting (SPD) */
    for (int i = 0; i < DOM; ++i)
        for (int j = 0; j < DOM; ++j)
            if (first + i >= n_l || first + j >= n_l) M[i * DOM + j] = (i == j) ? 1.0 : 0.0;
    /* a pivot that is zero or negligible against the largest diagonal entry (semi-definite block of a floating
     * sub-domain, empty row) takes its unknown out of the domain solve: row and column k become the identity's */
    double dmax = 0.0;
    for (int i = 0; i < DOM; ++i) dmax = fmax(dmax, fabs(M[i * DOM + i]));
    const double tiny = 64.0 * 2.220446049250313e-16 * dmax;
    for (int k = 0; k < DOM; ++k) {
        double piv = M[k * DOM + k];
        if (!(fabs(piv) > tiny)) {
            for (int j = 0; j < DOM; ++j) M[k * DOM + j] = M[j * DOM + k] = 0.0;
            M[k * DOM + k] = 1.0;
            piv = 1.0;
        }
        const double ip = 1.0 / piv;
        double rowk[DOM];
        for (int j = 0; j < DOM; ++j) rowk[j] = (j == k) ? ip : M[k * DOM + j] * ip;
        for (int j = 0; j < DOM; ++j) M[k * DOM + j] = rowk[j];
        for (int i = 0; i < DOM; ++i) {
            if (i == k) continue;
            const double f = M[i * DOM + k];
            for (int j = 0; j < DOM; ++j) M[i * DOM + j] = (j == k) ? -f * ip : M[i * DOM + j] - f * rowk[j];
        }
    }
}

/* level-l unknown of fine unknown i = node * bs + c: (node >> 6 l) * bs + c (components kept apart) */
static int64_t coarse_of(int64_t i, int shift, int bs)
{
    const int64_t node = i / bs;
    return (node >> shift) * bs + (i - node * bs);
}

struct orc_schwarz *orc_schwarz_create_bs(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int levels,
                                          int block_size)
{
    struct orc_schwarz *S = (struct orc_schwarz *)calloc(1, sizeof(*S));
    const int bs = (block_size > 1 && n % block_size == 0) ? block_size : 1;
    const int64_t nnodes = n / bs;
    S->bs = bs;
    int64_t n_l = n;
    for (int l = 0; l < levels && l < 8; ++l) {
        const int shift = 6 * l;
        const int64_t nblk = (n_l + DOM - 1) / DOM;
        S->n[l] = n_l;
        S->nblk[l] = nblk;
        S->inv[l] = (double *)calloc((size_t)nblk * DOM * DOM, sizeof(double));
        S->r[l] = (double *)calloc((size_t)n_l + 1, sizeof(double));
        S->z[l] = (double *)calloc((size_t)n_l + 1, sizeof(double));
        /* row I = (g, c) of its domain block: its fine rows (nodes of group g, component c) in order, entries in
         * storage order */
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t I = 0; I < n_l; ++I) {
            const int64_t g = I / bs, c = I - g * bs;
            int64_t node1 = (g + 1) << shift;
            if (node1 > nnodes) node1 = nnodes;
            double *row = S->inv[l] + (I >> 6) * (DOM * DOM) + (I & 63) * DOM;
            for (int64_t node = g << shift; node < node1; ++node) {
                const int64_t r = node * bs + c;
                for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                    const int64_t cc = col[k];
                    if (cc < 0 || cc >= n) continue;
                    const int64_t cl = coarse_of(cc, shift, bs);
                    if ((cl >> 6) == (I >> 6)) row[cl & 63] += val[k];
                }
            }
        }
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t b = 0; b < nblk; ++b) invert_block(S->inv[l] + b * (DOM * DOM), b * DOM, n_l);
        S->levels = l + 1;
        if (n_l <= DOM) break;
        n_l = ((n_l / bs + DOM - 1) / DOM) * bs;
    }
    return S;
}

struct orc_schwarz *orc_schwarz_create(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, int levels)
{
    return orc_schwarz_create_bs(n, rowptr, col, val, levels, 1);
}

void orc_schwarz_destroy(struct orc_schwarz *S)
{
    if (!S) return;
    for (int l = 0; l < 8; ++l) {
        free(S->inv[l]);
        free(S->r[l]);
        free(S->z[l]);
    }
    free(S);
}

int orc_schwarz_levels(const struct orc_schwarz *S) { return S->levels; }

/* the device sums 64 values by a butterfly: (lane ^ 32), 16, 8, 4, 2, 1 -- the same pairing here */
static double butterfly64(const double *v, int64_t avail)
{
    double t[DOM];
    for (int i = 0; i < DOM; ++i) t[i] = i < avail ? v[i] : 0.0;
    for (int off = 32; off > 0; off >>= 1) {
        double u[DOM];
        for (int i = 0; i < DOM; ++i) u[i] = t[i] + t[i ^ off];
        memcpy(t, u, sizeof(t));
    }
    return t[0];
}

void orc_schwarz_apply(struct orc_schwarz *S, const double *r, double *z)
{
    const int nl = S->levels;
    for (int l = 1; l < nl; ++l) {
        const double *rf = l == 1 ? r : S->r[l - 1];
        const int64_t nf = S->n[l - 1];
#pragma omp parallel for schedule(static)
        for (int64_t I = 0; I < S->n[l]; ++I) {
            const int64_t g = I / S->bs, c = I - g * S->bs;
            double v[DOM];
            for (int k = 0; k < DOM; ++k) {
                const int64_t i = (g * DOM + k) * S->bs + c;
                v[k] = i < nf ? rf[i] : 0.0;
            }
            S->r[l][I] = butterfly64(v, DOM);
        }
    }
    for (int l = nl - 1; l >= 0; --l) {
        const double *rl = l == 0 ? r : S->r[l];
        double *zl = l == 0 ? z : S->z[l];
        const double *zc = l + 1 < nl ? S->z[l + 1] : NULL;
        const int64_t n_l = S->n[l];
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < S->nblk[l]; ++b) {
            const double *B = S->inv[l] + b * (DOM * DOM);
            for (int i = 0; i < DOM; ++i) {
                if (b * DOM + i >= n_l) break;
                double acc = 0.0;
                for (int j = 0; j < DOM; ++j) {
                    const double rj = b * DOM + j < n_l ? rl[b * DOM + j] : 0.0;
                    acc += B[j * DOM + i] * rj;
                }
                if (zc) acc += zc[coarse_of(b * DOM + i, 6, S->bs)];
                zl[b * DOM + i] = acc;
            }
        }
    }
}
