/*
 * ic_oracle.c -- CPU restatement of Eigen::IncompleteCholesky<double, Lower, NaturalOrdering<int>>.
 * TEST INFRASTRUCTURE ONLY (see psolve_oracle.c): the product never links, imports or calls it.
 *
 * Reference call sites: the preconditioner name "Eigen::IncompleteCholesky" of the factory
 *     /root/reference/src/polysolve/linear/Solver.cpp:179-183  (ConjugateGradient<.., IncompleteCholesky<double>>)
 *     /root/reference/src/polysolve/linear/Solver.cpp:591-604  (available_preconds())
 * The arithmetic lives in the un-vendored Eigen 5.0.1 (cmake/recipes/eigen.cmake:26),
 * Eigen/src/IterativeLinearSolvers/IncompleteCholesky.h, restated here [upstream, recalled -- PARITY UNPINNED]:
 *   * the lower triangle of the matrix, columns as stored (diagonal first);
 *   * scaling  s_j = 1 / sqrt(|| column j of the symmetric matrix ||_2),  A <- S A S;
 *   * shift: 0 when the scaled diagonal is positive, else initial_shift (1e-3) - min diag; a failed attempt
 *     (non-positive pivot) restarts with shift = max(initial_shift, 2 shift), at most 10 attempts;
 *   * left-looking ("jki") factorization column by column; the columns that update column j are kept in a list per
 *     row index (listCol), each with the position of its next entry (firstElt);
 *   * dropping: column j keeps as many off-diagonal entries as the matrix column had (the p largest in magnitude,
 *     selected by the QuickSplit partition of IncompleteLUT.h), fill-in beyond that is dropped;
 *   * solve: z = S P^T L^-T L^-1 P S r.
 * Two details are restated as the mathematics has them, [upstream, recalled] without the source at hand: a fill-in
 * entry starts at -l_ik l_jk (older Eigen releases assigned +l_ik l_jk there, a reported sign error), and the
 * row -> slot map of the working column is cleared for every entry of the column, dropped ones included.
 * The reference's default ORDERING -- IncompleteCholesky<double> defaults to AMDOrdering<int> -- is restated separately
 * (amd_oracle.c, round 4): oracle.IC(A, ordering="amd") factors the explicitly permuted matrix with this file and permutes
 * right-hand sides in and solutions out, as IncompleteCholesky::_solve_impl does.  This file is the factorization in whatever
 * order its input comes.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t idx_t;

typedef struct {
    int64_t n, nnz;
    idx_t *colptr, *rowidx; /* L by columns, diagonal first, the rest in the order the factorization left them */
    double *vals;
    double *scale;
    double shift;
    int attempts, ok;
} orc_ic;

/* Eigen::internal::QuickSplit (IncompleteLUT.h): the ncut largest |row| first (not sorted) */
static void quick_split(double *row, idx_t *ind, int64_t n, int64_t ncut)
{
    int64_t first = 0, last = n - 1, mid;
    ncut--;
    if (ncut < first || ncut > last) return;
    do {
        mid = first;
        const double abskey = fabs(row[mid]);
        for (int64_t j = first + 1; j <= last; j++) {
            if (fabs(row[j]) > abskey) {
                ++mid;
                double tv = row[mid]; row[mid] = row[j]; row[j] = tv;
                idx_t ti = ind[mid]; ind[mid] = ind[j]; ind[j] = ti;
            }
        }
        { double tv = row[mid]; row[mid] = row[first]; row[first] = tv; }
        { idx_t ti = ind[mid]; ind[mid] = ind[first]; ind[first] = ti; }
        if (mid > ncut) last = mid - 1;
        else if (mid < ncut) first = mid + 1;
    } while (mid != ncut);
}

/* per-row lists of columns in insertion order (std::list<StorageIndex>::push_back); a column sits in one list at most */
typedef struct { idx_t *head, *tail, *next; } col_lists;

static void list_push(col_lists *L, idx_t row, idx_t col)
{
    L->next[col] = -1;
    if (L->head[row] < 0) L->head[row] = col;
    else L->next[L->tail[row]] = col;
    L->tail[row] = col;
}

/* IncompleteCholesky::updateList: the smallest remaining row index of column `col` moves to position jk; the column
 * enters the list of that row */
static void update_list(const idx_t *colptr, idx_t *rowidx, double *vals, idx_t col, int64_t jk, idx_t *first_elt,
                        col_lists *L)
{
    if (jk < colptr[col + 1]) {
        int64_t minpos = jk;
        for (int64_t q = jk + 1; q < colptr[col + 1]; ++q)
            if (rowidx[q] < rowidx[minpos]) minpos = q;
        if (rowidx[minpos] != rowidx[jk]) {
            idx_t ti = rowidx[jk]; rowidx[jk] = rowidx[minpos]; rowidx[minpos] = ti;
            double tv = vals[jk]; vals[jk] = vals[minpos]; vals[minpos] = tv;
        }
        first_elt[col] = (idx_t)jk;
        list_push(L, rowidx[jk], col);
    }
}

void orc_ic_destroy(void *h)
{
    orc_ic *I = (orc_ic *)h;
    if (!I) return;
    free(I->colptr); free(I->rowidx); free(I->vals); free(I->scale); free(I);
}

/* rowptr / col / val: the CSC (= CSR for a symmetric matrix) arrays with sorted inner indices; only the entries with
 * row >= column are read.  Returns NULL when a column has no stored diagonal. */
void *orc_ic_create(int64_t n, const idx_t *rowptr, const idx_t *col, const double *val, double initial_shift)
{
    orc_ic *I = (orc_ic *)calloc(1, sizeof(orc_ic));
    I->n = n;
    I->colptr = (idx_t *)malloc(((size_t)n + 1) * sizeof(idx_t));
    int64_t nnz = 0;
    for (int64_t j = 0; j < n; ++j) {
        I->colptr[j] = (idx_t)nnz;
        int has_diag = 0;
        for (idx_t k = rowptr[j]; k < rowptr[j + 1]; ++k) {
            if (col[k] == j) has_diag = 1;
            if (col[k] >= j) ++nnz;
        }
        if (!has_diag) { free(I->colptr); free(I); return NULL; }
    }
    I->colptr[n] = (idx_t)nnz;
    I->nnz = nnz;
    I->rowidx = (idx_t *)malloc((size_t)nnz * sizeof(idx_t));
    I->vals = (double *)malloc((size_t)nnz * sizeof(double));
    I->scale = (double *)calloc((size_t)n, sizeof(double));
    idx_t *rowidx = I->rowidx, *colptr = I->colptr;
    double *vals = I->vals, *scale = I->scale;
    for (int64_t j = 0, w = 0; j < n; ++j)
        for (idx_t k = rowptr[j]; k < rowptr[j + 1]; ++k)
            if (col[k] >= j) { rowidx[w] = col[k]; vals[w] = val[k]; ++w; }
    /* scaling factors */
    for (int64_t j = 0; j < n; ++j)
        for (idx_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            scale[j] += vals[k] * vals[k];
            if (rowidx[k] != j) scale[rowidx[k]] += vals[k] * vals[k];
        }
    for (int64_t j = 0; j < n; ++j) {
        const double s = sqrt(sqrt(scale[j]));
        scale[j] = s > DBL_MIN ? 1.0 / s : 1.0;
    }
    double mindiag = DBL_MAX;
    for (int64_t j = 0; j < n; ++j) {
        for (idx_t k = colptr[j]; k < colptr[j + 1]; ++k) vals[k] *= scale[j] * scale[rowidx[k]];
        if (vals[colptr[j]] < mindiag) mindiag = vals[colptr[j]];
    }
    idx_t *save_row = (idx_t *)malloc((size_t)nnz * sizeof(idx_t));
    double *save_val = (double *)malloc((size_t)nnz * sizeof(double));
    memcpy(save_row, rowidx, (size_t)nnz * sizeof(idx_t));
    memcpy(save_val, vals, (size_t)nnz * sizeof(double));
    double shift = 0.0;
    if (mindiag <= 0.0) shift = initial_shift - mindiag;
    idx_t *first_elt = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    col_lists L;
    L.head = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    L.tail = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    L.next = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    double *col_vals = (double *)malloc((size_t)n * sizeof(double));
    idx_t *col_irow = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    idx_t *col_pattern = (idx_t *)malloc((size_t)n * sizeof(idx_t));
    for (int64_t i = 0; i < n; ++i) { col_pattern[i] = -1; L.head[i] = L.tail[i] = L.next[i] = -1; }
    int iter = 0, success = 0, gave_up = 0;
    do {
        for (int64_t j = 0; j < n; ++j) vals[colptr[j]] += shift;
        int64_t j = 0;
        for (; j < n; ++j) {
            double diag = vals[colptr[j]];
            int64_t col_nnz = 0;
            for (idx_t i = colptr[j] + 1; i < colptr[j + 1]; i++) {
                const idx_t l = rowidx[i];
                col_vals[col_nnz] = vals[i];
                col_irow[col_nnz] = l;
                col_pattern[l] = (idx_t)col_nnz;
                col_nnz++;
            }
            /* all previous columns that update column j, in the order they entered its list */
            for (idx_t k = L.head[j]; k >= 0;) {
                const idx_t knext = L.next[k]; /* k moves to another list below */
                int64_t jk = first_elt[k];
                const double v_j_jk = vals[jk];
                jk += 1;
                for (int64_t i = jk; i < colptr[k + 1]; i++) {
                    const idx_t l = rowidx[i];
                    if (col_pattern[l] < 0) {
                        col_vals[col_nnz] = -vals[i] * v_j_jk; /* fill-in: 0 - l_ik l_jk */
                        col_irow[col_nnz] = l;
                        col_pattern[l] = (idx_t)col_nnz;
                        col_nnz++;
                    } else {
                        col_vals[col_pattern[l]] -= vals[i] * v_j_jk;
                    }
                }
                update_list(colptr, rowidx, vals, k, jk, first_elt, &L);
                k = knext;
            }
            L.head[j] = L.tail[j] = -1;
            if (diag <= 0.0) {
                if (++iter >= 10) { gave_up = 1; break; }
                shift = initial_shift > 2.0 * shift ? initial_shift : 2.0 * shift;
                memcpy(rowidx, save_row, (size_t)nnz * sizeof(idx_t));
                memcpy(vals, save_val, (size_t)nnz * sizeof(double));
                for (int64_t i = 0; i < n; ++i) { col_pattern[i] = -1; L.head[i] = L.tail[i] = L.next[i] = -1; }
                break;
            }
            const double rdiag = sqrt(diag);
            vals[colptr[j]] = rdiag;
            for (int64_t k = 0; k < col_nnz; ++k) {
                const idx_t i = col_irow[k];
                col_vals[k] /= rdiag;
                vals[colptr[i]] -= col_vals[k] * col_vals[k];
            }
            const int64_t p = colptr[j + 1] - colptr[j] - 1;
            quick_split(col_vals, col_irow, col_nnz, p);
            int64_t cpt = 0;
            for (idx_t i = colptr[j] + 1; i < colptr[j + 1]; i++) {
                vals[i] = col_vals[cpt];
                rowidx[i] = col_irow[cpt];
                cpt++;
            }
            for (int64_t k = 0; k < col_nnz; ++k) col_pattern[col_irow[k]] = -1; /* dropped entries included */
            update_list(colptr, rowidx, vals, (idx_t)j, colptr[j] + 1, first_elt, &L);
        }
        if (j == n) success = 1;
    } while (!success && !gave_up);
    I->shift = shift;
    I->attempts = iter + 1;
    I->ok = success;
    free(save_row); free(save_val); free(first_elt); free(L.head); free(L.tail); free(L.next);
    free(col_vals); free(col_irow); free(col_pattern);
    return I;
}

void orc_ic_info(void *h, double *shift, int64_t *nnz, int *attempts, int *ok)
{
    orc_ic *I = (orc_ic *)h;
    if (shift) *shift = I->shift;
    if (nnz) *nnz = I->nnz;
    if (attempts) *attempts = I->attempts;
    if (ok) *ok = I->ok;
}

void orc_ic_copy(void *h, idx_t *colptr, idx_t *rowidx, double *vals, double *scale)
{
    orc_ic *I = (orc_ic *)h;
    memcpy(colptr, I->colptr, ((size_t)I->n + 1) * sizeof(idx_t));
    memcpy(rowidx, I->rowidx, (size_t)I->nnz * sizeof(idx_t));
    memcpy(vals, I->vals, (size_t)I->nnz * sizeof(double));
    memcpy(scale, I->scale, (size_t)I->n * sizeof(double));
}

/* z = S L^-T L^-1 S r  (IncompleteCholesky::_solve_impl with the identity permutation); a failed factorization acts
 * as the identity */
void orc_ic_apply(void *h, const double *r, double *z)
{
    orc_ic *I = (orc_ic *)h;
    const int64_t n = I->n;
    if (!I->ok) { memcpy(z, r, (size_t)n * sizeof(double)); return; }
    const idx_t *colptr = I->colptr, *rowidx = I->rowidx;
    const double *vals = I->vals, *scale = I->scale;
    for (int64_t i = 0; i < n; ++i) z[i] = scale[i] * r[i];
    for (int64_t i = 0; i < n; ++i) { /* column-major lower solve */
        double tmp = z[i];
        if (tmp != 0.0) {
            tmp = z[i] = tmp / vals[colptr[i]];
            for (idx_t k = colptr[i] + 1; k < colptr[i + 1]; ++k) z[rowidx[k]] -= tmp * vals[k];
        }
    }
    for (int64_t i = n - 1; i >= 0; --i) { /* row-major upper solve with the adjoint */
        double tmp = z[i];
        for (idx_t k = colptr[i] + 1; k < colptr[i + 1]; ++k) tmp -= vals[k] * z[rowidx[k]];
        z[i] = tmp / vals[colptr[i]];
    }
    for (int64_t i = 0; i < n; ++i) z[i] *= scale[i];
}
