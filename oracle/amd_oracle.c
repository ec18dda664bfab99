/*
 * amd_oracle.c -- CPU restatement of Eigen::AMDOrdering<int> (approximate minimum degree).
 * TEST INFRASTRUCTURE ONLY (see psolve_oracle.c): the product never links, imports or calls it.
 *
 * Reference call site: IncompleteCholesky<double> -- the preconditioner behind the factory's name
 * "Eigen::IncompleteCholesky" (/root/reference/src/polysolve/linear/Solver.cpp:179-183) -- defaults its ordering to
 * AMDOrdering<int>: analyzePattern() calls ord(mat.selfadjointView<UpLo>(), pinv) and keeps perm = pinv.inverse(),
 * factorize() factors the matrix twistedBy(perm), solve() permutes the right-hand side in and the solution out
 * (Eigen 5.0.1, Eigen/src/IterativeLinearSolvers/IncompleteCholesky.h).  The ordering itself is
 * Eigen/src/OrderingMethods/Amd.h: internal::minimum_degree_ordering, which its header describes as adapted from T.
 * Davis' CSparse (cs_amd, "Direct Methods for Sparse Linear Systems", SIAM 2006, ch. 7), the published Amestoy-Davis-Duff
 * approximate minimum degree algorithm with aggressive absorption, mass elimination, supernode detection by hashing and
 * a postordering of the assembly tree.  None of it is in /root/reference or in the image: restated here from the published
 * algorithm [upstream, recalled -- PARITY UNPINNED], with the two things Eigen's adaptation changes, as recalled:
 *   * the matrix keeps its diagonal entries (CSparse drops them first): a node whose only entry is its diagonal is
 *     eliminated at once, a node without a diagonal entry is treated like a dense one (absorbed into the dummy
 *     element n), and the initial degrees count the diagonal;
 *   * dense rows: more than max(16, 10 sqrt(n)) entries (capped at n - 2).
 * Result: order[k] = the k-th pivot (Eigen's `pinv.indices()`; IncompleteCholesky's m_perm is its inverse).
 * Ties are broken by the degree lists being LIFO, exactly as in the published code.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t idx_t;

#define AMD_FLIP(i) (-(i)-2)

static idx_t amd_wclear(idx_t mark, idx_t lemax, idx_t *w, idx_t n)
{
    if (mark < 2 || (mark + lemax < 0)) {
        for (idx_t k = 0; k < n; k++)
            if (w[k] != 0) w[k] = 1;
        mark = 2;
    }
    return mark; /* at this point, w[0..n-1] < mark holds */
}

/* depth-first search and postorder of a tree rooted at node j */
static idx_t amd_tdfs(idx_t j, idx_t k, idx_t *head, const idx_t *next, idx_t *post, idx_t *stack)
{
    idx_t top = 0;
    stack[0] = j;
    while (top >= 0) {
        const idx_t p = stack[top];
        const idx_t i = head[p];
        if (i == -1) {
            top--;
            post[k++] = p;
        } else {
            head[p] = next[i];
            stack[++top] = i;
        }
    }
    return k;
}

/* rowptr / col: the FULL symmetric pattern (both triangles, diagonal included), n columns.  order[n]: the pivots in order.
 * Returns 0, -1 when out of memory. */
int orc_amd_order(int64_t n64, const idx_t *rowptr, const idx_t *col, idx_t *order)
{
    const idx_t n = (idx_t)n64;
    if (n <= 0) return 0;
    idx_t dense = (idx_t)(10.0 * sqrt((double)n));
    if (dense < 16) dense = 16;
    if (dense > n - 2) dense = n - 2;
    idx_t cnz = rowptr[n];
    const int64_t t = (int64_t)cnz + cnz / 5 + 2 * (int64_t)n; /* elbow room */
    idx_t *Cp = (idx_t *)malloc(((size_t)n + 1) * sizeof(idx_t));
    idx_t *Ci = (idx_t *)malloc((size_t)(t > 0 ? t : 1) * sizeof(idx_t));
    idx_t *W = (idx_t *)malloc(8 * ((size_t)n + 1) * sizeof(idx_t));
    idx_t *P = (idx_t *)malloc(((size_t)n + 1) * sizeof(idx_t));
    if (!Cp || !Ci || !W || !P) {
        free(Cp); free(Ci); free(W); free(P);
        return -1;
    }
    memcpy(Cp, rowptr, ((size_t)n + 1) * sizeof(idx_t));
    memcpy(Ci, col, (size_t)cnz * sizeof(idx_t));
    idx_t *len = W, *nv = W + (n + 1), *next = W + 2 * (n + 1), *head = W + 3 * (n + 1), *elen = W + 4 * (n + 1),
          *degree = W + 5 * (n + 1), *w = W + 6 * (n + 1), *hhead = W + 7 * (n + 1), *last = P;
    idx_t i, j, k, k1, k2, k3, p, p1, p2, p3, p4, pj, pk, pk1, pk2, pn, q, d, dk, dext, e, eln, elenk, h, jlast, ln, lemax = 0,
          mark, mindeg = 0, nel = 0, nvi, nvj, nvk, wnvi, ok;
    const idx_t nzmax = (idx_t)t;

    /* --- initialize the quotient graph --- */
    for (k = 0; k < n; k++) len[k] = Cp[k + 1] - Cp[k];
    len[n] = 0;
    for (i = 0; i <= n; i++) {
        head[i] = -1;
        last[i] = -1;
        next[i] = -1;
        hhead[i] = -1;
        nv[i] = 1;
        w[i] = 1;
        elen[i] = 0;
        degree[i] = len[i];
    }
    mark = amd_wclear(0, 0, w, n);
    /* --- initialize the degree lists --- */
    for (i = 0; i < n; i++) {
        int has_diag = 0;
        for (p = Cp[i]; p < Cp[i + 1]; ++p)
            if (Ci[p] == i) {
                has_diag = 1;
                break;
            }
        d = degree[i];
        if (d == 1 && has_diag) { /* node i is empty */
            elen[i] = -2;
            nel++;
            Cp[i] = -1;
            w[i] = 0;
        } else if (d > dense || !has_diag) { /* dense, or no structural diagonal */
            nv[i] = 0;
            elen[i] = -1;
            nel++;
            Cp[i] = AMD_FLIP(n);
            nv[n]++;
        } else {
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            head[d] = i;
        }
    }
    elen[n] = -2;
    Cp[n] = -1;
    w[n] = 0;

    while (nel < n) {
        /* --- select the node of minimum approximate degree --- */
        for (k = -1; mindeg < n && (k = head[mindeg]) == -1; mindeg++) {}
        if (next[k] != -1) last[next[k]] = -1;
        head[mindeg] = next[k];
        elenk = elen[k];
        nvk = nv[k];
        nel += nvk;
        /* --- garbage collection --- */
        if (elenk > 0 && cnz + mindeg >= nzmax) {
            for (j = 0; j < n; j++) {
                if ((p = Cp[j]) >= 0) {
                    Cp[j] = Ci[p];
                    Ci[p] = AMD_FLIP(j);
                }
            }
            for (q = 0, p = 0; p < cnz;) {
                if ((j = AMD_FLIP(Ci[p++])) >= 0) {
                    Ci[q] = Cp[j];
                    Cp[j] = q++;
                    for (k3 = 0; k3 < len[j] - 1; k3++) Ci[q++] = Ci[p++];
                }
            }
            cnz = q;
        }
        /* --- construct the new element --- */
        dk = 0;
        nv[k] = -nvk;
        p = Cp[k];
        pk1 = (elenk == 0) ? p : cnz;
        pk2 = pk1;
        for (k1 = 1; k1 <= elenk + 1; k1++) {
            if (k1 > elenk) {
                e = k;
                pj = p;
                ln = len[k] - elenk;
            } else {
                e = Ci[p++];
                pj = Cp[e];
                ln = len[e];
            }
            for (k2 = 1; k2 <= ln; k2++) {
                i = Ci[pj++];
                if ((nvi = nv[i]) <= 0) continue;
                dk += nvi;
                nv[i] = -nvi;
                Ci[pk2++] = i;
                if (next[i] != -1) last[next[i]] = last[i];
                if (last[i] != -1) next[last[i]] = next[i];
                else head[degree[i]] = next[i];
            }
            if (e != k) {
                Cp[e] = AMD_FLIP(k);
                w[e] = 0;
            }
        }
        if (elenk != 0) cnz = pk2;
        degree[k] = dk;
        Cp[k] = pk1;
        len[k] = pk2 - pk1;
        elen[k] = -2;
        /* --- find set differences --- */
        mark = amd_wclear(mark, lemax, w, n);
        for (pk = pk1; pk < pk2; pk++) {
            i = Ci[pk];
            if ((eln = elen[i]) <= 0) continue;
            nvi = -nv[i];
            wnvi = mark - nvi;
            for (p = Cp[i]; p <= Cp[i] + eln - 1; p++) {
                e = Ci[p];
                if (w[e] >= mark) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }
        /* --- degree update --- */
        for (pk = pk1; pk < pk2; pk++) {
            i = Ci[pk];
            p1 = Cp[i];
            p2 = p1 + elen[i] - 1;
            pn = p1;
            for (h = 0, d = 0, p = p1; p <= p2; p++) {
                e = Ci[p];
                if (w[e] != 0) {
                    dext = w[e] - mark;
                    if (dext > 0) {
                        d += dext;
                        Ci[pn++] = e;
                        h += e;
                    } else {
                        Cp[e] = AMD_FLIP(k); /* aggressive absorption */
                        w[e] = 0;
                    }
                }
            }
            elen[i] = pn - p1 + 1;
            p3 = pn;
            p4 = p1 + len[i];
            for (p = p2 + 1; p < p4; p++) {
                j = Ci[p];
                if ((nvj = nv[j]) <= 0) continue;
                d += nvj;
                Ci[pn++] = j;
                h += j;
            }
            if (d == 0) { /* mass elimination */
                Cp[i] = AMD_FLIP(k);
                nvi = -nv[i];
                dk -= nvi;
                nvk += nvi;
                nel += nvi;
                nv[i] = 0;
                elen[i] = -1;
            } else {
                degree[i] = degree[i] < d ? degree[i] : d;
                Ci[pn] = Ci[p3];
                Ci[p3] = Ci[p1];
                Ci[p1] = k;
                len[i] = pn - p1 + 1;
                h = ((h < 0) ? (-h) : h) % n;
                next[i] = hhead[h];
                hhead[h] = i;
                last[i] = h;
            }
        }
        degree[k] = dk;
        lemax = lemax > dk ? lemax : dk;
        mark = amd_wclear(mark + lemax, lemax, w, n);
        /* --- supernode detection --- */
        for (pk = pk1; pk < pk2; pk++) {
            i = Ci[pk];
            if (nv[i] >= 0) continue;
            h = last[i];
            i = hhead[h];
            hhead[h] = -1;
            for (; i != -1 && next[i] != -1; i = next[i], mark++) {
                ln = len[i];
                eln = elen[i];
                for (p = Cp[i] + 1; p <= Cp[i] + ln - 1; p++) w[Ci[p]] = mark;
                jlast = i;
                for (j = next[i]; j != -1;) {
                    ok = (len[j] == ln) && (elen[j] == eln);
                    for (p = Cp[j] + 1; ok && p <= Cp[j] + ln - 1; p++)
                        if (w[Ci[p]] != mark) ok = 0;
                    if (ok) {
                        Cp[j] = AMD_FLIP(i);
                        nv[i] += nv[j];
                        nv[j] = 0;
                        elen[j] = -1;
                        j = next[j];
                        next[jlast] = j;
                    } else {
                        jlast = j;
                        j = next[j];
                    }
                }
            }
        }
        /* --- finalize the new element --- */
        for (p = pk1, pk = pk1; pk < pk2; pk++) {
            i = Ci[pk];
            if ((nvi = -nv[i]) <= 0) continue;
            nv[i] = nvi;
            d = degree[i] + dk - nvi;
            d = d < n - nel - nvi ? d : n - nel - nvi;
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            last[i] = -1;
            head[d] = i;
            mindeg = mindeg < d ? mindeg : d;
            degree[i] = d;
            Ci[p++] = i;
        }
        nv[k] = nvk;
        if ((len[k] = p - pk1) == 0) {
            Cp[k] = -1;
            w[k] = 0;
        }
        if (elenk != 0) cnz = p;
    }
    /* --- postordering --- */
    for (i = 0; i < n; i++) Cp[i] = AMD_FLIP(Cp[i]);
    for (j = 0; j <= n; j++) head[j] = -1;
    for (j = n; j >= 0; j--) {
        if (nv[j] > 0) continue;
        next[j] = head[Cp[j]];
        head[Cp[j]] = j;
    }
    for (e = n; e >= 0; e--) {
        if (nv[e] <= 0) continue;
        if (Cp[e] != -1) {
            next[e] = head[Cp[e]];
            head[Cp[e]] = e;
        }
    }
    for (k = 0, i = 0; i <= n; i++)
        if (Cp[i] == -1) k = amd_tdfs(i, k, head, next, P, w);
    /* P holds n + 1 entries (the dummy element n comes last); the ordering is the first n */
    for (i = 0, k = 0; i <= n && k < n; i++)
        if (P[i] != n) order[k++] = P[i];
    free(Cp);
    free(Ci);
    free(W);
    free(P);
    return 0;
}
