// amd_order.cpp -- approximate minimum degree ordering for precond = "ic" (host; round 4).
//
// Eigen::IncompleteCholesky<double> -- what the reference instantiates under the name "Eigen::IncompleteCholesky"
// (/root/reference/src/polysolve/linear/Solver.cpp:179-183) -- orders the matrix with AMDOrdering<int> before it factors it
// (Eigen 5.0.1 IncompleteCholesky.h: analyzePattern -> ord(mat.selfadjointView<UpLo>(), pinv); perm = pinv.inverse()).
// The ordering (Eigen/src/OrderingMethods/Amd.h: internal::minimum_degree_ordering) is Eigen's adaptation of CSparse's
// cs_amd (T. Davis, "Direct Methods for Sparse Linear Systems", 2006): quotient graph, approximate external degrees,
// aggressive element absorption, mass elimination, supernodes found by hashing, postordered assembly tree.  Neither Eigen
// nor CSparse is in the image; this is a restatement of the published algorithm with Eigen's two changes as recalled (the
// diagonal stays in the pattern: a node whose only entry is its diagonal is eliminated at once, one without a diagonal is
// treated as dense; dense = more than max(16, 10 sqrt(n)) entries) -- parity unpinned, like oracle/amd_oracle.c, an
// independent transcription the CPU tests compare this one with entry by entry.
// Why it matters on a GPU beyond the name: the natural ordering of an N^3 grid gives the triangular solves 3 N dependency
// levels; a fill-reducing order gives a bushy elimination tree -- far fewer levels for ic.hip's waiting solves.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "ic.hpp"

namespace psolve {

namespace {

inline int32_t flip(int32_t i) { return -i - 2; }

struct Amd {
    int32_t n;
    std::vector<int32_t> Cp, Ci, len, nv, next, head, elen, degree, w, hhead, last;

    int32_t wclear(int32_t mark, int32_t lemax)
    {
        if (mark < 2 || mark + lemax < 0) {
            for (int32_t k = 0; k < n; ++k)
                if (w[(size_t)k] != 0) w[(size_t)k] = 1;
            mark = 2;
        }
        return mark;
    }

    // postorder of the subtree rooted at j (children lists in head / next), appended to post from position k
    int32_t tdfs(int32_t j, int32_t k, std::vector<int32_t> &post, std::vector<int32_t> &stack)
    {
        int32_t top = 0;
        stack[0] = j;
        while (top >= 0) {
            const int32_t p = stack[(size_t)top], i = head[(size_t)p];
            if (i == -1) {
                --top;
                post[(size_t)k++] = p;
            } else {
                head[(size_t)p] = next[(size_t)i];
                stack[(size_t)++top] = i;
            }
        }
        return k;
    }
};

// A u A^T by rows (sorted, duplicates merged) into (sp, sc) when the pattern as given is not already symmetric; returns
// false -- and leaves the caller's arrays in use -- when it is.  Indices outside [0, n) are refused.
bool symmetrised_pattern(int32_t n, const int32_t *rowptr, const int32_t *col, std::vector<int32_t> &sp,
                         std::vector<int32_t> &sc)
{
    const int64_t nnz = rowptr[n];
    PS_REQUIRE(rowptr[0] == 0 && nnz >= 0, PSOLVE_HIP_EINVAL, "amd ordering: bad row pointers");
    // The usual caller (the IC setup) hands over a structurally symmetric pattern with sorted, unique columns: checked first
    // -- strictly increasing rows, then a binary search for (j, i) per entry (i, j), leaving at the first miss -- so that the
    // transpose and the merged copy below (3 x nnz of int32 and an O(nnz log) pass) are built only when something differs.
    {
        bool clean = true;
        for (int32_t i = 0; i < n && clean; ++i) {
            PS_REQUIRE(rowptr[i + 1] >= rowptr[i], PSOLVE_HIP_EINVAL, "amd ordering: row pointers decrease");
            for (int32_t p = rowptr[i]; p < rowptr[i + 1] && clean; ++p) {
                PS_REQUIRE(col[p] >= 0 && col[p] < n, PSOLVE_HIP_ERANGE, "amd ordering: column index out of range");
                if (p > rowptr[i] && col[p] <= col[p - 1]) clean = false; // unsorted or duplicated: the merge below decides
            }
        }
        for (int32_t i = 0; i < n && clean; ++i)
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                const int32_t j = col[p];
                if (j == i) continue;
                if (!std::binary_search(col + rowptr[j], col + rowptr[j + 1], i)) { // (rows are sorted: checked above)
                    clean = false;
                    break;
                }
            }
        if (clean) return false; // symmetric as given: the caller's arrays stay in use
    }
    std::vector<int32_t> tp((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; ++i) {
        PS_REQUIRE(rowptr[i + 1] >= rowptr[i], PSOLVE_HIP_EINVAL, "amd ordering: row pointers decrease");
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            PS_REQUIRE(col[p] >= 0 && col[p] < n, PSOLVE_HIP_ERANGE, "amd ordering: column index out of range");
            ++tp[(size_t)col[p] + 1];
        }
    }
    for (int32_t i = 0; i < n; ++i) tp[(size_t)i + 1] += tp[(size_t)i];
    std::vector<int32_t> tc((size_t)nnz), fill(tp.begin(), tp.end() - 1);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) tc[(size_t)fill[(size_t)col[p]]++] = i; // rows of A^T come out sorted
    std::vector<int32_t> row;
    sp.assign((size_t)n + 1, 0);
    sc.clear();
    sc.reserve((size_t)nnz);
    bool differs = false;
    for (int32_t i = 0; i < n; ++i) {
        row.assign(col + rowptr[i], col + rowptr[i + 1]);
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        if (row.size() != (size_t)(rowptr[i + 1] - rowptr[i])) differs = true; // duplicates: the merged list replaces them
        const int32_t *a = row.data(), *ae = a + row.size();
        const int32_t *b = tc.data() + tp[(size_t)i], *be = tc.data() + tp[(size_t)i + 1];
        while (a < ae || b < be) {
            int32_t c;
            if (b == be || (a < ae && *a < *b)) {
                c = *a++;
                differs = true; // (i, c) without (c, i)
            } else if (a == ae || *b < *a) {
                c = *b++;
                differs = true;
            } else {
                c = *a;
                ++a;
                ++b;
            }
            while (b < be && *b == c) ++b; // duplicates of the transposed side
            PS_REQUIRE(sc.size() < (size_t)INT32_MAX, PSOLVE_HIP_ERANGE, "amd ordering: symmetrised pattern exceeds int32 indexing");
            sc.push_back(c);
        }
        sp[(size_t)i + 1] = (int32_t)sc.size();
    }
    return differs;
}

} // namespace

void amd_order(int64_t n64, const int32_t *rowptr, const int32_t *col, std::vector<int32_t> &order)
{
    const int32_t n = (int32_t)n64;
    order.assign((size_t)std::max<int32_t>(n, 0), 0);
    if (n <= 0) return;
    int32_t dense = (int32_t)(10.0 * std::sqrt((double)n));
    dense = std::max<int32_t>(16, dense);
    dense = std::min<int32_t>(n - 2, dense);
    // The quotient graph below is that of a symmetric pattern (Eigen hands the ordering mat.selfadjointView, i.e. A + A^T's
    // pattern).  A caller's pattern that is not symmetric is symmetrised here (A u A^T) instead of being trusted: on an
    // unsymmetric list the element lists can outgrow the 1.2 nnz + 2 n workspace the algorithm's bound is proved for.
    std::vector<int32_t> sym_ptr, sym_col;
    if (symmetrised_pattern(n, rowptr, col, sym_ptr, sym_col)) {
        rowptr = sym_ptr.data();
        col = sym_col.data();
    }
    int32_t cnz = rowptr[n];
    const int64_t room = (int64_t)cnz + cnz / 5 + 2 * (int64_t)n;
    PS_REQUIRE(room < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "amd ordering: pattern exceeds int32 indexing");
    const int32_t nzmax = (int32_t)room;
    Amd S;
    S.n = n;
    S.Cp.assign(rowptr, rowptr + n + 1);
    S.Ci.assign((size_t)std::max<int32_t>(nzmax, 1), 0);
    std::copy(col, col + cnz, S.Ci.begin());
    const size_t m = (size_t)n + 1;
    S.len.assign(m, 0);
    S.nv.assign(m, 1);
    S.next.assign(m, -1);
    S.head.assign(m, -1);
    S.elen.assign(m, 0);
    S.degree.assign(m, 0);
    S.w.assign(m, 1);
    S.hhead.assign(m, -1);
    S.last.assign(m, -1);
    auto &Cp = S.Cp, &Ci = S.Ci, &len = S.len, &nv = S.nv, &next = S.next, &head = S.head, &elen = S.elen, &degree = S.degree,
         &w = S.w, &hhead = S.hhead, &last = S.last;

    for (int32_t k = 0; k < n; ++k) len[(size_t)k] = Cp[(size_t)k + 1] - Cp[(size_t)k];
    len[(size_t)n] = 0;
    for (int32_t i = 0; i <= n; ++i) degree[(size_t)i] = len[(size_t)i];
    int32_t mark = S.wclear(0, 0), nel = 0, mindeg = 0, lemax = 0;

    // degree lists (last-in first-out: the list head is the most recently inserted node)
    for (int32_t i = 0; i < n; ++i) {
        bool has_diag = false;
        for (int32_t p = Cp[(size_t)i]; p < Cp[(size_t)i + 1]; ++p)
            if (Ci[(size_t)p] == i) {
                has_diag = true;
                break;
            }
        const int32_t d = degree[(size_t)i];
        if (d == 1 && has_diag) { // an empty node: a root of the assembly tree
            elen[(size_t)i] = -2;
            ++nel;
            Cp[(size_t)i] = -1;
            w[(size_t)i] = 0;
        } else if (d > dense || !has_diag) { // dense (or no structural diagonal): absorbed into the dummy element n
            nv[(size_t)i] = 0;
            elen[(size_t)i] = -1;
            ++nel;
            Cp[(size_t)i] = flip(n);
            ++nv[(size_t)n];
        } else {
            if (head[(size_t)d] != -1) last[(size_t)head[(size_t)d]] = i;
            next[(size_t)i] = head[(size_t)d];
            head[(size_t)d] = i;
        }
    }
    elen[(size_t)n] = -2;
    Cp[(size_t)n] = -1;
    w[(size_t)n] = 0;

    while (nel < n) {
        // the node of minimum approximate degree
        int32_t k = -1;
        for (; mindeg < n && (k = head[(size_t)mindeg]) == -1; ++mindeg) {}
        if (next[(size_t)k] != -1) last[(size_t)next[(size_t)k]] = -1;
        head[(size_t)mindeg] = next[(size_t)k];
        const int32_t elenk = elen[(size_t)k];
        int32_t nvk = nv[(size_t)k];
        nel += nvk;
        // garbage collection when the new element may not fit behind the used part of Ci
        if (elenk > 0 && cnz + mindeg >= nzmax) {
            for (int32_t j = 0; j < n; ++j) {
                const int32_t p = Cp[(size_t)j];
                if (p >= 0) {
                    Cp[(size_t)j] = Ci[(size_t)p];
                    Ci[(size_t)p] = flip(j);
                }
            }
            int32_t q = 0;
            for (int32_t p = 0; p < cnz;) {
                const int32_t j = flip(Ci[(size_t)p++]);
                if (j >= 0) {
                    Ci[(size_t)q] = Cp[(size_t)j];
                    Cp[(size_t)j] = q++;
                    for (int32_t k3 = 0; k3 < len[(size_t)j] - 1; ++k3) Ci[(size_t)q++] = Ci[(size_t)p++];
                }
            }
            cnz = q;
        }
        // the new element Lk: the live nodes of k's own list and of the elements it is adjacent to
        int32_t dk = 0;
        nv[(size_t)k] = -nvk;
        int32_t p = Cp[(size_t)k];
        const int32_t pk1 = (elenk == 0) ? p : cnz;
        int32_t pk2 = pk1;
        for (int32_t k1 = 1; k1 <= elenk + 1; ++k1) {
            int32_t e, pj, ln;
            if (k1 > elenk) {
                e = k;
                pj = p;
                ln = len[(size_t)k] - elenk;
            } else {
                e = Ci[(size_t)p++];
                pj = Cp[(size_t)e];
                ln = len[(size_t)e];
            }
            for (int32_t k2 = 1; k2 <= ln; ++k2) {
                const int32_t i = Ci[(size_t)pj++];
                const int32_t nvi = nv[(size_t)i];
                if (nvi <= 0) continue;
                dk += nvi;
                nv[(size_t)i] = -nvi;
                PS_REQUIRE(pk2 < nzmax, PSOLVE_HIP_ERANGE, "amd ordering: element list outgrew its workspace");
                Ci[(size_t)pk2++] = i;
                if (next[(size_t)i] != -1) last[(size_t)next[(size_t)i]] = last[(size_t)i];
                if (last[(size_t)i] != -1) next[(size_t)last[(size_t)i]] = next[(size_t)i];
                else head[(size_t)degree[(size_t)i]] = next[(size_t)i];
            }
            if (e != k) {
                Cp[(size_t)e] = flip(k);
                w[(size_t)e] = 0;
            }
        }
        if (elenk != 0) cnz = pk2;
        degree[(size_t)k] = dk;
        Cp[(size_t)k] = pk1;
        len[(size_t)k] = pk2 - pk1;
        elen[(size_t)k] = -2;
        // scan 1: |Le \ Lk| for every element e adjacent to a node of Lk
        mark = S.wclear(mark, lemax);
        for (int32_t pk = pk1; pk < pk2; ++pk) {
            const int32_t i = Ci[(size_t)pk], eln = elen[(size_t)i];
            if (eln <= 0) continue;
            const int32_t nvi = -nv[(size_t)i], wnvi = mark - nvi;
            for (int32_t q = Cp[(size_t)i]; q <= Cp[(size_t)i] + eln - 1; ++q) {
                const int32_t e = Ci[(size_t)q];
                if (w[(size_t)e] >= mark) w[(size_t)e] -= nvi;
                else if (w[(size_t)e] != 0) w[(size_t)e] = degree[(size_t)e] + wnvi;
            }
        }
        // scan 2: degree update, aggressive absorption, mass elimination, hash for the supernode search
        for (int32_t pk = pk1; pk < pk2; ++pk) {
            const int32_t i = Ci[(size_t)pk];
            const int32_t p1 = Cp[(size_t)i], p2 = p1 + elen[(size_t)i] - 1;
            int32_t pn = p1, h = 0, d = 0;
            for (int32_t q = p1; q <= p2; ++q) {
                const int32_t e = Ci[(size_t)q];
                if (w[(size_t)e] != 0) {
                    const int32_t dext = w[(size_t)e] - mark;
                    if (dext > 0) {
                        d += dext;
                        Ci[(size_t)pn++] = e;
                        h += e;
                    } else {
                        Cp[(size_t)e] = flip(k);
                        w[(size_t)e] = 0;
                    }
                }
            }
            elen[(size_t)i] = pn - p1 + 1;
            const int32_t p3 = pn, p4 = p1 + len[(size_t)i];
            for (int32_t q = p2 + 1; q < p4; ++q) {
                const int32_t j = Ci[(size_t)q], nvj = nv[(size_t)j];
                if (nvj <= 0) continue;
                d += nvj;
                Ci[(size_t)pn++] = j;
                h += j;
            }
            if (d == 0) {
                Cp[(size_t)i] = flip(k);
                const int32_t nvi = -nv[(size_t)i];
                dk -= nvi;
                nvk += nvi;
                nel += nvi;
                nv[(size_t)i] = 0;
                elen[(size_t)i] = -1;
            } else {
                degree[(size_t)i] = std::min(degree[(size_t)i], d);
                Ci[(size_t)pn] = Ci[(size_t)p3];
                Ci[(size_t)p3] = Ci[(size_t)p1];
                Ci[(size_t)p1] = k;
                len[(size_t)i] = pn - p1 + 1;
                h = ((h < 0) ? (-h) : h) % n;
                next[(size_t)i] = hhead[(size_t)h];
                hhead[(size_t)h] = i;
                last[(size_t)i] = h;
            }
        }
        degree[(size_t)k] = dk;
        lemax = std::max(lemax, dk);
        mark = S.wclear(mark + lemax, lemax);
        // supernodes: nodes of Lk with identical adjacency (same hash bucket, same lists)
        for (int32_t pk = pk1; pk < pk2; ++pk) {
            int32_t i = Ci[(size_t)pk];
            if (nv[(size_t)i] >= 0) continue;
            const int32_t h = last[(size_t)i];
            i = hhead[(size_t)h];
            hhead[(size_t)h] = -1;
            for (; i != -1 && next[(size_t)i] != -1; i = next[(size_t)i], ++mark) {
                const int32_t ln = len[(size_t)i], eln = elen[(size_t)i];
                for (int32_t q = Cp[(size_t)i] + 1; q <= Cp[(size_t)i] + ln - 1; ++q) w[(size_t)Ci[(size_t)q]] = mark;
                int32_t jlast = i;
                for (int32_t j = next[(size_t)i]; j != -1;) {
                    bool same = len[(size_t)j] == ln && elen[(size_t)j] == eln;
                    for (int32_t q = Cp[(size_t)j] + 1; same && q <= Cp[(size_t)j] + ln - 1; ++q)
                        if (w[(size_t)Ci[(size_t)q]] != mark) same = false;
                    if (same) {
                        Cp[(size_t)j] = flip(i);
                        nv[(size_t)i] += nv[(size_t)j];
                        nv[(size_t)j] = 0;
                        elen[(size_t)j] = -1;
                        j = next[(size_t)j];
                        next[(size_t)jlast] = j;
                    } else {
                        jlast = j;
                        j = next[(size_t)j];
                    }
                }
            }
        }
        // the new element is final: its nodes go back into the degree lists with their external degrees
        int32_t pf = pk1;
        for (int32_t pk = pk1; pk < pk2; ++pk) {
            const int32_t i = Ci[(size_t)pk], nvi = -nv[(size_t)i];
            if (nvi <= 0) continue;
            nv[(size_t)i] = nvi;
            int32_t d = degree[(size_t)i] + dk - nvi;
            d = std::min(d, n - nel - nvi);
            if (head[(size_t)d] != -1) last[(size_t)head[(size_t)d]] = i;
            next[(size_t)i] = head[(size_t)d];
            last[(size_t)i] = -1;
            head[(size_t)d] = i;
            mindeg = std::min(mindeg, d);
            degree[(size_t)i] = d;
            Ci[(size_t)pf++] = i;
        }
        nv[(size_t)k] = nvk;
        if ((len[(size_t)k] = pf - pk1) == 0) {
            Cp[(size_t)k] = -1;
            w[(size_t)k] = 0;
        }
        if (elenk != 0) cnz = pf;
    }
    // postorder of the assembly tree
    for (int32_t i = 0; i < n; ++i) Cp[(size_t)i] = flip(Cp[(size_t)i]);
    for (int32_t j = 0; j <= n; ++j) head[(size_t)j] = -1;
    for (int32_t j = n; j >= 0; --j) {
        if (nv[(size_t)j] > 0) continue;
        next[(size_t)j] = head[(size_t)Cp[(size_t)j]];
        head[(size_t)Cp[(size_t)j]] = j;
    }
    for (int32_t e = n; e >= 0; --e) {
        if (nv[(size_t)e] <= 0) continue;
        if (Cp[(size_t)e] != -1) {
            next[(size_t)e] = head[(size_t)Cp[(size_t)e]];
            head[(size_t)Cp[(size_t)e]] = e;
        }
    }
    std::vector<int32_t> post(m, 0);
    int32_t k = 0;
    for (int32_t i = 0; i <= n; ++i)
        if (Cp[(size_t)i] == -1) k = S.tdfs(i, k, post, w);
    int32_t out = 0;
    for (int32_t i = 0; i <= n && out < n; ++i)
        if (post[(size_t)i] != n) order[(size_t)out++] = post[(size_t)i];
}

} // namespace psolve
