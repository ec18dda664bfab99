// ic_factor.cpp -- host factorization of precond = "ic" (see ic.hpp).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <utility>

#include "ic.hpp"

namespace psolve {

namespace {

// the ncut entries of largest magnitude first (a quickselect partition; Eigen::internal::QuickSplit, IncompleteLUT.h)
void split_largest(std::vector<double> &v, std::vector<int32_t> &idx, int64_t count, int64_t ncut)
{
    int64_t first = 0, last = count - 1, mid;
    --ncut;
    if (ncut < first || ncut > last) return;
    do {
        mid = first;
        const double key = std::fabs(v[(size_t)mid]);
        for (int64_t j = first + 1; j <= last; ++j)
            if (std::fabs(v[(size_t)j]) > key) {
                ++mid;
                std::swap(v[(size_t)mid], v[(size_t)j]);
                std::swap(idx[(size_t)mid], idx[(size_t)j]);
            }
        std::swap(v[(size_t)mid], v[(size_t)first]);
        std::swap(idx[(size_t)mid], idx[(size_t)first]);
        if (mid > ncut) last = mid - 1;
        else if (mid < ncut) first = mid + 1;
    } while (mid != ncut);
}

// the columns waiting to update a given column, in arrival order; a column waits in one queue at a time
struct Queues {
    std::vector<int32_t> head, tail, next;
    explicit Queues(size_t n) : head(n, -1), tail(n, -1), next(n, -1) {}
    void clear()
    {
        std::fill(head.begin(), head.end(), -1);
        std::fill(tail.begin(), tail.end(), -1);
        std::fill(next.begin(), next.end(), -1);
    }
    void push(int32_t row, int32_t col)
    {
        next[(size_t)col] = -1;
        if (head[(size_t)row] < 0) head[(size_t)row] = col;
        else next[(size_t)tail[(size_t)row]] = col;
        tail[(size_t)row] = col;
    }
};

} // namespace

void ic_factorize(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val, double initial_shift, IcFactor &F)
{
    F = IcFactor();
    F.n = n;
    F.colptr.assign((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; ++j) {
        bool diag = false;
        int32_t cnt = 0;
        for (int32_t k = rowptr[j]; k < rowptr[j + 1]; ++k) {
            diag = diag || col[k] == j;
            cnt += col[k] >= j;
        }
        PS_REQUIRE(diag, PSOLVE_HIP_ENUMERIC, "incomplete Cholesky: row " + std::to_string(j) + " has no stored diagonal entry");
        F.colptr[(size_t)j + 1] = F.colptr[(size_t)j] + cnt;
    }
    const int64_t nnz = F.colptr[(size_t)n];
    F.rowidx.resize((size_t)nnz);
    F.vals.resize((size_t)nnz);
    F.scale.assign((size_t)n, 0.0);
    std::vector<int32_t> &cp = F.colptr, &ri = F.rowidx;
    std::vector<double> &lv = F.vals, &sc = F.scale;
    for (int64_t j = 0; j < n; ++j) {
        int32_t w = cp[(size_t)j];
        for (int32_t k = rowptr[j]; k < rowptr[j + 1]; ++k)
            if (col[k] >= j) {
                ri[(size_t)w] = col[k];
                lv[(size_t)w] = val[k];
                ++w;
            }
    }
    // S = diag(1 / sqrt(2-norm of the column of the symmetric matrix))
    for (int64_t j = 0; j < n; ++j)
        for (int32_t k = cp[(size_t)j]; k < cp[(size_t)j + 1]; ++k) {
            const double a2 = lv[(size_t)k] * lv[(size_t)k];
            sc[(size_t)j] += a2;
            if (ri[(size_t)k] != j) sc[(size_t)ri[(size_t)k]] += a2;
        }
    for (int64_t j = 0; j < n; ++j) {
        const double s = std::sqrt(std::sqrt(sc[(size_t)j]));
        sc[(size_t)j] = s > DBL_MIN ? 1.0 / s : 1.0;
    }
    double mindiag = DBL_MAX;
    for (int64_t j = 0; j < n; ++j) {
        for (int32_t k = cp[(size_t)j]; k < cp[(size_t)j + 1]; ++k) lv[(size_t)k] *= sc[(size_t)j] * sc[(size_t)ri[(size_t)k]];
        mindiag = std::min(mindiag, lv[(size_t)cp[(size_t)j]]);
    }
    const std::vector<int32_t> ri0 = ri;
    const std::vector<double> lv0 = lv;
    double shift = mindiag <= 0.0 ? initial_shift - mindiag : 0.0;
    std::vector<int32_t> next_entry((size_t)n, 0), slot((size_t)n, -1), wrow((size_t)n);
    std::vector<double> wval((size_t)n);
    Queues Q((size_t)n);
    // column c hands its next entry (the smallest remaining row index, moved to position pos) to that row's queue
    auto advance = [&](int32_t c, int32_t pos) {
        const int32_t end = cp[(size_t)c + 1];
        if (pos >= end) return;
        int32_t m = pos;
        for (int32_t q = pos + 1; q < end; ++q)
            if (ri[(size_t)q] < ri[(size_t)m]) m = q;
        if (ri[(size_t)m] != ri[(size_t)pos]) {
            std::swap(ri[(size_t)m], ri[(size_t)pos]);
            std::swap(lv[(size_t)m], lv[(size_t)pos]);
        }
        next_entry[(size_t)c] = pos;
        Q.push(ri[(size_t)pos], c);
    };
    int attempts = 0;
    bool ok = false, gave_up = false;
    while (!ok && !gave_up) {
        for (int64_t j = 0; j < n; ++j) lv[(size_t)cp[(size_t)j]] += shift;
        int64_t j = 0;
        for (; j < n; ++j) {
            const int32_t cb = cp[(size_t)j], ce = cp[(size_t)j + 1];
            const double diag = lv[(size_t)cb];
            int64_t cnt = 0;
            for (int32_t i = cb + 1; i < ce; ++i) {
                wval[(size_t)cnt] = lv[(size_t)i];
                wrow[(size_t)cnt] = ri[(size_t)i];
                slot[(size_t)ri[(size_t)i]] = (int32_t)cnt;
                ++cnt;
            }
            for (int32_t k = Q.head[(size_t)j]; k >= 0;) {
                const int32_t knext = Q.next[(size_t)k];
                int32_t pos = next_entry[(size_t)k];
                const double ljk = lv[(size_t)pos];
                ++pos;
                for (int32_t i = pos; i < cp[(size_t)k + 1]; ++i) {
                    const int32_t r = ri[(size_t)i];
                    const double u = lv[(size_t)i] * ljk;
                    if (slot[(size_t)r] < 0) { // fill-in
                        wval[(size_t)cnt] = -u;
                        wrow[(size_t)cnt] = r;
                        slot[(size_t)r] = (int32_t)cnt;
                        ++cnt;
                    } else {
                        wval[(size_t)slot[(size_t)r]] -= u;
                    }
                }
                advance(k, pos);
                k = knext;
            }
            Q.head[(size_t)j] = Q.tail[(size_t)j] = -1;
            if (diag <= 0.0) { // the shift was too small: start over with twice as much
                if (++attempts >= 10) {
                    gave_up = true;
                    break;
                }
                shift = std::max(initial_shift, 2.0 * shift);
                ri = ri0;
                lv = lv0;
                std::fill(slot.begin(), slot.end(), -1);
                Q.clear();
                break;
            }
            const double rd = std::sqrt(diag);
            lv[(size_t)cb] = rd;
            for (int64_t k = 0; k < cnt; ++k) {
                wval[(size_t)k] /= rd;
                lv[(size_t)cp[(size_t)wrow[(size_t)k]]] -= wval[(size_t)k] * wval[(size_t)k];
            }
            split_largest(wval, wrow, cnt, (int64_t)(ce - cb - 1));
            for (int64_t k = 0; k < cnt; ++k) slot[(size_t)wrow[(size_t)k]] = -1;
            for (int32_t i = cb + 1, k = 0; i < ce; ++i, ++k) {
                lv[(size_t)i] = wval[(size_t)k];
                ri[(size_t)i] = wrow[(size_t)k];
            }
            advance((int32_t)j, cb + 1);
        }
        ok = j == n;
    }
    F.shift = shift;
    F.attempts = attempts + 1;
    F.ok = ok;
}

} // namespace psolve
