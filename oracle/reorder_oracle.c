/* reorder_oracle.c -- CPU statement of the backend's optional renumbering ("reorder"): Cuthill-McKee by
 * breadth-first search.  TEST INFRASTRUCTURE ONLY (see psolve_oracle.c): the product never links this file.
 *
 * There is no Eigen / AMGCL counterpart: the reference's iterative backends take the caller's numbering as it is.
 * The reference precedent for renumbering INSIDE a backend is MASSolver, which partitions the matrix graph and
 * permutes the system before building its preconditioner
 * (/root/reference/src/polysolve/linear/mas_utils/GraphPartition.cpp:240-243, MASSolver.cu:304-321).  What this file
 * pins is the DEFINITION the device kernels (polysolve_amd/csrc/reorder.hip) must reproduce bit for bit -- the order
 * is integer work -- and, through it, parity of a reordered solve: the HIP solve with `reorder` equals the oracle's
 * solve of the explicitly permuted system P A P^T (P b).
 *
 * Definition (classical Cuthill-McKee [Cuthill & McKee 1969] without the degree sort inside a level, which the
 * level-parallel construction has no use for):
 *   1. rows without an off-diagonal entry first, ascending;
 *   2. start vertex of a component: fewest stored entries in its row, smallest index among those;
 *   3. breadth-first search, a dequeued vertex appends its unvisited neighbours in the order its row stores them;
 *   4. after max_components components the remaining vertices follow in index order.
 * order[k] = old index of the vertex at new position k.  info[0..3] = levels, components, isolated, leftover. */
#include <stdint.h>
#include <stdlib.h>

int orc_cuthill_mckee(int64_t n, const int32_t *ptr, const int32_t *col, int max_components, int32_t *order,
                      int64_t *info)
{
    int32_t *pos = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    if (!pos) return -1;
    int64_t placed = 0, levels = 0, comps = 0, iso = 0, leftover = 0;
    for (int64_t i = 0; i < n; ++i) {
        int isolated = 1;
        for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k)
            if (col[k] != i) {
                isolated = 0;
                break;
            }
        pos[i] = -1;
        if (isolated) {
            order[placed] = (int32_t)i;
            pos[i] = (int32_t)placed++;
            ++iso;
        }
    }
    while (placed < n) {
        if (comps == max_components) {
            for (int64_t i = 0; i < n; ++i)
                if (pos[i] < 0) {
                    order[placed] = (int32_t)i;
                    pos[i] = (int32_t)placed++;
                    ++leftover;
                }
            break;
        }
        int64_t start = -1;
        int32_t best = INT32_MAX;
        for (int64_t i = 0; i < n; ++i)
            if (pos[i] < 0 && ptr[i + 1] - ptr[i] < best) {
                best = ptr[i + 1] - ptr[i];
                start = i;
            }
        ++comps;
        order[placed] = (int32_t)start;
        pos[start] = (int32_t)placed;
        int64_t head = placed, level_end = placed + 1;
        ++placed;
        ++levels;
        while (head < placed) {
            if (head == level_end) {
                level_end = placed;
                ++levels;
            }
            const int32_t v = order[head++];
            for (int32_t k = ptr[v]; k < ptr[v + 1]; ++k) {
                const int32_t w = col[k];
                if (pos[w] < 0) {
                    order[placed] = w;
                    pos[w] = (int32_t)placed++;
                }
            }
        }
    }
    if (info) {
        info[0] = levels;
        info[1] = comps;
        info[2] = iso;
        info[3] = leftover;
    }
    free(pos);
    return 0;
}
