// reorder.hpp -- optional locality renumbering of the factorized operator ("reorder").
//
// Precedent in the reference: the only in-tree GPU backend renumbers the unknowns before it builds anything --
// MASSolver partitions the matrix graph and permutes the system at analyze / factorize time
// (/root/reference/src/polysolve/linear/mas_utils/GraphPartition.cpp:240-243, MASSolver.cu:304-321
// `lazy_partitioning`).  Here the renumbering serves the products: a caller's mesh numbering decides how many
// distinct cache lines the gathers of a wave touch (bench: the 256^3 Poisson matrix runs at 0.70 of the HBM peak in
// the grid numbering, 0.39 with the rows shuffled inside 4096-row windows, 0.11 under a random permutation).
//
// The order is Cuthill-McKee by breadth-first levels, defined sequentially (oracle/reorder_oracle.c):
//   1. rows without an off-diagonal entry (Dirichlet rows after elimination, FEMSolver.cpp:136-161) first, ascending;
//   2. start vertex: the unvisited vertex of the fewest stored entries, the smallest index among those;
//   3. breadth-first search; a dequeued vertex appends its unvisited neighbours in the order its row stores them;
//   4. the next component from 2.; after kMaxComponents components the remaining vertices follow in index order.
// The device builds exactly this sequence level by level (reorder.hip): integer work, bit-exact against the oracle.
#pragma once
#include "amg_symbolic.hpp"
#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

constexpr int kReorderMaxComponents = 64;

struct ReorderScratch {
    DeviceBuffer<int> claim, cnt, tsum, state;
    DeviceBuffer<unsigned long long> key;
    PinnedBuffer<int> host;
};

struct ReorderInfo {
    int levels = 0;      // breadth-first levels walked (all components)
    int components = 0;  // components walked by the search (without the isolated rows)
    int isolated = 0;    // rows without an off-diagonal entry
    int leftover = 0;    // vertices appended in index order after kReorderMaxComponents components
};

// order[k] = old index of the vertex at new position k, new_of_old[order[k]] = k, for the graph (ptr, col) of n
// vertices (symmetric pattern, no duplicate columns inside a row).  Synchronises the stream.
void device_cuthill_mckee(const Launch &L, int n, const int *ptr, const int *col, int *order, int *new_of_old,
                          ReorderScratch &W, SymbolicScratch &S, ReorderInfo *info);

// dof_order[b k + c] = b order[k] + c and its inverse (block_size b: whole nodes move)
void launch_reverse_order(const Launch &L, int n, int *order, int *new_of_old); // the order read backwards
void launch_expand_node_order(const Launch &L, int nb, int b, const int *order, int *dof_order, int *dof_new_of_old);

// locality figure of a numbering: the distinct lines of eight consecutive unknowns (64 bytes of a vector; with block
// value types eight consecutive NODES, col / b) that the gathers of 64 consecutive rows touch, divided by the fewest lines
// that many distinct unknowns could occupy (1 = perfectly dense; a 7-point grid or a Q1 elasticity mesh in its natural
// order: ~1.1; a random numbering: ~8).  Sampled (every `stride`-th group); synchronises.
double device_gather_spread(const Launch &L, int n, const int *ptr, const int *col, int b, int stride, SymbolicScratch &S);

} // namespace psolve
