// ic.hpp -- incomplete Cholesky preconditioner (precond = "ic"): Eigen::IncompleteCholesky in its default AMDOrdering
// ("ic.ordering" 1, amd_order.cpp; round 4) or in the NaturalOrdering (0), the preconditioner the reference reaches through the name "Eigen::IncompleteCholesky"
// (/root/reference/src/polysolve/linear/Solver.cpp:179-183, 591-604; Eigen 5.0.1 IncompleteCholesky.h).
//
//   factorize (host, ic_factor.cpp): Eigen's algorithm -- symmetric scaling S = diag(||col_j||_2)^-1/2, Lin-More style
//     shift on the scaled diagonal (restart with a doubled shift after a non-positive pivot), left-looking
//     factorization column by column, each column keeping as many off-diagonal entries as the matrix column has (the
//     largest in magnitude).  Sequential by construction (every column depends on the dropping decisions of the
//     columns before it), like the aggregation sweep of AMGCL it stays on the host; what the reference's default adds,
//     the AMD ordering, is applied on the host before the factorization since round 4 (amd_order.cpp).
//   apply (device, ic.hip): z = S L^-T L^-1 S r by two triangular solves in which every row waits for the rows it
//     depends on inside ONE kernel: rows are laid out level by level (dependency depth), a thread owns a row and takes
//     its entries in order, each as soon as the flag of the column says its value is final; completion happens inside
//     the polling loop, so lanes of one wavefront may depend on each other, and a row only ever waits for rows placed
//     before it, whose workgroups were dispatched before its own: no deadlock whatever the residency.
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

class Context;

// L by columns (diagonal first, the rest in the order the factorization left them) + the scaling
struct IcFactor {
    int64_t n = 0;
    std::vector<int32_t> colptr, rowidx;
    std::vector<double> vals, scale;
    double shift = 0.0;
    int attempts = 0;
    bool ok = false;
};

// host only; rowptr / col / val: CSC (= CSR of a symmetric matrix) with sorted inner indices, entries with row >= column
// are read.  Throws PSOLVE_HIP_ENUMERIC when a column has no stored diagonal.
void ic_factorize(int64_t n, const int32_t *rowptr, const int32_t *col, const double *val, double initial_shift, IcFactor &F);

// host only (amd_order.cpp): Eigen::AMDOrdering<int> restated -- order[k] = the k-th pivot of the approximate minimum degree
// ordering of a symmetric pattern given in full (both triangles, diagonal included)
void amd_order(int64_t n, const int32_t *rowptr, const int32_t *col, std::vector<int32_t> &order);

class IcPrecond {
public:
    // A: the (shard's diagonal block of the) factorized matrix on the device
    // ordering: 0 = natural (NaturalOrdering<int>), 1 = approximate minimum degree (AMDOrdering<int>: the class template's default)
    void setup(Context &ctx, const CsrDev &A, double initial_shift, int ordering = 0);
    void apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag = nullptr);
    int rows() const { return n_; }
    double shift() const { return shift_; }
    int attempts() const { return attempts_; }
    int levels_forward() const { return lev_f_; }
    int levels_backward() const { return lev_b_; }
    int ordering() const { return ordering_; }
    // new_of_old of the ordering (host copy; empty: natural)
    const std::vector<int32_t> &order() const { return order_host_; }

private:
    int n_ = 0, lev_f_ = 0, lev_b_ = 0, attempts_ = 0, epoch_ = 0;
    double shift_ = 0.0;
    bool ok_ = false;
    DeviceBuffer<int> fptr_, fcol_, bptr_, bcol_, order_f_, order_b_, flag_f_, flag_b_, ticket_;
    DeviceBuffer<double> fval_, bval_, dinv_, scale_, y_, w_;
    int ordering_ = 0;
    static constexpr int kLevelLaunchMax = 48; // up to this many dependency levels: one launch per level instead of the waiting kernel
    int64_t order_nnz_ = -1;
    unsigned long long order_id_ = 0;          // pattern id (Context::pattern_id_of_A) the kept order belongs to
    std::vector<int> start_f_, start_b_;       // first position of every level in order_f_ / order_b_
    std::vector<int32_t> order_host_;        // order[k] = row of the caller's matrix that became row k
    DeviceBuffer<int> perm_, iperm_;          // the same on the device, and its inverse
    DeviceBuffer<double> rp_, zp_;            // the residual / the result in the factor's numbering
};

} // namespace psolve
