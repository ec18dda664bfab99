// sell.hpp -- device-built SELL-64-sigma copies of wide-row CSR operators (see SellDev in kernels.hpp).
//
// Not a reference structure: the reference multiplies through cusparseSpMV on CSR / BSR (MASSolver.cu:271-290) and
// AMGCL's builtin backend on CSR (amgcl/backend/builtin.hpp spmv_impl).  The products are the same numbers -- each
// row is summed in column order, as in those scalar loops -- only the storage order of the stream differs.
#pragma once
#include "amg_symbolic.hpp"
#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

struct SellMatrix {
    DeviceBuffer<int> slice_ptr, col;
    DeviceBuffer<double> val;
    DeviceBuffer<int2> slot;
    SellDev view;
    bool valid = false;
    int64_t padded = 0; // stored entries including the padding

    // structure + values from the CSR operator A.  Returns false (and stays invalid) when the padded copy would be
    // more than max_fill times the CSR entries or exceed int32 indexing.
    bool build(const Launch &L, const CsrDev &A, SymbolicScratch &S, double max_fill = 1.25);
    // values only (same pattern as at build(): the numeric refresh of a hierarchy, a refactorize of constant pattern)
    void refill(const Launch &L, const CsrDev &A);
    void reset()
    {
        valid = false;
        view = SellDev();
    }
};

} // namespace psolve
