// pattern.hpp -- device-built dictionary of the column-offset patterns of a CSR operator (see PatDev in kernels.hpp).
//
// Not a reference structure (the reference multiplies through cusparseSpMV / AMGCL's CSR loop); it is storage only:
// entry j of row r is still val[rowptr[r] + j] times x[r + off[id[r]][j]], the same column in the same order.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace psolve {

struct PatMatrix {
    DeviceBuffer<unsigned short> id;
    DeviceBuffer<int> off;
    DeviceBuffer<unsigned long long> keys; // hash table: pattern hash -> slot
    DeviceBuffer<int> rep, slot_pid, ctrl;
    PinnedBuffer<int> host;
    PatDev view;
    bool valid = false;

    // false (and no dictionary) when the operator has more than kPatMaxPatterns distinct patterns, a row longer
    // than kPatMaxLen, or column ids that are not sorted by row the same way everywhere -- the plain stream then
    bool build(const Launch &L, const CsrDev &A);
    void reset()
    {
        valid = false;
        view = PatDev();
    }
};

} // namespace psolve
