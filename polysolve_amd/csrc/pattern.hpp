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
    // row kinds (PatDev::kind): pattern + values
    DeviceBuffer<unsigned short> kind;
    DeviceBuffer<double> kval, scoef;
    DeviceBuffer<unsigned> smask;
    DeviceBuffer<int> koff, klen, vrep, vslot_kid;
    DeviceBuffer<unsigned long long> vkeys;
    PatDev view;
    bool valid = false;

    // false (and no dictionary) when the operator has more than kPatMaxPatterns distinct patterns, a row longer
    // than kPatMaxLen, or column ids that are not sorted by row the same way everywhere -- the plain stream then
    bool build(const Launch &L, const CsrDev &A);
    // the kinds of the rows under a valid dictionary, from A's CURRENT values (every factorize); false -- and a view
    // without kinds -- when the rows repeat too little for the kinds to fit LDS
    // (same_pattern: A's pattern is the one the kinds were last built for -- they are verified in one pass before anything
    // is rebuilt)
    bool build_values(const Launch &L, const CsrDev &A, bool same_pattern = false);
    int kinds_n = 0;
    DeviceBuffer<int> krep; // a row of each kind
    // table[k] = v[a row of kind k], checked against EVERY row (v[r] == table[kind[r]] bit for bit): true when v is constant
    // within every kind.  Synchronises the stream.
    bool build_row_table(const Launch &L, int n, const double *v, DeviceBuffer<double> &table);
    void drop_values()
    {
        view.kind = nullptr;
        view.kval = nullptr;
        view.koff = nullptr;
        view.klen = nullptr;
        view.nkind = 0;
        view.kml = 0;
        view.scoef = nullptr;
        view.smask = nullptr;
        view.nslot = 0;
        view.sdiag = -1;
    }
    void reset()
    {
        valid = false;
        view = PatDev();
    }
};

// block-row kinds of a 3x3-block copy (Bsr3KindDev)
struct Bsr3Kinds {
    DeviceBuffer<unsigned short> kind, kblk;
    DeviceBuffer<int> koff, klen, rep, slot_kid, ctrl, krep;
    DeviceBuffer<double> kraw, blocks;
    DeviceBuffer<unsigned long long> keys, rowhash;
    PinnedBuffer<int> host;
    Bsr3KindDev view;
    bool valid = false;
    int built_nb = 0;
    // from B's CURRENT values; false when the block rows do not repeat (or the tables would not fit LDS).  same_pattern: B's
    // block pattern is the one of the previous build -- the previous kinds are verified in one pass before anything is rebuilt
    bool build(const Launch &L, const Bsr3Dev &B, bool same_pattern = false);
    void reset()
    {
        valid = false;
        view = Bsr3KindDev();
    }
};

} // namespace psolve
