// amg_dist.hip -- the distributed smoothed-aggregation hierarchy (see amg_dist.hpp for the design).
#include "amg_dist.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "amg_setup.hpp"
#include "solver.hpp"

namespace psolve {

namespace {

// ---- small kernels ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_i32_kernel(int n, const int *__restrict__ idx, const int *__restrict__ x,
                                                             int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = x[idx[i]];
}

// out[i] = id[i] >= 0 ? id[i] + shift : id[i]
__global__ __launch_bounds__(kBlock) void shift_ids_kernel(int n, const int *__restrict__ id, int shift, int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int a = id[i];
        out[i] = a >= 0 ? a + shift : a;
    }
}

__global__ __launch_bounds__(kBlock) void row_lengths_kernel(int n, const int *__restrict__ idx, const int *__restrict__ ptr,
                                                              int *__restrict__ len)
{
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) len[k] = ptr[idx[k] + 1] - ptr[idx[k]];
}

// rows idx[0..n) of (ptr, col, val) packed back to back at pk_ptr[k]
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(int n, const int *__restrict__ idx, const int *__restrict__ ptr,
                                                            const int *__restrict__ col, const double *__restrict__ val,
                                                            const int *__restrict__ pk_ptr, int *__restrict__ pk_col,
                                                            double *__restrict__ pk_val)
{
    const int lane = threadIdx.x & 15, g = (blockIdx.x * kBlock + threadIdx.x) >> 4, ng = (gridDim.x * kBlock) >> 4;
    for (int k = g; k < n; k += ng) {
        const int b = ptr[idx[k]], len = ptr[idx[k] + 1] - b, o = pk_ptr[k];
        for (int j = lane; j < len; j += 16) {
            pk_col[o + j] = col[b + j];
            pk_val[o + j] = val[b + j];
        }
    }
}

// the values of rows idx[0..n) packed back to back at pk_ptr[k] (numeric refresh: the pattern went over at the setup)
__global__ __launch_bounds__(kBlock) void pack_values_kernel(int n, const int *__restrict__ idx, const int *__restrict__ ptr,
                                                              const double *__restrict__ val, const int *__restrict__ pk_ptr,
                                                              double *__restrict__ pk_val)
{
    const int lane = threadIdx.x & 15, g = (blockIdx.x * kBlock + threadIdx.x) >> 4, ng = (gridDim.x * kBlock) >> 4;
    for (int k = g; k < n; k += ng) {
        const int b = ptr[idx[k]], len = ptr[idx[k] + 1] - b, o = pk_ptr[k];
        for (int j = lane; j < len; j += 16) pk_val[o + j] = val[b + j];
    }
}

// ptr_ext[i] = i <= n_loc ? ptr[i] : nnz_loc + hptr[i - n_loc]
__global__ __launch_bounds__(kBlock) void stack_ptr_kernel(int n_loc, int n_halo, const int *__restrict__ ptr,
                                                            const int *__restrict__ hptr, int nnz_loc, int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i <= n_loc + n_halo; i += gridDim.x * kBlock)
        out[i] = i <= n_loc ? ptr[i] : nnz_loc + hptr[i - n_loc];
}

// rows restricted to columns in [c0, c1), shifted by -c0
template <bool FILL>
__global__ __launch_bounds__(kBlock) void column_range_kernel(int n, int c0, int c1, const int *__restrict__ ptr,
                                                               const int *__restrict__ col, const double *__restrict__ val,
                                                               int *__restrict__ optr, int *__restrict__ ocol,
                                                               double *__restrict__ oval, int *__restrict__ osrc)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int w = FILL ? optr[i] : 0;
        for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
            const int c = col[j];
            if (c >= c0 && c < c1) {
                if (FILL) {
                    ocol[w] = c - c0;
                    oval[w] = val[j];
                    osrc[w] = j; // where the entry came from (the numeric refresh gathers through it)
                }
                ++w;
            }
        }
        if (!FILL) optr[i] = w;
    }
}

// graph rows restricted to columns < ncols; id0 = the aggregation sweep's start state on the restricted graph
template <bool FILL>
__global__ __launch_bounds__(kBlock) void graph_filter_kernel(int n, int ncols, const int *__restrict__ ptr,
                                                               const int *__restrict__ col, int *__restrict__ optr,
                                                               int *__restrict__ ocol, int *__restrict__ id0)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int w = FILL ? optr[i] : 0;
        bool any = false;
        for (int j = ptr[i]; j < ptr[i + 1]; ++j) {
            const int c = col[j];
            if (c < ncols) {
                if (FILL) ocol[w] = c;
                ++w;
                any = any || c != i;
            }
        }
        if (!FILL) optr[i] = w;
        else id0[i] = any ? -1 : -2;
    }
}

// out[i] = id[i / bs]   (node values spread over the node's scalar rows) / out[k] = in[k * bs]
__global__ __launch_bounds__(kBlock) void spread_nodes_kernel(int n, int bs, const int *__restrict__ id, int *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = id[i / bs];
}
__global__ __launch_bounds__(kBlock) void pick_nodes_kernel(int nn, int bs, const int *__restrict__ in, int *__restrict__ out)
{
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < nn; k += gridDim.x * kBlock) out[k] = in[k * bs];
}

struct DevCsrD { // an owned CSR matrix on the device
    DeviceBuffer<int> ptr, col;
    DeviceBuffer<double> val;
    CsrDev view;
    void set_view(int nrows, int ncols, int64_t nnz)
    {
        view = CsrDev();
        view.n = nrows;
        view.n_ext = ncols;
        view.nnz = nnz;
        view.rowptr = ptr.ptr;
        view.col = col.ptr;
        view.val = val.ptr;
        view.rows_per_block = spmv_rows_per_block(nrows ? (double)nnz / (double)nrows : 1.0);
    }
};

// ---- halo links ----------------------------------------------------------------------------------------------
// The column ids of `cols` (global ids of a column space partitioned by `offsets`) become local ids: own columns
// [0, n_local), halo columns n_local + position in the sorted list of off-rank ids that occur in any of the arrays.
// bs > 1: the halo consists of whole nodes (all bs scalar columns of a node), so that block views of the operators
// stay aligned.
void build_halo_link(Comm &comm, const Launch &L, const std::vector<int64_t> &offsets,
                     const std::vector<std::pair<int *, int64_t>> &cols, HaloLink &H, int bs)
{
    hipStream_t s = L.stream;
    const int W = comm.world(), me = comm.rank();
    const int c0 = (int)offsets[(size_t)me], c1 = (int)offsets[(size_t)me + 1];
    H.plan = HaloPlan();
    H.plan.rank = me;
    H.plan.world = W;
    H.plan.row_offsets = offsets;
    H.n_local = c1 - c0;
    DeviceBuffer<int> flags;
    flags.ensure(8);
    PS_HIP_CHECK(hipMemsetAsync(flags.ptr, 0, 8 * sizeof(int), s));
    for (auto &c : cols)
        if (c.second > 0) launch_offrange_count(L, c.second, c.first, c0, c1, flags.ptr);
    int n_off = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&n_off, flags.ptr, sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int32_t> off((size_t)n_off);
    if (n_off > 0) {
        DeviceBuffer<int> d_off;
        d_off.ensure((size_t)n_off);
        for (auto &c : cols)
            if (c.second > 0) launch_offrange_collect(L, c.second, c.first, c0, c1, d_off.ptr, flags.ptr + 1);
        PS_HIP_CHECK(hipMemcpyAsync(off.data(), d_off.ptr, (size_t)n_off * sizeof(int), hipMemcpyDeviceToHost, s));
        PS_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (bs > 1) {
        std::vector<int32_t> whole;
        whole.reserve(off.size() * (size_t)bs);
        for (int32_t g : off)
            for (int c = 0; c < bs; ++c) whole.push_back(g / bs * bs + c);
        off.swap(whole);
        n_off = (int)off.size();
    }
    plan_halo(me, W, offsets.data(), n_off, off.data(), H.plan.halo, H.plan.recv_counts);
    const int n_halo = (int)H.plan.halo.size();
    H.plan.recv_offsets.assign((size_t)W, 0);
    for (int q = 1; q < W; ++q) H.plan.recv_offsets[q] = H.plan.recv_offsets[q - 1] + H.plan.recv_counts[q - 1];
    H.halo_dev.ensure((size_t)n_halo + 1);
    if (n_halo)
        PS_HIP_CHECK(hipMemcpyAsync(H.halo_dev.ptr, H.plan.halo.data(), (size_t)n_halo * sizeof(int), hipMemcpyHostToDevice, s));
    for (auto &c : cols)
        if (c.second > 0) launch_remap_cols(L, c.second, c.first, c0, c1, H.n_local, H.halo_dev.ptr, n_halo);
    // who needs what from whom
    DeviceBuffer<int64_t> d_cnt;
    d_cnt.ensure((size_t)W * (W + 1));
    std::vector<int64_t> h_cnt((size_t)W * W);
    PS_HIP_CHECK(hipMemcpyAsync(d_cnt.ptr + (size_t)W * W, H.plan.recv_counts.data(), (size_t)W * sizeof(int64_t),
                                hipMemcpyHostToDevice, s));
    comm.allgather_i64(d_cnt.ptr + (size_t)W * W, d_cnt.ptr, W, s);
    PS_HIP_CHECK(hipMemcpyAsync(h_cnt.data(), d_cnt.ptr, (size_t)W * W * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    H.plan.send_counts.assign((size_t)W, 0);
    H.plan.send_offsets.assign((size_t)W, 0);
    for (int q = 0; q < W; ++q) H.plan.send_counts[q] = h_cnt[(size_t)q * W + me];
    for (int q = 1; q < W; ++q) H.plan.send_offsets[q] = H.plan.send_offsets[q - 1] + H.plan.send_counts[q - 1];
    H.plan.n_send = H.plan.send_offsets[W - 1] + H.plan.send_counts[W - 1];
    H.send_idx.ensure((size_t)H.plan.n_send + 1);
    H.send_buf.ensure((size_t)H.plan.n_send + 1);
    H.send_buf_i.ensure((size_t)H.plan.n_send + 1);
    comm.exchange_i32(H.halo_dev.ptr, H.plan.recv_counts, H.plan.recv_offsets, H.send_idx.ptr, H.plan.send_counts,
                      H.plan.send_offsets, s);
    launch_add_offset_i32(L, H.plan.n_send, H.send_idx.ptr, -c0);
    PS_HIP_CHECK(hipStreamSynchronize(s));
}

void exchange_halo(Comm &comm, const Launch &L, HaloLink &H, double *d_ext)
{
    if (comm.world() <= 1) return;
    launch_gather(L, (int)H.plan.n_send, H.send_idx.ptr, d_ext, H.send_buf.ptr);
    comm.exchange_f64(H.send_buf.ptr, H.plan.send_counts, H.plan.send_offsets, d_ext + H.n_local, H.plan.recv_counts,
                      H.plan.recv_offsets, L.stream);
}

void exchange_halo_i32(Comm &comm, const Launch &L, HaloLink &H, int *d_ext)
{
    if (comm.world() <= 1) return;
    if (H.plan.n_send > 0) {
        hipLaunchKernelGGL(gather_i32_kernel, dim3(L.grid), dim3(kBlock), 0, L.stream, (int)H.plan.n_send, H.send_idx.ptr,
                           d_ext, H.send_buf_i.ptr);
        PS_HIP_CHECK(hipGetLastError());
    }
    comm.exchange_i32(H.send_buf_i.ptr, H.plan.send_counts, H.plan.send_offsets, d_ext + H.n_local, H.plan.recv_counts,
                      H.plan.recv_offsets, L.stream);
}

// what a row exchange over a halo link looked like (kept for the numeric refresh: same rows, same lengths, new values)
struct RowExchange {
    DeviceBuffer<int> pack_ptr; // where the rows this rank sends start in the packed buffer
    DeviceBuffer<double> pk_val;
    std::vector<int64_t> esc, eso, erc, ero;
    int64_t n_pack = 0, n_recv = 0, nnz_loc = 0;
};

// The rows of the row-partitioned matrix M (local rows, any column ids) that belong to this rank's halo columns,
// fetched from their owners: out = M stacked on top of them (n_local + n_halo rows).
void stack_halo_rows(Comm &comm, const Launch &L, HaloLink &H, const CsrDev &M, DevCsrD &out, SymbolicScratch &S,
                     RowExchange *keep = nullptr)
{
    hipStream_t s = L.stream;
    const int W = comm.world(), n_send = (int)H.plan.n_send, n_halo = H.n_halo(), n_loc = M.n;
    PS_REQUIRE(n_loc == H.n_local, PSOLVE_HIP_EINVAL, "stack_halo_rows: matrix / halo link mismatch");
    DeviceBuffer<int> len_s, pk_col, hptr;
    DeviceBuffer<double> pk_val;
    len_s.ensure((size_t)n_send + 2);
    hptr.ensure((size_t)n_halo + 2);
    PS_HIP_CHECK(hipMemsetAsync(len_s.ptr, 0, ((size_t)n_send + 2) * sizeof(int), s));
    PS_HIP_CHECK(hipMemsetAsync(hptr.ptr, 0, ((size_t)n_halo + 2) * sizeof(int), s));
    if (n_send > 0) {
        hipLaunchKernelGGL(row_lengths_kernel, dim3(L.grid), dim3(kBlock), 0, s, n_send, H.send_idx.ptr, M.rowptr, len_s.ptr);
        PS_HIP_CHECK(hipGetLastError());
    }
    comm.exchange_i32(len_s.ptr, H.plan.send_counts, H.plan.send_offsets, hptr.ptr, H.plan.recv_counts, H.plan.recv_offsets, s);
    std::vector<int> h_ls((size_t)n_send + 1, 0), h_lr((size_t)n_halo + 1, 0);
    if (n_send) PS_HIP_CHECK(hipMemcpyAsync(h_ls.data(), len_s.ptr, (size_t)n_send * sizeof(int), hipMemcpyDeviceToHost, s));
    if (n_halo) PS_HIP_CHECK(hipMemcpyAsync(h_lr.data(), hptr.ptr, (size_t)n_halo * sizeof(int), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int64_t> esc((size_t)W, 0), eso((size_t)W, 0), erc((size_t)W, 0), ero((size_t)W, 0);
    for (int q = 0; q < W; ++q) {
        for (int64_t k = H.plan.send_offsets[q]; k < H.plan.send_offsets[q] + H.plan.send_counts[q]; ++k) esc[q] += h_ls[(size_t)k];
        for (int64_t k = H.plan.recv_offsets[q]; k < H.plan.recv_offsets[q] + H.plan.recv_counts[q]; ++k) erc[q] += h_lr[(size_t)k];
    }
    for (int q = 1; q < W; ++q) {
        eso[q] = eso[q - 1] + esc[q - 1];
        ero[q] = ero[q - 1] + erc[q - 1];
    }
    const int64_t n_pack = eso[W - 1] + esc[W - 1], n_recv = ero[W - 1] + erc[W - 1];
    PS_REQUIRE(M.nnz + n_recv < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
    pk_col.ensure((size_t)n_pack + 4);
    pk_val.ensure((size_t)n_pack + 4);
    if (n_send > 0) {
        const int64_t tot = device_exclusive_scan(L, len_s.ptr, n_send, S); // len_s becomes the pack offsets
        PS_REQUIRE(tot == n_pack, PSOLVE_HIP_ECOMM, "stack_halo_rows: pack sizes disagree");
        hipLaunchKernelGGL(pack_rows_kernel, dim3(L.grid), dim3(kBlock), 0, s, n_send, H.send_idx.ptr, M.rowptr, M.col, M.val,
                           len_s.ptr, pk_col.ptr, pk_val.ptr);
        PS_HIP_CHECK(hipGetLastError());
    }
    out.ptr.ensure((size_t)n_loc + n_halo + 2);
    out.col.ensure((size_t)(M.nnz + n_recv) + 4);
    out.val.ensure((size_t)(M.nnz + n_recv) + 4);
    if (M.nnz) {
        PS_HIP_CHECK(hipMemcpyAsync(out.col.ptr, M.col, (size_t)M.nnz * sizeof(int), hipMemcpyDeviceToDevice, s));
        PS_HIP_CHECK(hipMemcpyAsync(out.val.ptr, M.val, (size_t)M.nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    comm.exchange_i32(pk_col.ptr, esc, eso, out.col.ptr + M.nnz, erc, ero, s);
    comm.exchange_f64(pk_val.ptr, esc, eso, out.val.ptr + M.nnz, erc, ero, s);
    if (n_halo > 0) {
        const int64_t tot = device_exclusive_scan(L, hptr.ptr, n_halo, S);
        PS_REQUIRE(tot == n_recv, PSOLVE_HIP_ECOMM, "stack_halo_rows: received sizes disagree");
    }
    hipLaunchKernelGGL(stack_ptr_kernel, dim3(L.grid), dim3(kBlock), 0, s, n_loc, n_halo, M.rowptr, hptr.ptr, (int)M.nnz,
                       out.ptr.ptr);
    PS_HIP_CHECK(hipGetLastError());
    out.set_view(n_loc + n_halo, M.n_ext, M.nnz + n_recv);
    PS_HIP_CHECK(hipStreamSynchronize(s)); // the scratch buffers of this frame are in use until here
    if (keep) {
        keep->pack_ptr.swap(len_s);
        keep->pk_val.ensure((size_t)n_pack + 4);
        keep->esc = esc;
        keep->eso = eso;
        keep->erc = erc;
        keep->ero = ero;
        keep->n_pack = n_pack;
        keep->n_recv = n_recv;
        keep->nnz_loc = M.nnz;
    }
}

// numeric refresh of a stacked matrix: the local values copied, those of the halo rows fetched again
void restack_values(Comm &comm, const Launch &L, HaloLink &H, RowExchange &X, const CsrDev &M, double *out_val)
{
    hipStream_t s = L.stream;
    const int n_send = (int)H.plan.n_send;
    PS_REQUIRE(M.nnz == X.nnz_loc, PSOLVE_HIP_EINVAL, "restack_values: the pattern changed");
    if (M.nnz && M.val != out_val) // (A P lives in the head of its stacked copy: nothing to copy there)
        PS_HIP_CHECK(hipMemcpyAsync(out_val, M.val, (size_t)M.nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (n_send > 0) {
        hipLaunchKernelGGL(pack_values_kernel, dim3(L.grid), dim3(kBlock), 0, s, n_send, H.send_idx.ptr, M.rowptr, M.val,
                           X.pack_ptr.ptr, X.pk_val.ptr);
        PS_HIP_CHECK(hipGetLastError());
    }
    comm.exchange_f64(X.pk_val.ptr, X.esc, X.eso, out_val + M.nnz, X.erc, X.ero, s);
}

// every rank's rows (global column ids) of a row-partitioned matrix, assembled on every rank
void gather_rows(Comm &comm, const Launch &L, const std::vector<int64_t> &offsets, const CsrDev &M, DevCsrD &G)
{
    hipStream_t s = L.stream;
    const int W = comm.world(), me = comm.rank();
    const int64_t n_glob = offsets[(size_t)W];
    DeviceBuffer<int64_t> d_cnt;
    d_cnt.ensure((size_t)W + 1);
    std::vector<int64_t> cnt((size_t)W), off((size_t)W + 1, 0);
    const int64_t mine = M.nnz;
    PS_HIP_CHECK(hipMemcpyAsync(d_cnt.ptr + W, &mine, sizeof(int64_t), hipMemcpyHostToDevice, s));
    comm.allgather_i64(d_cnt.ptr + W, d_cnt.ptr, 1, s);
    PS_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.ptr, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    for (int q = 0; q < W; ++q) off[(size_t)q + 1] = off[(size_t)q] + cnt[(size_t)q];
    const int64_t gnnz = off[(size_t)W];
    PS_REQUIRE(gnnz < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE, "replicated AMG level exceeds int32 indexing");
    G.ptr.ensure((size_t)n_glob + 1);
    G.col.ensure((size_t)gnnz + 4);
    G.val.ensure((size_t)gnnz + 4);
    const int64_t r0 = offsets[(size_t)me];
    PS_HIP_CHECK(hipMemcpyAsync(G.ptr.ptr + r0, M.rowptr, (size_t)M.n * sizeof(int), hipMemcpyDeviceToDevice, s));
    launch_add_offset_i32(L, M.n, G.ptr.ptr + r0, (int)off[(size_t)me]);
    if (M.nnz) {
        PS_HIP_CHECK(hipMemcpyAsync(G.col.ptr + off[(size_t)me], M.col, (size_t)M.nnz * sizeof(int), hipMemcpyDeviceToDevice, s));
        PS_HIP_CHECK(hipMemcpyAsync(G.val.ptr + off[(size_t)me], M.val, (size_t)M.nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    const int last = (int)gnnz;
    PS_HIP_CHECK(hipMemcpyAsync(G.ptr.ptr + n_glob, &last, sizeof(int), hipMemcpyHostToDevice, s));
    std::vector<int64_t> sc((size_t)W, 0), so((size_t)W, 0), rc((size_t)W, 0), ro((size_t)W, 0);
    for (int q = 0; q < W; ++q) {
        if (q == me) continue;
        sc[(size_t)q] = M.n;
        so[(size_t)q] = r0;
        rc[(size_t)q] = offsets[(size_t)q + 1] - offsets[(size_t)q];
        ro[(size_t)q] = offsets[(size_t)q];
    }
    comm.exchange_i32(G.ptr.ptr, sc, so, G.ptr.ptr, rc, ro, s);
    for (int q = 0; q < W; ++q) {
        if (q == me) continue;
        sc[(size_t)q] = M.nnz;
        so[(size_t)q] = off[(size_t)me];
        rc[(size_t)q] = cnt[(size_t)q];
        ro[(size_t)q] = off[(size_t)q];
    }
    comm.exchange_i32(G.col.ptr, sc, so, G.col.ptr, rc, ro, s);
    comm.exchange_f64(G.val.ptr, sc, so, G.val.ptr, rc, ro, s);
    PS_HIP_CHECK(hipStreamSynchronize(s));
    G.set_view((int)n_glob, (int)n_glob, gnnz);
}

double allreduce_max(Comm &comm, hipStream_t s, double v)
{
    const int W = comm.world();
    if (W <= 1) return v;
    DeviceBuffer<int64_t> d;
    d.ensure((size_t)W + 1);
    std::vector<int64_t> h((size_t)W);
    int64_t bits;
    std::memcpy(&bits, &v, sizeof(bits));
    PS_HIP_CHECK(hipMemcpyAsync(d.ptr + W, &bits, sizeof(int64_t), hipMemcpyHostToDevice, s));
    comm.allgather_i64(d.ptr + W, d.ptr, 1, s);
    PS_HIP_CHECK(hipMemcpyAsync(h.data(), d.ptr, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    double m = v;
    for (int q = 0; q < W; ++q) {
        double x;
        std::memcpy(&x, &h[(size_t)q], sizeof(x));
        m = std::max(m, x);
    }
    return m;
}

std::vector<int64_t> allgather_counts(Comm &comm, hipStream_t s, int64_t mine)
{
    const int W = comm.world();
    std::vector<int64_t> h((size_t)W, mine);
    if (W <= 1) return h;
    DeviceBuffer<int64_t> d;
    d.ensure((size_t)W + 1);
    PS_HIP_CHECK(hipMemcpyAsync(d.ptr + W, &mine, sizeof(int64_t), hipMemcpyHostToDevice, s));
    comm.allgather_i64(d.ptr + W, d.ptr, 1, s);
    PS_HIP_CHECK(hipMemcpyAsync(h.data(), d.ptr, (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    return h;
}

struct DLevel {
    int n = 0, n_ext = 0;            // local rows; + halo columns
    std::vector<int64_t> offsets;    // partition of this level's rows over the ranks
    CsrDev A;                        // local rows, columns = local rows + halo (level 0: the solver's shard)
    DevCsrD A_own;
    HaloLink link;                   // the level's halo: of A and of the prolongation that maps onto this level
    DevCsrD P, R;                    // P: n x (next level's local + halo columns, or GLOBAL ids in front of the replicated tail)
                                     // R: (coarse nodes this rank owns) x (this level's local + halo columns)
    bool has_next = false;
    // kept for the numeric refresh (same pattern, new values): the stacked P and A P with their row exchanges, the
    // aggregate ids (global numbering, halo included), where R's values sit in the stacked P, global-id copies of the
    // column arrays that were renumbered for the cycle, the block graph and block P of block value types
    DevCsrD Pext, AP, APext;
    RowExchange xP, xAP;
    DeviceBuffer<int> id_ext, rmap, pcol_glob, accol_glob, pbptr, pbcol;
    DeviceBuffer<double> pbval;
    BlockGraph Gf;
    unsigned long long nz_hash = 0; // which stored entries were nonzero when the strength graph was taken (scalar)
    int64_t pbnnz = 0;
    DeviceBuffer<double> dinv, dinv_blk, f, x_ext, xb_ext, t_ext, p;
    double rho = 0, d = 0, c = 0;
    Launch L;
};

} // namespace

struct DistAmg::Impl {
    std::vector<std::unique_ptr<DLevel>> lv;
    AmgParams prm;
    std::unique_ptr<AmgHierarchy> tail; // the replicated rest of the hierarchy (its level 0 = the gathered level)
    std::vector<int64_t> tail_offsets;  // partition of the tail's level 0 (= coarse partition of the last distributed level)
    DevCsrD tail_A;                     // the gathered level (the tail's level 0 aliases these arrays)
    std::unique_ptr<DLevel> tail_src;   // this rank's rows of the gathered level (what the refresh recomputes and gathers again)
    // identity of the pattern the hierarchy was built for
    bool symbolic_valid = false, reused = false;
    unsigned long long pattern_hash = 0;
    int pattern_n = 0, pattern_next = 0;
    int64_t pattern_nnz = 0;
    AmgParams built_prm;
    DeviceBuffer<unsigned long long> hash_dev;
    DeviceBuffer<double> tail_f, tail_u;
    SymbolicScratch sym;
    AggregateScratch agg;
    DeviceBuffer<double> partials, red;
    PinnedBuffer<double> host;
};

DistAmg::DistAmg() : impl(new Impl()) {}
DistAmg::~DistAmg() = default;
int DistAmg::distributed_levels() const { return (int)impl->lv.size(); }
bool DistAmg::last_setup_reused() const { return impl->reused; }
int DistAmg::levels() const { return (int)impl->lv.size() + (impl->tail ? impl->tail->levels() : 0); }

void DistAmg::level_shape(int l, int64_t *rows_global, int64_t *rows_local, int64_t *nnz_local, double *rho) const
{
    const int nd = (int)impl->lv.size();
    PS_REQUIRE(l >= 0 && l < levels(), PSOLVE_HIP_EINVAL, "amg_level_info: no such level");
    if (l < nd) {
        const DLevel &lv = *impl->lv[(size_t)l];
        if (rows_global) *rows_global = lv.offsets.back();
        if (rows_local) *rows_local = lv.n;
        if (nnz_local) *nnz_local = lv.A.nnz;
        if (rho) *rho = lv.rho;
        return;
    }
    int64_t r = 0, z = 0;
    double rh = 0;
    impl->tail->level_shape(l - nd, &r, &z, &rh);
    if (rows_global) *rows_global = r;
    if (rows_local) *rows_local = r;
    if (nnz_local) *nnz_local = z;
    if (rho) *rho = rh;
}

// rho(D^-1 A_l) by power iterations on the partitioned operator: the halo of the iterate travels before every product,
// the two sums of an iteration are all-reduced together.  Start vector: U(-1, 1) by global row index (stateless, so the
// estimate does not depend on the number of ranks).
static double dist_spectral_radius(Context &ctx, DistAmg::Impl &I, DLevel &lv, int iters)
{
    Comm &comm = ctx.comm();
    const Launch &L = lv.L;
    hipStream_t s = L.stream;
    double *part = I.partials.ptr, *red = I.red.ptr;
    const int bs = I.prm.block_size > 1 ? I.prm.block_size : 1;
    if (bs > 1) { // constant per node, as the single-device block power iteration starts
        launch_splitmix(L, lv.n / bs, 0x5eedull, lv.offsets[(size_t)comm.rank()] / bs, lv.t_ext.ptr);
        launch_scale_expand(L, lv.n, bs, 1.0, lv.t_ext.ptr, lv.xb_ext.ptr);
    } else {
        launch_splitmix(L, lv.n, 0x5eedull, lv.offsets[(size_t)comm.rank()], lv.xb_ext.ptr);
    }
    launch_dot(L, lv.n, lv.xb_ext.ptr, lv.xb_ext.ptr, part);
    launch_sum_partials(L, part, L.grid, kMaxPartials, red, 1);
    comm.allreduce_sum(red, 1, s);
    launch_scale_by_norm(L, lv.n, red, 1, lv.xb_ext.ptr, lv.xb_ext.ptr);
    SpmvExtra ex;
    ex.dinv = lv.dinv.ptr;
    ex.partials2 = part + kMaxPartials;
    for (int it = 0; it < iters; ++it) {
        exchange_halo(comm, L, lv.link, lv.xb_ext.ptr);
        if (bs > 1) {
            launch_spmv(L, lv.A, SPMV_PLAIN, lv.xb_ext.ptr, nullptr, lv.t_ext.ptr, nullptr, nullptr);
            launch_block_power(L, lv.n, bs, lv.dinv_blk.ptr, lv.t_ext.ptr, lv.xb_ext.ptr, part, part + kMaxPartials);
            launch_sum_partials(L, part, L.grid, kMaxPartials, red, 2);
        } else {
            launch_spmv(L, lv.A, SPMV_POWER, lv.xb_ext.ptr, nullptr, lv.t_ext.ptr, part, nullptr, &ex);
            launch_sum_partials(L, part, L.spmv_grid, kMaxPartials, red, 2); // sum s^2, sum |s x| (adjacent arrays)
        }
        comm.allreduce_sum(red, 2, s);
        if (it + 1 < iters) launch_scale_by_norm(L, lv.n, red, 1, lv.t_ext.ptr, lv.xb_ext.ptr);
    }
    PS_HIP_CHECK(hipMemcpyAsync(I.host.ptr, red + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    PS_HIP_CHECK(hipStreamSynchronize(s));
    return I.host.ptr[0];
}

static void dist_smoother(Context &ctx, DistAmg::Impl &I, DLevel &lv)
{
    const AmgParams &prm = I.prm;
    Comm &comm = ctx.comm();
    lv.L = fit_launch(ctx.launch_max(), lv.n, lv.A.rows_per_block, lv.n > 0 ? (double)lv.A.nnz / lv.n : 0.0);
    lv.L.stream = ctx.stream;
    if (lv.A.pat == nullptr && lv.L.spmv_kernel == 3) lv.L.spmv_kernel = -1;
    const Launch &L = lv.L;
    const size_t n = (size_t)lv.n, ne = (size_t)lv.n_ext;
    lv.dinv.ensure(n + 1);
    lv.x_ext.ensure(ne + 2);
    lv.xb_ext.ensure(ne + 2);
    lv.t_ext.ensure(ne + 2);
    lv.p.ensure(n + 2);
    lv.f.ensure(n + 2);
    DeviceBuffer<int> bad;
    bad.ensure(2);
    PS_HIP_CHECK(hipMemsetAsync(bad.ptr, 0, 2 * sizeof(int), L.stream));
    launch_diag_inverse(L, lv.A, lv.dinv.ptr, bad.ptr);
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    if (bs > 1) {
        lv.dinv_blk.ensure((size_t)(lv.n / bs) * bs * bs + 4);
        launch_block_diag_inverse(L, lv.A, bs, lv.dinv_blk.ptr, bad.ptr + 1);
        int nbad = 0;
        PS_HIP_CHECK(hipMemcpyAsync(&nbad, bad.ptr + 1, sizeof(int), hipMemcpyDeviceToHost, L.stream));
        PS_HIP_CHECK(hipStreamSynchronize(L.stream));
        PS_REQUIRE(nbad == 0, PSOLVE_HIP_ENUMERIC, "AMG: singular diagonal block");
        PS_REQUIRE(prm.cheb_power_iters > 0, PSOLVE_HIP_EINVAL, "amg.cheb_power_iters = 0 (Gershgorin) is scalar-only in this build");
    }
    double hi;
    if (prm.cheb_power_iters > 0) {
        hi = dist_spectral_radius(ctx, I, lv, prm.cheb_power_iters);
    } else {
        launch_gershgorin(L, lv.A, I.partials.ptr);
        std::vector<double> h((size_t)L.grid);
        PS_HIP_CHECK(hipMemcpyAsync(h.data(), I.partials.ptr, h.size() * sizeof(double), hipMemcpyDeviceToHost, L.stream));
        PS_HIP_CHECK(hipStreamSynchronize(L.stream));
        double m = 0.0;
        for (double v : h) m = std::max(m, v);
        hi = allreduce_max(comm, L.stream, m);
    }
    if (!(hi > 0) || !std::isfinite(hi)) hi = 2.0;
    lv.rho = hi;
    const double lo = hi * prm.cheb_lower;
    hi *= prm.cheb_higher;
    lv.d = 0.5 * (hi + lo);
    lv.c = 0.5 * (hi - lo);
}

static void dist_full_setup(Context &ctx, DistAmg::Impl &I)
{
    const AmgParams &prm = I.prm;
    Comm &comm = ctx.comm();
    const int W = comm.world(), me = comm.rank();
    hipStream_t s = ctx.stream;
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    PS_REQUIRE(prm.eps_strong == 0.0, PSOLVE_HIP_EINVAL, "the distributed AMG setup needs amg.eps_strong = 0");
    const bool timing = std::getenv("PSOLVE_TIMING") != nullptr;
    I.lv.clear();
    I.tail.reset();
    I.tail_src.reset();
    I.partials.ensure(2 * (size_t)kMaxPartials);
    I.red.ensure(8);
    I.host.ensure(8);

    std::unique_ptr<DLevel> cur(new DLevel());
    cur->A = ctx.A;
    cur->A.bsr3 = nullptr;
    cur->A.sell = nullptr;
    cur->n = ctx.A.n;
    cur->n_ext = ctx.A.n_ext;
    ctx.export_halo_link(cur->link);
    cur->offsets = cur->link.plan.row_offsets;
    const int64_t replicate_below = std::max<int64_t>(prm.coarse_enough, (int64_t)prm.dist_replicate_rows * W);

    DeviceBuffer<int> sptr_f, scol_f, id0_f, sptr, scol, id0, id_loc, id_s;
    DeviceBuffer<double> dia;
    std::vector<int32_t> h_sptr, h_scol, h_id;
    while (true) {
        DLevel &lv = *cur;
        const int64_t n_glob = lv.offsets.back();
        const int nl = (int)I.lv.size();
        if (timing && me == 0)
            std::fprintf(stderr, "[psolve timing] dist amg level %d: %lld rows global, %d local (+%d halo)\n", nl,
                         (long long)n_glob, lv.n, lv.n_ext - lv.n);
        if (n_glob <= prm.coarse_enough || nl + 1 >= prm.max_levels) break;
        if (nl >= 1 && n_glob <= replicate_below) {
            // small enough: every rank takes the whole level and the rest of the hierarchy is the single-device one
            DevCsrD Ag;
            Launch L = fit_launch(ctx.launch_max(), lv.n, lv.A.rows_per_block);
            L.stream = s;
            // (the level's columns were kept GLOBAL for this: see `next_replicated` below)
            gather_rows(comm, L, lv.offsets, lv.A, Ag);
            AmgParams tp = prm;
            tp.max_levels = std::max(1, prm.max_levels - nl);
            tp.renumber = 0;
            I.tail.reset(new AmgHierarchy());
            // the gathered arrays must outlive the hierarchy (its level 0 aliases them)
            I.tail_A.ptr.swap(Ag.ptr);
            I.tail_A.col.swap(Ag.col);
            I.tail_A.val.swap(Ag.val);
            I.tail_A.set_view(Ag.view.n, Ag.view.n_ext, Ag.view.nnz);
            I.tail->setup(ctx, I.tail_A.view, tp);
            I.tail_offsets = lv.offsets;
            I.tail_f.ensure((size_t)n_glob + 2);
            I.tail_u.ensure((size_t)n_glob + 2);
            I.tail_src = std::move(cur); // its rows live on in the gathered copy; the refresh recomputes and gathers them again
            break;
        }
        Launch L = fit_setup_launch(ctx.launch_max(), lv.n, lv.A.nnz, lv.A.rows_per_block);
        L.stream = s;
        const int n = lv.n, n_ext = lv.n_ext;
        // -- strength: the full graph (with halo columns) shapes P; its restriction to the shard shapes the aggregates.
        //    Block value types (AMGCL_Block<3>): the graph is the node graph of the b x b block view of A.
        PS_REQUIRE(n % bs == 0 && n_ext % bs == 0, PSOLVE_HIP_EINVAL, "AMG: level size / halo is not a multiple of block_size");
        const int ng = n / bs, ng_ext = n_ext / bs; // nodes of the strength graph (local, local + halo)
        BlockGraph &Gf = lv.Gf;
        DeviceBuffer<int> &id_ext = lv.id_ext, &pbptr = lv.pbptr, &pbcol = lv.pbcol;
        DeviceBuffer<double> &pbval = lv.pbval;
        id0_f.ensure((size_t)ng + 1);
        if (bs > 1) {
            Gf.didx.ensure((size_t)ng_ext + 2); // (the strength test looks up the diagonal block of every column: none for halo nodes)
            PS_HIP_CHECK(hipMemsetAsync(Gf.didx.ptr, 0xff, ((size_t)ng_ext + 2) * sizeof(int), s));
            device_block_graph(L, lv.A, bs, Gf, I.sym);
            device_block_values(L, lv.A, Gf);
            device_block_strength_graph(L, Gf, 0.0, sptr_f, scol_f, id0_f.ptr, I.sym);
        } else {
            dia.ensure((size_t)n + 1);
            launch_extract_diagonal(L, lv.A, dia.ptr);
            device_strength_graph(L, lv.A, 0.0, dia.ptr, sptr_f, scol_f, id0_f.ptr, I.sym);
            I.hash_dev.ensure(4);
            PS_HIP_CHECK(hipMemsetAsync(I.hash_dev.ptr, 0, sizeof(unsigned long long), s));
            launch_hash_nonzero(L, lv.A.nnz, lv.A.val, I.hash_dev.ptr);
            PS_HIP_CHECK(hipMemcpyAsync(&lv.nz_hash, I.hash_dev.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
        }
        sptr.ensure((size_t)ng + 2);
        id0.ensure((size_t)ng + 1);
        hipLaunchKernelGGL(graph_filter_kernel<false>, dim3(L.grid), dim3(kBlock), 0, s, ng, ng, sptr_f.ptr, scol_f.ptr, sptr.ptr,
                           (int *)nullptr, (int *)nullptr);
        PS_HIP_CHECK(hipGetLastError());
        const int64_t snnz = device_exclusive_scan(L, sptr.ptr, ng, I.sym);
        scol.ensure((size_t)snnz + 4);
        hipLaunchKernelGGL(graph_filter_kernel<true>, dim3(L.grid), dim3(kBlock), 0, s, ng, ng, sptr_f.ptr, scol_f.ptr, sptr.ptr,
                           scol.ptr, id0.ptr);
        PS_HIP_CHECK(hipGetLastError());
        id_loc.ensure((size_t)ng + 1);
        int64_t nagg = -1;
        if (prm.device_aggregation && ng >= prm.aggregation_min_rows) {
            int rounds = 0;
            nagg = device_aggregate(L, ng, sptr.ptr, scol.ptr, id0.ptr, id_loc.ptr, prm.aggregation_max_rounds, I.agg, I.sym,
                                    &rounds, prm.aggregation_rounds ? 1 : 2);
        }
        if (nagg < 0) {
            h_sptr.resize((size_t)ng + 1);
            h_scol.resize((size_t)snnz + 1);
            h_id.resize((size_t)ng);
            PS_HIP_CHECK(hipMemcpyAsync(h_sptr.data(), sptr.ptr, ((size_t)ng + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
            if (snnz) PS_HIP_CHECK(hipMemcpyAsync(h_scol.data(), scol.ptr, (size_t)snnz * sizeof(int), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipMemcpyAsync(h_id.data(), id0.ptr, (size_t)ng * sizeof(int), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
            nagg = aggregate_strength_graph(ng, h_sptr.data(), h_scol.data(), h_id, true);
            PS_HIP_CHECK(hipMemcpyAsync(id_loc.ptr, h_id.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
        }
        // -- coarse numbering: rank after rank
        const std::vector<int64_t> cnt = allgather_counts(comm, s, nagg);
        std::vector<int64_t> coff((size_t)W + 1, 0);
        bool empty_rank = false;
        for (int q = 0; q < W; ++q) {
            coff[(size_t)q + 1] = coff[(size_t)q] + cnt[(size_t)q];
            empty_rank = empty_rank || cnt[(size_t)q] <= 0;
        }
        const int64_t nc_glob = coff[(size_t)W] * bs; // scalar size of the coarse level
        if (empty_rank) break; // a shard without aggregates (diagonal block): the level stays the coarsest
        PS_REQUIRE(nc_glob < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
        const int nc_loc = (int)nagg * bs, c0 = (int)coff[(size_t)me] * bs;
        for (auto &v : coff) v *= bs; // from here on: the partition of the coarse SCALAR rows
        // aggregate (node) ids of the local nodes, global numbering; those of the halo nodes come from their owners
        id_ext.ensure((size_t)ng_ext + 2);
        hipLaunchKernelGGL(shift_ids_kernel, dim3(L.grid), dim3(kBlock), 0, s, ng, id_loc.ptr, c0 / bs, id_ext.ptr);
        PS_HIP_CHECK(hipGetLastError());
        if (bs > 1) { // the halo link is one of scalar columns: spread the node ids over the scalar rows and back
            id_s.ensure((size_t)n_ext + 2);
            hipLaunchKernelGGL(spread_nodes_kernel, dim3(L.grid), dim3(kBlock), 0, s, n, bs, id_ext.ptr, id_s.ptr);
            PS_HIP_CHECK(hipGetLastError());
            exchange_halo_i32(comm, L, lv.link, id_s.ptr);
            hipLaunchKernelGGL(pick_nodes_kernel, dim3(L.grid), dim3(kBlock), 0, s, ng_ext, bs, id_s.ptr, id_ext.ptr);
            PS_HIP_CHECK(hipGetLastError());
        } else {
            exchange_halo_i32(comm, L, lv.link, id_ext.ptr);
        }
        // -- P = (I - omega D^-1 A_F) P_tent on the local rows, global coarse column ids
        int64_t pnnz;
        if (bs > 1) {
            const double gersh = allreduce_max(comm, s, device_block_gershgorin(L, Gf, I.partials.ptr));
            const double omega = prm.sa_relax * (prm.estimate_spectral_radius ? (4.0 / 3.0) / gersh : 2.0 / 3.0);
            const int64_t pbnnz = device_spgemm_symbolic(L, ng, sptr_f.ptr, scol_f.ptr, nullptr, id_ext.ptr, (int)(nc_glob / bs),
                                                         pbptr, pbcol, I.sym);
            lv.pbnnz = pbnnz;
            pbval.ensure((size_t)pbnnz * bs * bs + 4);
            launch_block_prolongation_values(L, Gf, id_ext.ptr, omega, pbptr.ptr, pbcol.ptr, pbval.ptr);
            pnnz = pbnnz * bs * bs;
            PS_REQUIRE(pnnz < (int64_t)INT32_MAX - 1024, PSOLVE_HIP_ERANGE, "AMG level exceeds int32 indexing");
            lv.P.ptr.ensure((size_t)n + 1);
            lv.P.col.ensure((size_t)pnnz + 4);
            lv.P.val.ensure((size_t)pnnz + 4);
            launch_expand_block_csr(L, ng, bs, pbptr.ptr, pbcol.ptr, pbval.ptr, lv.P.ptr.ptr, lv.P.col.ptr, lv.P.val.ptr);
            lv.P.set_view(n, (int)nc_glob, pnnz);
        } else {
            launch_gershgorin(L, lv.A, I.partials.ptr);
            std::vector<double> hg((size_t)L.grid);
            PS_HIP_CHECK(hipMemcpyAsync(hg.data(), I.partials.ptr, hg.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
            double gersh = 0.0;
            for (double v : hg) gersh = std::max(gersh, v);
            gersh = allreduce_max(comm, s, gersh);
            const double omega = prm.sa_relax * (prm.estimate_spectral_radius ? (4.0 / 3.0) / gersh : 2.0 / 3.0);
            pnnz = device_spgemm_symbolic(L, n, sptr_f.ptr, scol_f.ptr, nullptr, id_ext.ptr, (int)nc_glob, lv.P.ptr, lv.P.col,
                                          I.sym);
            lv.P.val.ensure((size_t)pnnz + 4);
            lv.P.set_view(n, (int)nc_glob, pnnz);
            CsrMut Pm{n, lv.P.ptr.ptr, lv.P.col.ptr, lv.P.val.ptr};
            launch_prolongation_values(L, lv.A, id_ext.ptr, omega, nullptr, 0.0, Pm);
        }
        // -- A P on the local rows: needs the rows of P of the halo columns of A
        DevCsrD Pf;
        DevCsrD &Pext = lv.Pext, &AP = lv.AP, &APext = lv.APext;
        stack_halo_rows(comm, L, lv.link, lv.P.view, Pext, I.sym, &lv.xP);
        const int64_t apnnz = device_spgemm_symbolic(L, n, lv.A.rowptr, lv.A.col, Pext.ptr.ptr, Pext.col.ptr, (int)nc_glob,
                                                     AP.ptr, AP.col, I.sym);
        AP.val.ensure((size_t)apnnz + 4);
        AP.set_view(n, (int)nc_glob, apnnz);
        CsrMut APm{n, AP.ptr.ptr, AP.col.ptr, AP.val.ptr};
        launch_spgemm_numeric(L, APm, lv.A, Pext.view, (double)apnnz / std::max(1, n));
        // -- R: the rows of P^T this rank owns = transpose of [P ; halo rows of P] restricted to its coarse columns
        DeviceBuffer<int> pf_src, r_from_p;
        Pf.ptr.ensure((size_t)n_ext + 2);
        hipLaunchKernelGGL(column_range_kernel<false>, dim3(L.grid), dim3(kBlock), 0, s, n_ext, c0, c0 + nc_loc,
                           Pext.ptr.ptr, Pext.col.ptr, Pext.val.ptr, Pf.ptr.ptr, (int *)nullptr, (double *)nullptr,
                           (int *)nullptr);
        PS_HIP_CHECK(hipGetLastError());
        const int64_t pfnnz = device_exclusive_scan(L, Pf.ptr.ptr, n_ext, I.sym);
        Pf.col.ensure((size_t)pfnnz + 4);
        Pf.val.ensure((size_t)pfnnz + 4);
        pf_src.ensure((size_t)pfnnz + 4);
        hipLaunchKernelGGL(column_range_kernel<true>, dim3(L.grid), dim3(kBlock), 0, s, n_ext, c0, c0 + nc_loc, Pext.ptr.ptr,
                           Pext.col.ptr, Pext.val.ptr, Pf.ptr.ptr, Pf.col.ptr, Pf.val.ptr, pf_src.ptr);
        PS_HIP_CHECK(hipGetLastError());
        device_transpose_pattern(L, n_ext, nc_loc, Pf.ptr.ptr, Pf.col.ptr, pfnnz, lv.R.ptr, lv.R.col, r_from_p, I.sym);
        lv.R.val.ensure((size_t)pfnnz + 4);
        lv.R.set_view(nc_loc, n_ext, pfnnz);
        lv.rmap.ensure((size_t)pfnnz + 4); // R's entry k = entry rmap[k] of the stacked P
        if (pfnnz > 0) {
            hipLaunchKernelGGL(gather_i32_kernel, dim3(L.grid), dim3(kBlock), 0, s, (int)pfnnz, r_from_p.ptr, pf_src.ptr,
                               lv.rmap.ptr);
            PS_HIP_CHECK(hipGetLastError());
        }
        launch_gather(L, (int)pfnnz, lv.rmap.ptr, Pext.val.ptr, lv.R.val.ptr);
        // -- A_c = R (A P): the halo rows of A P come from their owners
        stack_halo_rows(comm, L, lv.link, AP.view, APext, I.sym, &lv.xAP);
        // the local rows of A P are the head of the stacked copy: keep one copy (the refresh writes the product there)
        AP.ptr.release();
        AP.col.release();
        AP.val.release();
        AP.view.rowptr = APext.ptr.ptr;
        AP.view.col = APext.col.ptr;
        AP.view.val = APext.val.ptr;
        std::unique_ptr<DLevel> nx(new DLevel());
        Launch Lc = fit_launch(ctx.launch_max(), nc_loc, 32);
        Lc.stream = s;
        const int64_t acnnz = device_spgemm_symbolic(Lc, nc_loc, lv.R.ptr.ptr, lv.R.col.ptr, APext.ptr.ptr, APext.col.ptr,
                                                     (int)nc_glob, nx->A_own.ptr, nx->A_own.col, I.sym);
        nx->A_own.val.ensure((size_t)acnnz + 4);
        nx->A_own.set_view(nc_loc, (int)nc_glob, acnnz);
        CsrMut Acm{nc_loc, nx->A_own.ptr.ptr, nx->A_own.col.ptr, nx->A_own.val.ptr};
        launch_spgemm_numeric(Lc, Acm, lv.R.view, APext.view, (double)acnnz / std::max(1, nc_loc));
        // -- the next level's column space: local coarse nodes + the halo A_c and P reach.  In front of the replicated
        //    tail both keep their global ids (P multiplies the gathered coarse solution, A_c is gathered as it is)
        nx->offsets = coff;
        nx->n = nc_loc;
        const bool next_replicated = (int)I.lv.size() + 2 < prm.max_levels && nc_glob > prm.coarse_enough &&
                                     nc_glob <= replicate_below;
        // (global-id copies of the two column arrays: the numeric refresh multiplies in that id space)
        lv.pcol_glob.ensure((size_t)pnnz + 4);
        lv.accol_glob.ensure((size_t)acnnz + 4);
        if (pnnz) PS_HIP_CHECK(hipMemcpyAsync(lv.pcol_glob.ptr, lv.P.col.ptr, (size_t)pnnz * sizeof(int), hipMemcpyDeviceToDevice, s));
        if (acnnz) PS_HIP_CHECK(hipMemcpyAsync(lv.accol_glob.ptr, nx->A_own.col.ptr, (size_t)acnnz * sizeof(int), hipMemcpyDeviceToDevice, s));
        if (next_replicated) {
            nx->n_ext = (int)nc_glob;
            nx->A = nx->A_own.view;
        } else {
            build_halo_link(comm, L, coff, {{nx->A_own.col.ptr, acnnz}, {lv.P.col.ptr, pnnz}}, nx->link, bs);
            nx->n_ext = nc_loc + nx->link.n_halo();
            nx->A_own.set_view(nc_loc, nx->n_ext, acnnz);
            nx->A = nx->A_own.view;
            lv.P.set_view(n, nx->n_ext, pnnz);
        }
        lv.has_next = true;
        I.lv.push_back(std::move(cur));
        cur = std::move(nx);
    }
    if (cur) I.lv.push_back(std::move(cur));
    for (auto &lv : I.lv) dist_smoother(ctx, I, *lv);
    PS_HIP_CHECK(hipStreamSynchronize(s));
    I.sym.tmp.release();
    I.sym.table.release();
    I.sym.cand.release();
    I.sym.tier.release();
    I.sym.cursor.release();
    I.agg.ints.release();
    I.agg.tptr.release();
    I.agg.tcol.release();
    I.agg.tmap.release();
}

// Same pattern, new values (Newton refactorizes a matrix of constant pattern every iteration, Newton.cpp:189-193): the
// aggregates, every pattern, the halo links and the row-exchange plans are kept; omega, P, the stacked P, A P, R, the
// stacked A P and R A P are recomputed level by level, the gathered level is gathered again and the replicated tail
// refreshes itself (amg.hpp).  Collective; returns false on the rank that finds its strength graph changed (an entry
// flipped between zero and nonzero: the aggregates would differ) -- the caller makes every rank rebuild then.  A rank
// that returns false has still gone through every collective of the sequence.
static bool dist_refresh(Context &ctx, DistAmg::Impl &I)
{
    const AmgParams &prm = I.prm;
    Comm &comm = ctx.comm();
    hipStream_t s = ctx.stream;
    const int bs = prm.block_size > 1 ? prm.block_size : 1;
    bool ok = true;
    DLevel &lv0 = *I.lv[0];
    lv0.A = ctx.A;
    lv0.A.bsr3 = nullptr;
    lv0.A.sell = nullptr;
    const int nd = (int)I.lv.size();
    for (int l = 0; l < nd; ++l) {
        DLevel &lv = *I.lv[(size_t)l];
        if (!lv.has_next) break;
        DLevel &nx = (l + 1 < nd) ? *I.lv[(size_t)l + 1] : *I.tail_src;
        Launch L = fit_setup_launch(ctx.launch_max(), lv.n, lv.A.nnz, lv.A.rows_per_block);
        L.stream = s;
        const int n = lv.n, ng = n / bs, nc_loc = nx.n;
        if (bs > 1) {
            BlockGraph &G = lv.Gf;
            device_block_values(L, lv.A, G);
            I.sym.tier.ensure((size_t)G.nnzb + 4);
            I.sym.cand.ensure((size_t)G.nb + 1);
            device_block_strong_flags(L, G, 0.0, I.sym.tier.ptr, I.sym.cand.ptr);
            if (device_block_flag_changes(L, G.nnzb, I.sym.tier.ptr, G.strong.ptr, I.sym) != 0) ok = false;
            const double gersh = allreduce_max(comm, s, device_block_gershgorin(L, G, I.partials.ptr));
            const double omega = prm.sa_relax * (prm.estimate_spectral_radius ? (4.0 / 3.0) / gersh : 2.0 / 3.0);
            launch_block_prolongation_values(L, G, lv.id_ext.ptr, omega, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr);
            launch_expand_block_csr(L, ng, bs, lv.pbptr.ptr, lv.pbcol.ptr, lv.pbval.ptr, nullptr, nullptr, lv.P.val.ptr);
        } else {
            unsigned long long h = 0;
            PS_HIP_CHECK(hipMemsetAsync(I.hash_dev.ptr, 0, sizeof(unsigned long long), s));
            launch_hash_nonzero(L, lv.A.nnz, lv.A.val, I.hash_dev.ptr);
            PS_HIP_CHECK(hipMemcpyAsync(&h, I.hash_dev.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            launch_gershgorin(L, lv.A, I.partials.ptr);
            std::vector<double> hg((size_t)L.grid);
            PS_HIP_CHECK(hipMemcpyAsync(hg.data(), I.partials.ptr, hg.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            PS_HIP_CHECK(hipStreamSynchronize(s));
            if (h != lv.nz_hash) ok = false;
            double gersh = 0.0;
            for (double v : hg) gersh = std::max(gersh, v);
            gersh = allreduce_max(comm, s, gersh);
            const double omega = prm.sa_relax * (prm.estimate_spectral_radius ? (4.0 / 3.0) / gersh : 2.0 / 3.0);
            CsrMut Pm{n, lv.P.ptr.ptr, lv.pcol_glob.ptr, lv.P.val.ptr}; // (global coarse ids, as the aggregate ids)
            launch_prolongation_values(L, lv.A, lv.id_ext.ptr, omega, nullptr, 0.0, Pm);
        }
        restack_values(comm, L, lv.link, lv.xP, lv.P.view, lv.Pext.val.ptr);
        CsrMut APm{n, lv.APext.ptr.ptr, lv.APext.col.ptr, lv.APext.val.ptr}; // (the head of the stacked copy)
        launch_spgemm_numeric(L, APm, lv.A, lv.Pext.view, (double)lv.AP.view.nnz / std::max(1, n));
        launch_gather(L, (int)lv.R.view.nnz, lv.rmap.ptr, lv.Pext.val.ptr, lv.R.val.ptr);
        restack_values(comm, L, lv.link, lv.xAP, lv.AP.view, lv.APext.val.ptr);
        Launch Lc = fit_launch(ctx.launch_max(), nc_loc, 32);
        Lc.stream = s;
        CsrMut Acm{nc_loc, nx.A_own.ptr.ptr, lv.accol_glob.ptr, nx.A_own.val.ptr};
        launch_spgemm_numeric(Lc, Acm, lv.R.view, lv.APext.view, (double)nx.A_own.view.nnz / std::max(1, nc_loc));
    }
    if (I.tail) {
        DLevel &src = *I.tail_src;
        Launch L = fit_launch(ctx.launch_max(), src.n, src.A_own.view.rows_per_block);
        L.stream = s;
        gather_rows(comm, L, src.offsets, src.A_own.view, I.tail_A); // same sizes: the arrays (and the tail's aliases) stay
        AmgParams tp = prm;
        tp.max_levels = std::max(1, prm.max_levels - nd);
        tp.renumber = 0;
        I.tail->setup(ctx, I.tail_A.view, tp);
    }
    for (auto &lv : I.lv) dist_smoother(ctx, I, *lv);
    PS_HIP_CHECK(hipStreamSynchronize(s));
    return ok;
}

static unsigned long long shard_pattern_hash(Context &ctx, DistAmg::Impl &I)
{
    const Launch L = ctx.launch_config();
    I.hash_dev.ensure(4);
    PS_HIP_CHECK(hipMemsetAsync(I.hash_dev.ptr, 0, 2 * sizeof(unsigned long long), L.stream));
    launch_hash_i32(L, (int64_t)ctx.A.n + 1, ctx.A.rowptr, I.hash_dev.ptr);
    launch_hash_i32(L, ctx.A.nnz, ctx.A.col, I.hash_dev.ptr + 1);
    unsigned long long h[2];
    PS_HIP_CHECK(hipMemcpyAsync(h, I.hash_dev.ptr, sizeof(h), hipMemcpyDeviceToHost, L.stream));
    PS_HIP_CHECK(hipStreamSynchronize(L.stream));
    return h[0] * 0x9E3779B97F4A7C15ull + h[1];
}

void DistAmg::setup(Context &ctx, const AmgParams &prm_in)
{
    Impl &I = *impl;
    Comm &comm = ctx.comm();
    I.reused = false;
    // (round 5's runtime classes are single-device / replicated-hierarchy features; every rank holds the same parameters, so
    // every rank throws alike)
    PS_REQUIRE(prm_in.relax_type == 0 && prm_in.precond_class == 0 && prm_in.coarsening == 0 && prm_in.direct_coarse == 0 && prm_in.cheb_scale != 0 &&
                   prm_in.aggregation == 0,
               PSOLVE_HIP_EINVAL,
               "the hierarchy built on shards (amg.dist_global 2) builds cg / smoothed_aggregation / chebyshev with amgcl's "
               "aggregation only; amg.relax_type, amg.coarsening, amg.aggregation, amg.direct_coarse, amg.cheb_scale = 0 need "
               "amg.dist_global 1 or 0, or one device");
    const unsigned long long h = shard_pattern_hash(ctx, I);
    const AmgParams &b = I.built_prm;
    bool same = I.symbolic_valid && prm_in.reuse && !I.lv.empty() && h == I.pattern_hash && ctx.A.n == I.pattern_n &&
                ctx.A.n_ext == I.pattern_next && ctx.A.nnz == I.pattern_nnz && prm_in.max_levels == b.max_levels &&
                prm_in.coarse_enough == b.coarse_enough && prm_in.sa_relax == b.sa_relax &&
                prm_in.estimate_spectral_radius == b.estimate_spectral_radius && prm_in.block_size == b.block_size &&
                prm_in.eps_strong == b.eps_strong && prm_in.dist_replicate_rows == b.dist_replicate_rows;
    // every rank takes the same path: one that cannot refresh makes all of them rebuild
    auto all_agree = [&](bool mine) {
        const std::vector<int64_t> v = allgather_counts(comm, ctx.stream, mine ? 0 : 1);
        for (int64_t x : v)
            if (x != 0) return false;
        return true;
    };
    I.prm = prm_in;
    if (all_agree(same)) {
        const bool ok = dist_refresh(ctx, I);
        if (all_agree(ok)) {
            I.reused = true;
            return;
        }
    }
    I.symbolic_valid = false;
    dist_full_setup(ctx, I);
    I.symbolic_valid = prm_in.reuse != 0;
    I.pattern_hash = h;
    I.pattern_n = ctx.A.n;
    I.pattern_next = ctx.A.n_ext;
    I.pattern_nnz = ctx.A.nnz;
    I.built_prm = prm_in;
}

// chebyshev::solve on the partitioned level: the iterate's halo travels before every product
static void dist_cheb(Context &ctx, DLevel &lv, int degree, const double *rhs, double *x_ext, bool x_is_zero, const int *done,
                      int bs)
{
    Comm &comm = ctx.comm();
    const Launch &L = lv.L;
    const double d = lv.d, c = lv.c;
    double alpha = 0.0, beta = 0.0;
    double *cur = x_ext, *other = lv.xb_ext.ptr;
    if (bs > 1) { // block scaling needs all residuals of a node: residual product, then a node-local update in place
        for (int k = 0; k < degree; ++k) {
            if (k == 0) {
                alpha = 1.0 / d;
                beta = 0.0;
            } else if (k == 1) {
                alpha = 2 * d * (1.0 / (2 * d * d - c * c));
                beta = alpha * d - 1.0;
            } else {
                alpha = 1.0 / (d - 0.25 * alpha * c * c);
                beta = alpha * d - 1.0;
            }
            const bool zero = (k == 0 && x_is_zero);
            const double *t = rhs;
            if (!zero) {
                exchange_halo(comm, L, lv.link, x_ext);
                launch_spmv(L, lv.A, SPMV_RESIDUAL, x_ext, rhs, lv.t_ext.ptr, nullptr, done);
                t = lv.t_ext.ptr;
            }
            launch_block_cheb_update(L, lv.n, bs, lv.dinv_blk.ptr, t, lv.p.ptr, x_ext, alpha, beta, zero);
        }
        return;
    }
    if (x_is_zero && ((degree - 1) & 1)) std::swap(cur, other);
    for (int k = 0; k < degree; ++k) {
        if (k == 0) {
            alpha = 1.0 / d;
            beta = 0.0;
        } else if (k == 1) {
            alpha = 2 * d * (1.0 / (2 * d * d - c * c));
            beta = alpha * d - 1.0;
        } else {
            alpha = 1.0 / (d - 0.25 * alpha * c * c);
            beta = alpha * d - 1.0;
        }
        if (k == 0 && x_is_zero) {
            launch_cheb_first(L, lv.n, alpha, lv.dinv.ptr, rhs, lv.p.ptr, cur);
            continue;
        }
        exchange_halo(comm, L, lv.link, cur);
        SpmvExtra ex;
        ex.dinv = lv.dinv.ptr;
        ex.p = lv.p.ptr;
        ex.alpha = alpha;
        ex.beta = beta;
        launch_spmv(L, lv.A, SPMV_CHEB, cur, rhs, other, nullptr, done, &ex);
        std::swap(cur, other);
    }
    if (cur != x_ext)
        PS_HIP_CHECK(hipMemcpyAsync(x_ext, cur, (size_t)lv.n * sizeof(double), hipMemcpyDeviceToDevice, L.stream));
}

static void dist_cycle(Context &ctx, DistAmg::Impl &I, size_t l, const double *rhs, double *x_ext, bool x_is_zero, const int *done)
{
    DLevel &lv = *I.lv[l];
    const AmgParams &prm = I.prm;
    Comm &comm = ctx.comm();
    const Launch &L = lv.L;
    hipStream_t s = L.stream;
    const bool last = l + 1 == I.lv.size();
    if (last && !I.tail) { // coarsest level, still partitioned: relaxed (direct_coarse = false, AMGCL.cpp:46)
        bool zero = x_is_zero;
        for (int i = 0; i < prm.npre + prm.npost; ++i) {
            dist_cheb(ctx, lv, prm.cheb_degree, rhs, x_ext, zero, done, prm.block_size > 1 ? prm.block_size : 1);
            zero = false;
        }
        if (zero) PS_HIP_CHECK(hipMemsetAsync(x_ext, 0, (size_t)lv.n * sizeof(double), s));
        return;
    }
    bool zero = x_is_zero;
    for (int j = 0; j < prm.ncycle; ++j) {
        for (int i = 0; i < prm.npre; ++i) {
            dist_cheb(ctx, lv, prm.cheb_degree, rhs, x_ext, zero, done, prm.block_size > 1 ? prm.block_size : 1);
            zero = false;
        }
        if (zero) {
            PS_HIP_CHECK(hipMemsetAsync(x_ext, 0, (size_t)lv.n * sizeof(double), s));
            zero = false;
        }
        exchange_halo(comm, L, lv.link, x_ext);
        launch_spmv(L, lv.A, SPMV_RESIDUAL, x_ext, rhs, lv.t_ext.ptr, nullptr, done);
        exchange_halo(comm, L, lv.link, lv.t_ext.ptr); // R's columns: this level's rows + halo
        if (!last) {
            DLevel &nx = *I.lv[l + 1];
            launch_spmv(nx.L, lv.R.view, SPMV_PLAIN, lv.t_ext.ptr, nullptr, nx.f.ptr, nullptr, done);
            dist_cycle(ctx, I, l + 1, nx.f.ptr, nx.x_ext.ptr, true, done);
            exchange_halo(comm, nx.L, nx.link, nx.x_ext.ptr); // P's columns: the next level's rows + halo
            launch_spmv(L, lv.P.view, SPMV_ADD, nx.x_ext.ptr, nullptr, x_ext, nullptr, done);
        } else {
            // the replicated tail: every rank restricts onto the coarse nodes it owns, the right-hand side is gathered,
            // the tail's cycle runs redundantly, P reads the gathered solution through global column ids
            const int W = comm.world(), me = comm.rank();
            const int64_t r0 = I.tail_offsets[(size_t)me];
            const int nc_loc = (int)(I.tail_offsets[(size_t)me + 1] - r0);
            Launch Lr = fit_launch(ctx.launch_max(), nc_loc, lv.R.view.rows_per_block);
            Lr.stream = s;
            launch_spmv(Lr, lv.R.view, SPMV_PLAIN, lv.t_ext.ptr, nullptr, I.tail_f.ptr + r0, nullptr, done);
            std::vector<int64_t> sc((size_t)W, 0), so((size_t)W, 0), rc((size_t)W, 0), ro((size_t)W, 0);
            for (int q = 0; q < W; ++q) {
                if (q == me) continue;
                sc[(size_t)q] = nc_loc;
                so[(size_t)q] = r0;
                rc[(size_t)q] = I.tail_offsets[(size_t)q + 1] - I.tail_offsets[(size_t)q];
                ro[(size_t)q] = I.tail_offsets[(size_t)q];
            }
            comm.exchange_f64(I.tail_f.ptr, sc, so, I.tail_f.ptr, rc, ro, s);
            I.tail->apply(ctx, I.tail_f.ptr, I.tail_u.ptr, done);
            launch_spmv(L, lv.P.view, SPMV_ADD, I.tail_u.ptr, nullptr, x_ext, nullptr, done);
        }
        for (int i = 0; i < prm.npost; ++i) dist_cheb(ctx, lv, prm.cheb_degree, rhs, x_ext, false, done, prm.block_size > 1 ? prm.block_size : 1);
    }
}

void DistAmg::apply(Context &ctx, const double *d_r, double *d_z, const int *done_flag)
{
    Impl &I = *impl;
    PS_REQUIRE(!I.lv.empty(), PSOLVE_HIP_EINVAL, "distributed AMG hierarchy is empty");
    DLevel &lv0 = *I.lv[0];
    dist_cycle(ctx, I, 0, d_r, lv0.x_ext.ptr, true, done_flag);
    PS_HIP_CHECK(hipMemcpyAsync(d_z, lv0.x_ext.ptr, (size_t)lv0.n * sizeof(double), hipMemcpyDeviceToDevice, lv0.L.stream));
}

} // namespace psolve
