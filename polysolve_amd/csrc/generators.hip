// generators.hip -- synthetic block-3 elasticity system generated on the device (BASELINE.json configs[2],
// SURVEY.md 8(d) "Elasticity-Q1(M)"), so that the 3 M-DOF configuration never crosses PCIe:
// trilinear (Q1) hexahedra on an M^3-node unit cube, isotropic linear elasticity (E, nu), 2x2x2 Gauss
// quadrature, node-interleaved dofs (3 node + component), the face x = 0 clamped the way PolyFEM's
// dirichlet_solve does it (FEMSolver.cpp:136-161: entries in a Dirichlet row or column dropped, unit diagonal).
//
// Assembly is row-wise and deterministic: the thread of node (i, j, k) accumulates, for each of its 27
// neighbour slots, the 3x3 blocks of the up to 8 elements around the node, in a fixed element order; its
// three scalar rows then list the present neighbours with sorted columns.  Row lengths are counted first,
// prefix-summed by the device scan, then the entries are written.
#include <cmath>
#include <cstring>

#include "solver.hpp"

namespace psolve {

namespace {

// 24 x 24 stiffness matrix of one cubic Q1 element of edge h: K = sum_q w B_q^T D B_q
void element_matrix(double h, double E, double nu, double *Ke /* [24 * 24] row-major */)
{
    const double lam = E * nu / ((1 + nu) * (1 - 2 * nu)), mu = E / (2 * (1 + nu));
    double D[6][6] = {};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) D[a][b] = lam + (a == b ? 2 * mu : 0.0);
    for (int a = 3; a < 6; ++a) D[a][a] = mu;
    std::memset(Ke, 0, 24 * 24 * sizeof(double));
    const double g = 1.0 / std::sqrt(3.0);
    for (int q = 0; q < 8; ++q) { // Gauss points (+-g)^3, unit weights
        const double xi[3] = {(q & 1) ? g : -g, (q & 2) ? g : -g, (q & 4) ? g : -g};
        double dN[8][3]; // gradients of the 8 shape functions; local node l sits at bits (l&1, l>>1&1, l>>2&1)
        for (int l = 0; l < 8; ++l) {
            const double s[3] = {(l & 1) ? 1.0 : -1.0, (l & 2) ? 1.0 : -1.0, (l & 4) ? 1.0 : -1.0};
            for (int d = 0; d < 3; ++d) {
                double v = 0.125 * s[d];
                for (int e = 0; e < 3; ++e)
                    if (e != d) v *= (1 + s[e] * xi[e]);
                dN[l][d] = v * (2.0 / h);
            }
        }
        double B[6][24] = {};
        for (int l = 0; l < 8; ++l) {
            B[0][3 * l + 0] = dN[l][0];
            B[1][3 * l + 1] = dN[l][1];
            B[2][3 * l + 2] = dN[l][2];
            B[3][3 * l + 0] = dN[l][1]; B[3][3 * l + 1] = dN[l][0];
            B[4][3 * l + 1] = dN[l][2]; B[4][3 * l + 2] = dN[l][1];
            B[5][3 * l + 0] = dN[l][2]; B[5][3 * l + 2] = dN[l][0];
        }
        const double w = (h / 2) * (h / 2) * (h / 2);
        double DB[6][24];
        for (int a = 0; a < 6; ++a)
            for (int c = 0; c < 24; ++c) {
                double s = 0;
                for (int b = 0; b < 6; ++b) s += D[a][b] * B[b][c];
                DB[a][c] = s;
            }
        for (int r = 0; r < 24; ++r)
            for (int c = 0; c < 24; ++c) {
                double s = 0;
                for (int a = 0; a < 6; ++a) s += B[a][r] * DB[a][c];
                Ke[24 * r + c] += w * s;
            }
    }
}

// neighbour nodes present around node (i, j, k) that are not clamped (column survives), per scalar row
__device__ __forceinline__ int row_entries(int M, int i, int j, int k)
{
    if (i == 0) return 1; // clamped dof: unit diagonal
    const int ni = (i + 1 < M ? 1 : 0) + 1 + (i - 1 > 0 ? 1 : 0); // neighbour at x = 0 is dropped
    const int nj = (j > 0 ? 1 : 0) + 1 + (j + 1 < M ? 1 : 0);
    const int nk = (k > 0 ? 1 : 0) + 1 + (k + 1 < M ? 1 : 0);
    return 3 * ni * nj * nk;
}

__global__ __launch_bounds__(kBlock) void elasticity_count_kernel(int M, int *rowptr)
{
    const int64_t nodes = (int64_t)M * M * M;
    for (int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x; a < nodes; a += (int64_t)gridDim.x * kBlock) {
        const int i = (int)(a % M), j = (int)((a / M) % M), k = (int)(a / ((int64_t)M * M));
        const int c = row_entries(M, i, j, k);
        rowptr[3 * a] = c;
        rowptr[3 * a + 1] = c;
        rowptr[3 * a + 2] = c;
    }
}

// Assembly of the stiffness rows (the bench's stand-in for the caller's assembly; inside the timed generate + refresh of the
// bench line).  Round 5: 32 lanes per node, lane = neighbour slot (27 of them): every lane sums the element contributions of
// ITS 3 x 3 block (elements in ascending order, as before: the same numbers) and the lanes of a node write a row's entries side
// by side -- one thread per node kept 243 accumulators in scratch and wrote three rows 2 KB apart from its neighbour's
// (rocprofv3: 15.8 GB written for 2.8 GB of matrix, 4.7 ms at M = 100).
__global__ __launch_bounds__(kBlock) void elasticity_fill_kernel(int M, const double *__restrict__ Ke,
                                                                  const int *__restrict__ rowptr, int *__restrict__ col,
                                                                  double *__restrict__ val)
{
    __shared__ double ke[24 * 24];
    for (int t = threadIdx.x; t < 24 * 24; t += kBlock) ke[t] = Ke[t];
    __syncthreads();
    const int64_t nodes = (int64_t)M * M * M;
    const int slot = threadIdx.x & 31;
    const int64_t teams = (int64_t)gridDim.x * (kBlock / 32);
    for (int64_t a = (int64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5); a < nodes; a += teams) {
        const int i = (int)(a % M), j = (int)((a / M) % M), k = (int)(a / ((int64_t)M * M));
        if (i == 0) {
            if (slot < 3) {
                const int p = rowptr[3 * a + slot];
                col[p] = (int)(3 * a + slot);
                val[p] = 1.0;
            }
            continue;
        }
        if (slot >= 27) continue;
        const int di = slot % 3 - 1, dj = (slot / 3) % 3 - 1, dk = slot / 9 - 1;
        const int bi = i + di, bj = j + dj, bk = k + dk;
        const bool valid = bi > 0 && bi < M && bj >= 0 && bj < M && bk >= 0 && bk < M; // (bi == 0: Dirichlet column dropped)
        if (!valid) continue;
        // position of this block in the row: the valid slots before it
        int rank = 0;
        for (int s2 = 0; s2 < slot; ++s2) {
            const int ci = i + s2 % 3 - 1, cj = j + (s2 / 3) % 3 - 1, ck = k + s2 / 9 - 1;
            rank += (ci > 0 && ci < M && cj >= 0 && cj < M && ck >= 0 && ck < M) ? 1 : 0;
        }
        double acc[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] = 0.0;
        for (int e = 0; e < 8; ++e) { // elements around the node, origin (i - ei, j - ej, k - ek)
            const int ei = e & 1, ej = (e >> 1) & 1, ek = (e >> 2) & 1;
            const int ox = i - ei, oy = j - ej, oz = k - ek;
            if (ox < 0 || oy < 0 || oz < 0 || ox >= M - 1 || oy >= M - 1 || oz >= M - 1) continue;
            const int lx = bi - ox, ly = bj - oy, lz = bk - oz; // the neighbour's corner in this element, if it is one
            if (lx < 0 || lx > 1 || ly < 0 || ly > 1 || lz < 0 || lz > 1) continue;
            const int la = ei | (ej << 1) | (ek << 2), lb = lx | (ly << 1) | (lz << 2);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[3 * c + d] += ke[24 * (3 * la + c) + 3 * lb + d];
        }
        const int64_t b = bi + (int64_t)M * (bj + (int64_t)M * bk);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p = rowptr[3 * a + c] + 3 * rank;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                col[p + d] = (int)(3 * b + d);
                val[p + d] = acc[3 * c + d];
            }
        }
    }
}

} // namespace

namespace {
__global__ void node_renumber_kernel(int64_t nb, int b, int mode, int64_t window, uint64_t seed, int *dof_new);
}

void Context::generate_elasticity_q1(int M, double E, double nu) { generate_elasticity_q1_permuted(M, E, nu, 0, 0, 0); }

void Context::generate_elasticity_q1_permuted(int M, double E, double nu, int mode, int64_t window, uint64_t seed)
{
    use_device();
    PS_REQUIRE(mode == 0 || mode == 1 || (mode == 2 && window >= 2), PSOLVE_HIP_EINVAL,
               "generate_elasticity_q1_permuted: mode 0 (grid numbering), 1 (global) or 2 (windows of >= 2 nodes)");
    PS_REQUIRE(M >= 2 && M <= 890, PSOLVE_HIP_EINVAL, "generate_elasticity_q1: need 2 <= M <= 890 (3 M^3 < 2^31)");
    PS_REQUIRE(E > 0 && nu > -1.0 && nu < 0.5, PSOLVE_HIP_EINVAL, "generate_elasticity_q1: need E > 0, -1 < nu < 0.5");
    PS_REQUIRE(!comm_.active() || comm_.world() == 1, PSOLVE_HIP_EINVAL,
               "generate_elasticity_q1 builds the whole system on one device");
    const int64_t n = 3ll * M * M * M;
    factorized_ = false;
    rowptr_own_.ensure((size_t)n + 1);
    Launch L = Lmax_;
    L.stream = stream;
    hipLaunchKernelGGL(elasticity_count_kernel, dim3(L.grid), dim3(kBlock), 0, stream, M, rowptr_own_.ptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t nnz = device_exclusive_scan(L, rowptr_own_.ptr, n, bsr_scratch_);
    check_sizes_public(n, nnz);
    col_own_.ensure((size_t)nnz + 4);
    val_own_.ensure((size_t)nnz + 4);
    double Ke[24 * 24];
    element_matrix(1.0 / (M - 1), E, nu, Ke);
    DeviceBuffer<double> d_ke;
    d_ke.ensure(24 * 24);
    PS_HIP_CHECK(hipMemcpyAsync(d_ke.ptr, Ke, sizeof(Ke), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(elasticity_fill_kernel, dim3(L.grid), dim3(kBlock), 0, stream, M, d_ke.ptr, rowptr_own_.ptr,
                       col_own_.ptr, val_own_.ptr);
    PS_HIP_CHECK(hipGetLastError());
    PS_HIP_CHECK(hipStreamSynchronize(stream)); // Ke and d_ke die with this frame
    if (mode != 0) {
        // the NODES renumbered pseudo-randomly (xyz of a node stay together: the 3 x 3 blocks stay blocks): what an
        // unstructured mesh's numbering does to the same stiffness matrix
        DeviceBuffer<int> dof_new, pptr, pcol;
        DeviceBuffer<double> pval;
        dof_new.ensure((size_t)n + 1);
        hipLaunchKernelGGL(node_renumber_kernel, dim3(L.grid), dim3(kBlock), 0, stream, n / 3, 3, mode, window, seed,
                           dof_new.ptr);
        PS_HIP_CHECK(hipGetLastError());
        device_permute_csr(L, (int)n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, dof_new.ptr, dof_new.ptr, pptr,
                           pcol, &pval, bsr_scratch_);
        PS_HIP_CHECK(hipStreamSynchronize(stream));
        rowptr_own_.swap(pptr);
        col_own_.swap(pcol);
        val_own_.swap(pval);
    }
    n_global_ = n;
    row_begin_ = 0;
    row_end_ = n;
    gen_nx_ = gen_ny_ = gen_nz_ = 0;
    factorize_device(n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
}

// ---------------------------------------------------------------------------------------------------------
// 7-point Poisson under a symmetric pseudo-random renumbering, B = Pi A Pi^T ("report index
// compression separately"): the unstructured leg of the bench.  A caller's mesh numbering has no constant
// column offsets, so no pattern dictionary: the plain 12-byte-per-entry stream with real gathers.
//   mode 1: one pseudo-random permutation of all n rows (worst case: every gather its own cache line);
//   mode 2: rows shuffled inside consecutive windows of `window` rows (the locality a mesh generator's or an
//           RCM numbering leaves: neighbours a bounded distance away, no repeating offsets).
// The permutation is a keyed 4-round Feistel network on the smallest even-width bit field covering the domain,
// cycle-walked into [0, m): a bijection with a closed-form inverse, evaluated per row on the device and by
// permutation_host() for the tests.
// ---------------------------------------------------------------------------------------------------------
namespace {

__host__ __device__ inline uint32_t feistel_round(uint32_t v, uint64_t key, int r)
{
    uint64_t z = key + 0x9E3779B97F4A7C15ull * (uint64_t)(r + 1) + v;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}

// bijection of [0, m), m >= 1
__host__ __device__ inline int64_t feistel_perm(int64_t i, int64_t m, uint64_t key, bool inverse)
{
    if (m <= 1) return 0;
    int half = 1;
    while ((1ll << (2 * half)) < m) ++half;
    const uint32_t mask = (1u << half) - 1u;
    int64_t v = i;
    do {
        uint32_t l = (uint32_t)(v >> half) & mask, r = (uint32_t)v & mask;
        if (!inverse) {
            for (int k = 0; k < 4; ++k) {
                const uint32_t t = l ^ (feistel_round(r, key, k) & mask);
                l = r;
                r = t;
            }
        } else {
            for (int k = 3; k >= 0; --k) {
                const uint32_t t = r ^ (feistel_round(l, key, k) & mask);
                r = l;
                l = t;
            }
        }
        v = ((int64_t)l << half) | r;
    } while (v >= m);
    return v;
}

// new index of original row i (inverse = false) / original row of new index i (inverse = true)
__host__ __device__ inline int64_t renumber(int64_t i, int64_t n, int mode, int64_t window, uint64_t seed, bool inverse)
{
    if (mode == 1) return feistel_perm(i, n, seed, inverse);
    const int64_t w = i / window, base = w * window;
    const int64_t m = (base + window <= n) ? window : n - base;
    return base + feistel_perm(i - base, m, seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(w + 1)), inverse);
}

// dof_new[b i + c] = b renumber(i) + c
__global__ __launch_bounds__(kBlock) void node_renumber_kernel(int64_t nb, int b, int mode, int64_t window, uint64_t seed,
                                                                int *dof_new)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kBlock) {
        const int64_t j = renumber(i, nb, mode, window, seed, false);
        for (int c = 0; c < b; ++c) dof_new[i * b + c] = (int)(j * b + c);
    }
}

__global__ __launch_bounds__(kBlock) void poisson7_perm_count_kernel(int nx, int ny, int nz, int mode, int64_t window,
                                                                      uint64_t seed, int *rowptr)
{
    const int64_t plane = (int64_t)nx * ny, n = plane * nz;
    for (int64_t rp = (int64_t)blockIdx.x * kBlock + threadIdx.x; rp < n; rp += (int64_t)gridDim.x * kBlock) {
        const int64_t r = renumber(rp, n, mode, window, seed, true);
        const int64_t k = r / plane, rem = r - k * plane;
        const int j = (int)(rem / nx), i = (int)(rem - (int64_t)j * nx);
        rowptr[rp] = 1 + (k > 0) + (j > 0) + (i > 0) + (i < nx - 1) + (j < ny - 1) + (k < nz - 1);
    }
}

__global__ __launch_bounds__(kBlock) void poisson7_perm_fill_kernel(int nx, int ny, int nz, int mode, int64_t window,
                                                                     uint64_t seed, const int *__restrict__ rowptr,
                                                                     int *__restrict__ col, double *__restrict__ val)
{
    const int64_t plane = (int64_t)nx * ny, n = plane * nz;
    for (int64_t rp = (int64_t)blockIdx.x * kBlock + threadIdx.x; rp < n; rp += (int64_t)gridDim.x * kBlock) {
        const int64_t r = renumber(rp, n, mode, window, seed, true);
        const int64_t k = r / plane, rem = r - k * plane;
        const int j = (int)(rem / nx), i = (int)(rem - (int64_t)j * nx);
        int c[7];
        double v[7];
        int m = 0;
        auto put = [&](int64_t orig, double a) {
            const int cc = (int)renumber(orig, n, mode, window, seed, false);
            int q = m++;
            while (q > 0 && c[q - 1] > cc) { // insertion: columns ascending
                c[q] = c[q - 1];
                v[q] = v[q - 1];
                --q;
            }
            c[q] = cc;
            v[q] = a;
        };
        if (k > 0) put(r - plane, -1.0);
        if (j > 0) put(r - nx, -1.0);
        if (i > 0) put(r - 1, -1.0);
        put(r, 6.0);
        if (i < nx - 1) put(r + 1, -1.0);
        if (j < ny - 1) put(r + nx, -1.0);
        if (k < nz - 1) put(r + plane, -1.0);
        const int p = rowptr[rp];
        for (int q = 0; q < m; ++q) {
            col[p + q] = c[q];
            val[p + q] = v[q];
        }
    }
}

} // namespace

void permutation_host(int64_t n, int mode, int64_t window, uint64_t seed, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = (int32_t)renumber(i, n, mode, window, seed, false);
}

void Context::generate_poisson7_permuted(int nx, int ny, int nz, int mode, int64_t window, uint64_t seed)
{
    use_device();
    PS_REQUIRE(nx > 0 && ny > 0 && nz > 0, PSOLVE_HIP_EINVAL, "generate_poisson7_permuted: bad grid");
    PS_REQUIRE(mode == 1 || (mode == 2 && window >= 2), PSOLVE_HIP_EINVAL,
               "generate_poisson7_permuted: mode 1 (global) or 2 (windows of >= 2 rows)");
    PS_REQUIRE(!comm_.active() || comm_.world() == 1, PSOLVE_HIP_EINVAL,
               "generate_poisson7_permuted builds the whole system on one device");
    const int64_t n = (int64_t)nx * ny * nz;
    PS_REQUIRE(n < (int64_t)INT32_MAX, PSOLVE_HIP_ERANGE, "global size exceeds int32 column ids");
    factorized_ = false;
    rowptr_own_.ensure((size_t)n + 1);
    Launch L = Lmax_;
    L.stream = stream;
    hipLaunchKernelGGL(poisson7_perm_count_kernel, dim3(L.grid), dim3(kBlock), 0, stream, nx, ny, nz, mode, window, seed,
                       rowptr_own_.ptr);
    PS_HIP_CHECK(hipGetLastError());
    const int64_t nnz = device_exclusive_scan(L, rowptr_own_.ptr, n, bsr_scratch_);
    check_sizes_public(n, nnz);
    col_own_.ensure((size_t)nnz + 4);
    val_own_.ensure((size_t)nnz + 4);
    hipLaunchKernelGGL(poisson7_perm_fill_kernel, dim3(L.grid), dim3(kBlock), 0, stream, nx, ny, nz, mode, window, seed,
                       rowptr_own_.ptr, col_own_.ptr, val_own_.ptr);
    PS_HIP_CHECK(hipGetLastError());
    n_global_ = n;
    row_begin_ = 0;
    row_end_ = n;
    gen_nx_ = gen_ny_ = gen_nz_ = 0;
    factorize_device(n, nnz, rowptr_own_.ptr, col_own_.ptr, val_own_.ptr, true);
}

} // namespace psolve
