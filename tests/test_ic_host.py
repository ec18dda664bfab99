"""Host half of precond = "ic" (psolve_hip_ic_host_factorize, no GPU) against the CPU oracle's restatement of
Eigen::IncompleteCholesky<double, Lower, NaturalOrdering<int>> (oracle/ic_oracle.c), and the oracle itself against what
the algorithm guarantees: exact Cholesky where nothing is dropped, an SPD preconditioner that cuts the PCG count."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl


def _cases(oracle):
    rng = np.random.default_rng(7)
    n = 400
    R = sp.random(n, n, density=0.02, random_state=3, format="csr")
    R = (R + R.T).tocsr()
    weak = (R + sp.diags(np.asarray(abs(R).sum(axis=1)).ravel() * 0.45 + 1e-3)).tocsr()  # not diagonally dominant
    weak.sort_indices()
    return {"poisson": oracle.poisson7(9, 7, 8), "gr3030": oracle.gr_30_30(), "elasticity": oracle.elasticity_q1(5),
            "weak_diagonal": oracle.CSR.from_scipy(weak)}


@pytest.mark.parametrize("name", ["poisson", "gr3030", "elasticity", "weak_diagonal"])
def test_host_factor_is_the_oracles(oracle, name):
    from polysolve_amd import ic_host_factorize
    A = _cases(oracle)[name]
    ref = oracle.IC(A)
    assert ref.ok
    colptr, rowidx, vals, scale, shift, attempts = ic_host_factorize(A.n, A.rowptr, A.col, A.val)
    rc, rr, rv, rs = ref.factor()
    assert shift == ref.shift and attempts == ref.attempts
    assert np.array_equal(colptr, rc) and np.array_equal(rowidx, rr)
    assert np.array_equal(vals, rv) and np.array_equal(scale, rs)  # the same operations in the same order
    if name == "weak_diagonal":
        assert ref.attempts > 1 and ref.shift > 0  # a non-positive pivot made the factorization restart with a shift
    # every column keeps the diagonal first and exactly as many entries as the matrix column has below the diagonal
    low = sp.tril(A.to_scipy()).tocsc()
    assert np.array_equal(np.diff(colptr), np.diff(low.indptr))
    assert np.array_equal(rowidx[colptr[:-1]], np.arange(A.n))


def test_oracle_ic_is_exact_without_dropping_and_preconditions_pcg(oracle):
    n = 60
    T = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    A = oracle.CSR.from_scipy(T)
    ic = oracle.IC(A)
    r = np.random.default_rng(0).uniform(-1, 1, n)
    assert np.abs(ic.apply(r) - spl.spsolve(T.tocsc(), r)).max() < 1e-13  # a tridiagonal matrix has no fill-in to drop
    P = oracle.poisson7(16)
    ic = oracle.IC(P)
    colptr, rowidx, vals, scale = ic.factor()
    L = sp.csc_matrix((vals, rowidx, colptr), shape=(P.n, P.n))
    Minv = sp.diags(scale) @ spl.inv((L @ L.T).tocsc()) @ sp.diags(scale)
    z = ic.apply(r0 := oracle.splitmix_vector(P.n, 3))
    assert np.abs(z - Minv @ r0).max() < 1e-10 * np.abs(z).max()       # apply = S L^-T L^-1 S
    w = np.linalg.eigvalsh(((Minv + Minv.T) * 0.5).toarray())
    assert w.min() > 0                                                 # SPD: valid for PCG
    b = oracle.spmv(P, oracle.splitmix_vector(P.n, 42))
    _, it_j, _ = oracle.cg_eigen(P, b, tol=1e-8)
    x, it_ic, _ = oracle.cg_eigen(P, b, precond=ic, tol=1e-8)
    assert it_ic < 0.5 * it_j and np.linalg.norm(oracle.spmv(P, x) - b) < 1.5e-8 * np.linalg.norm(b)


def test_missing_diagonal_is_an_error(oracle):
    from polysolve_amd import ic_host_factorize
    M = sp.csr_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 2.0]]))
    M.eliminate_zeros()
    with pytest.raises(RuntimeError, match="no stored diagonal"):
        ic_host_factorize(3, M.indptr, M.indices, M.data)
    with pytest.raises(ValueError):
        oracle.IC(oracle.CSR.from_scipy(M))


def test_amd_ordering_product_equals_oracle_and_reduces_fill(oracle):
    """Eigen::AMDOrdering<int> restated twice from the published algorithm (CSparse cs_amd as Eigen's Amd.h adapts it):
    oracle/amd_oracle.c and polysolve_amd/csrc/amd_order.cpp, independent transcriptions -- the same pivot sequence entry by
    entry; a permutation; and a fill-reducing one: the exact Cholesky factor of the reordered matrix has far fewer entries
    than in the natural order (what the algorithm is for; SuperLU's multiple minimum degree is within a few per cent)."""
    import ctypes as C
    import scipy.sparse.linalg as sla
    from polysolve_amd import _lib
    L = _lib.load()

    def fill(A):
        return sla.splu(A.to_scipy().tocsc(), permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)).L.nnz

    for A, gain in ((oracle.poisson7(14, 12, 10), 0.55), (oracle.gr_30_30(), 0.7), (oracle.elasticity_q1(5), 0.95), (oracle.poisson7(300, 1, 1), 1.01)):
        o = oracle.amd_order(A)
        assert np.array_equal(np.sort(o), np.arange(A.n))
        p = np.empty(A.n, np.int32)
        assert L.psolve_hip_amd_order(A.n, A.rowptr.ctypes.data, A.col.ctypes.data, p.ctypes.data) == 0
        assert np.array_equal(o, p)
        f_nat, f_amd = fill(A), fill(oracle.permuted(A, o))
        mmd = sla.splu(A.to_scipy().tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)).L.nnz
        assert f_amd <= gain * f_nat and f_amd <= 1.1 * mmd
    # a diagonal matrix: every node is eliminated at once, in index order
    D = oracle.CSR(5, np.arange(6, dtype=np.int32), np.arange(5, dtype=np.int32), np.ones(5))
    assert np.array_equal(oracle.amd_order(D), np.arange(5))


def test_amd_ordering_symmetrises_the_pattern_it_is_given(oracle):
    """The quotient graph is that of A + A^T (Eigen orders mat.selfadjointView): a pattern handed over as one triangle, or
    with a few entries of one triangle missing, or with repeated entries, is symmetrised inside amd_order.cpp and gives the
    pivot sequence of the full pattern; an index outside [0, n) is refused rather than written through."""
    from polysolve_amd import _lib
    L = _lib.load()

    def order(M):
        M = M.tocsr()
        rp, ci = M.indptr.astype(np.int32), M.indices.astype(np.int32)
        p = np.empty(M.shape[0], np.int32)
        assert L.psolve_hip_amd_order(M.shape[0], rp.ctypes.data, ci.ctypes.data, p.ctypes.data) == 0
        return p

    rng = np.random.default_rng(5)
    for A in (oracle.poisson7(9, 8, 7), oracle.gr_30_30(), oracle.elasticity_q1(4)):
        M = A.to_scipy().tocsr()
        full = order(M)
        assert np.array_equal(full, oracle.amd_order(A))
        assert np.array_equal(order(sp.triu(M)), full)
        assert np.array_equal(order(sp.tril(M)), full)
        C = M.tocoo()
        keep = (C.row <= C.col) | (rng.random(C.nnz) < 0.5)
        assert np.array_equal(order(sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=M.shape)), full)
        # repeated entries (an unmerged assembly): rows with duplicates
        rp = np.concatenate([[0], np.cumsum(2 * np.diff(M.indptr))]).astype(np.int32)
        ci = np.concatenate([np.tile(M.indices[M.indptr[i]:M.indptr[i + 1]], 2) for i in range(M.shape[0])]).astype(np.int32)
        p = np.empty(M.shape[0], np.int32)
        assert L.psolve_hip_amd_order(M.shape[0], rp.ctypes.data, ci.ctypes.data, p.ctypes.data) == 0
        assert np.array_equal(p, full)
    M = oracle.poisson7(4, 4, 4).to_scipy().tocsr()
    rp, ci = M.indptr.astype(np.int32), M.indices.astype(np.int32).copy()
    ci[5] = M.shape[0]
    p = np.empty(M.shape[0], np.int32)
    assert L.psolve_hip_amd_order(M.shape[0], rp.ctypes.data, ci.ctypes.data, p.ctypes.data) != 0


def test_ic_in_the_amd_ordering_is_spd_and_helps(oracle):
    """IncompleteCholesky<double>'s default instantiation: M^-1 = P^T S L^-T L^-1 S P is symmetric positive definite and
    PCG with it needs about as many iterations as with the natural-ordering factor."""
    A = oracle.poisson7(9, 8, 7)
    ic = oracle.IC(A, ordering="amd")
    assert ic.order is not None and ic.ok
    n = A.n
    Minv = np.column_stack([ic.apply(np.eye(n)[:, k]) for k in range(n)])
    assert np.abs(Minv - Minv.T).max() <= 1e-13 * np.abs(Minv).max() and np.linalg.eigvalsh(0.5 * (Minv + Minv.T)).min() > 0
    b = oracle.spmv(A, oracle.splitmix_vector(n, 42))
    Ap = oracle.permuted(A, ic.order)
    _, it_amd, _ = oracle.cg_eigen(Ap, b[ic.order], precond=oracle.IC(Ap), tol=1e-9)
    _, it_nat, _ = oracle.cg_eigen(A, b, precond=oracle.IC(A), tol=1e-9)
    _, it_jac, _ = oracle.cg_eigen(A, b, tol=1e-9)
    assert it_amd < it_jac and it_amd <= it_nat + 6
