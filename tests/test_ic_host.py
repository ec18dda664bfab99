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
