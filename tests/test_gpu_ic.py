"""GPU parity of precond = "ic" (Eigen::IncompleteCholesky's factorization in its default approximate-minimum-degree
ordering -- "ic.ordering" 1, oracle/amd_oracle.c -- and in the natural one; ic.hpp) against oracle/ic_oracle.c: the action z = S L^-T L^-1 S r (two triangular solves in which rows wait for the rows they depend on:
the additions of a row come in the factor's column order, as in the oracle's row-oriented backward solve; the forward
solve is column-oriented there and row-oriented here -- 1e-12 relative), PCG counts within one of Eigen's recurrence
with the oracle's preconditioner, the names at the boundary."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


def _mat(oracle, name):
    if name == "weak_diagonal":  # SPD by construction, far from an M-matrix: the factorization restarts with shifts
        B = sp.random(2000, 2000, density=0.003, random_state=5, format="csr")
        M = (B.T @ B + 0.1 * sp.identity(2000)).tocsr()
        M.sort_indices()
        return oracle.CSR.from_scipy(M)
    if name == "tets":  # an unstructured mesh: an irregular dependency graph for the waiting triangular solves
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import mesh_utils as mu
        P, T, bd = mu.tet_mesh(12, seed=4)
        return oracle.CSR.from_scipy(mu.renumber_nodes(mu.p1_laplace(P, T, bd), 1, seed=5)[0])
    return {"poisson": lambda: oracle.poisson7(20, 17, 23), "gr3030": oracle.gr_30_30, "elasticity": lambda: oracle.elasticity_q1(8),
            "line": lambda: oracle.poisson7(700, 1, 1)}[name]()


@pytest.mark.parametrize("ordering", ["natural", "amd"])
@pytest.mark.parametrize("name", ["poisson", "gr3030", "elasticity", "weak_diagonal", "line", "tets"])
def test_ic_apply_and_pcg_match_oracle(S, oracle, name, ordering):
    A = _mat(oracle, name)
    ref = oracle.IC(A, ordering=ordering)
    s = S.create("HIP", "")
    assert s.get_param("ic.ordering") == 1  # the default: Eigen's default
    s.set_parameters({"HIP": {"precond": "ic", "tolerance": 1e-9, "max_iter": 2000, "ic": {"ordering": int(ordering == "amd")}}})
    M = A.to_scipy()
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    assert s.get_param("ic.shift") == ref.shift and s.get_param("ic.attempts") == ref.attempts
    assert ref.attempts > 1 or name != "weak_diagonal"
    if ordering == "natural":
        assert s.get_param("ic.levels") >= (A.n if name == "line" else 2)  # a chain is one row per level
    else:  # (minimum degree walks a chain from one end: as deep as the natural order; a grid gets a handful of levels)
        assert 2 <= s.get_param("ic.levels") <= A.n and (name != "poisson" or s.get_param("ic.levels") < 20)
    for seed in (1, 2):  # twice: the flags of the waiting kernels are epochs, not cleared between applies
        r = oracle.splitmix_vector(A.n, seed)
        z = s.device_array(A.n)
        s.precond_apply_device(s.to_device(r), z)
        zo = ref.apply(r)
        assert np.abs(z.download() - zo).max() <= 1e-12 * np.abs(zo).max()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    if ordering == "amd":  # the oracle's PCG on the explicitly permuted system (its preconditioner: the natural factor of it)
        Ap = oracle.permuted(A, ref.order)
        xp, ito, _ = oracle.cg_eigen(Ap, b[ref.order], precond=oracle.IC(Ap), tol=1e-9, max_iter=2000)
        xo = np.empty(A.n)
        xo[ref.order] = xp
    else:
        xo, ito, _ = oracle.cg_eigen(A, b, precond=ref, tol=1e-9, max_iter=2000)
    info = s.get_info()
    assert abs(info["solver_iter"] - ito) <= 1 and np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    assert info["true_residual"] < 1.5e-9
    _, itj, _ = oracle.cg_eigen(A, b, tol=1e-9, max_iter=5000)
    assert info["solver_iter"] < itj  # it is worth something against Jacobi


def test_ic_names_refactorize_and_shards(S, oracle):
    from polysolve_amd import HIPSolver
    A = oracle.poisson7(12, 11, 16)
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = S.create("HIP", "Eigen::IncompleteCholesky")  # the factory's name: honoured, with a word about the ordering
    assert any("AMD" in str(x.message) and "not validated" in str(x.message) for x in w)
    assert s.get_param("precond") == 4
    s.set_parameters({"HIP": {"tolerance": 1e-9}})
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    it1 = s.get_info()["solver_iter"]
    order = oracle.amd_order(A)
    Ap = oracle.permuted(A, order)
    xp, ito, _ = oracle.cg_eigen(Ap, b[order], precond=oracle.IC(Ap), tol=1e-9)
    xo = np.empty(A.n)
    xo[order] = xp
    assert abs(it1 - ito) <= 1
    # new values, same pattern (Newton): factorized again
    s.factorize((M * 3.0).tocsc())
    x3 = np.zeros(A.n)
    s.solve(b, x3)
    assert np.abs(3 * x3 - x).max() <= 1e-6 * np.abs(x).max()
    # selected after factorize: refused until the next factorize
    t = S.create("HIP", "")
    t.factorize(M)
    t.set_parameters({"HIP": {"precond": "ic"}})
    with pytest.raises(RuntimeError, match="precond=ic was selected after factorize"):
        t.solve(b, np.zeros(A.n))
    # shards: incomplete factors of the diagonal blocks (block Jacobi of them)
    m = HIPSolver("", devices=[0, 0, 0])
    m.set_parameters({"HIP": {"precond": "ic", "tolerance": 1e-9}})
    m.factorize(M)
    xm = np.zeros(A.n)
    m.solve(b, xm)
    assert np.abs(xm - xo).max() <= 1e-6 * np.abs(xo).max()
    assert it1 <= m.get_info()["solver_iter"] < 3 * it1
    # JSON spec: the /HIP/ic object
    j = S.create({"solver": "HIP", "HIP": {"precond": "ic", "ic": {"initial_shift": 0.01, "ordering": 0}}})
    assert j.get_param("precond") == 4 and j.get_param("ic.initial_shift") == 0.01 and j.get_param("ic.ordering") == 0
