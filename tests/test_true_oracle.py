"""Opportunistic pin of the CPU oracle against the REAL reference libraries: runs only where
Eigen / AMGCL headers exist (not in this image -- both legs then skip, and the oracle stays "parity unpinned")."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    L = oracle.true_oracle()
    if L is None:
        pytest.skip("true-oracle harness could not be built")
    return L


def test_harness_builds_and_reports_availability(ref):
    assert ref.ref_have_eigen() in (0, 1) and ref.ref_have_amgcl() in (0, 1)


@pytest.mark.parametrize("precond", [0, 1])
def test_cg_eigen_restatement_vs_real_eigen(oracle, ref, precond):
    if not ref.ref_have_eigen():
        pytest.skip("Eigen headers not present on this box")
    for A in (oracle.poisson7(12), oracle.poisson7(9, 7, 11), oracle.elasticity_q1(5), oracle.gr_30_30()):
        b = oracle.splitmix_vector(A.n, 42)
        x = np.zeros(A.n)
        it, err = C.c_int64(), C.c_double()
        assert ref.ref_eigen_cg(A.n, A.rowptr, A.col, A.val, b, x, precond, 1e-10, 10000, C.byref(it), C.byref(err)) == 0
        xo, ito, erro = oracle.cg_eigen(A, b, precond="jacobi" if precond else "none", tol=1e-10)
        assert it.value == ito
        assert np.abs(x - xo).max() <= 1e-12 * np.abs(xo).max()
        assert abs(err.value - erro) <= 1e-6 * erro


@pytest.mark.parametrize("bs", [1, 3])
def test_amgcl_restatement_vs_real_amgcl(oracle, ref, bs):
    if not ref.ref_have_amgcl():
        pytest.skip("AMGCL headers not present on this box")
    A = oracle.elasticity_q1(9) if bs == 3 else oracle.poisson7(24)
    b = oracle.splitmix_vector(A.n, 42)
    x = np.zeros(A.n)
    it, err = C.c_int64(), C.c_double()
    assert ref.ref_amgcl_solve(A.n, A.rowptr, A.col, A.val, b, x, bs, 1e-10, 1000, C.byref(it), C.byref(err), None, None) == 0
    amg = oracle.AMG(A, block_size=bs)
    xo, ito, erro = oracle.cg_amgcl(A, b, precond=amg, tol=1e-10, max_iter=1000)
    assert it.value == ito
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
