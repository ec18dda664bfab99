"""GPU parity of "reorder" (Cuthill-McKee renumbering at factorize, polysolve_amd/csrc/reorder.hip) against
oracle/reorder_oracle.c.  The order is integer work: bit-exact.  The renumbered solve is the oracle's solve of the
explicitly permuted system P A P^T (P b): row sums bit-equal (sorted columns in both), iteration counts within one,
solutions to 1e-6 relative (the stated floating-point tolerance of the PCG parity tests, tests/test_gpu_solver.py).
Reference precedent for renumbering inside a backend: MASSolver (mas_utils/GraphPartition.cpp:240-243)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


def _shuffled(oracle, A, seed):
    """the same operator under a random symmetric renumbering (what a mesh generator's numbering may look like)"""
    rng = np.random.default_rng(seed)
    return oracle.permuted(A, rng.permutation(A.n).astype(np.int32))


def _cases(oracle, name):
    if name == "poisson_shuffled":
        return _shuffled(oracle, oracle.poisson7(19, 14, 11), 1)
    if name == "poisson_natural":
        return oracle.poisson7(12, 9, 7)
    if name == "gr3030_shuffled":
        return _shuffled(oracle, oracle.gr_30_30(), 2)
    if name == "dirichlet_rows":  # identity rows (FEMSolver.cpp:136-161) are isolated vertices: placed first
        M = oracle.poisson7(10, 10, 6).to_scipy().tolil()
        for i in range(0, M.shape[0], 7):
            M[i, :] = 0.0
            M[:, i] = 0.0
            M[i, i] = 1.0
        M = M.tocsr()
        M.eliminate_zeros()
        return _shuffled(oracle, oracle.CSR.from_scipy(M), 3)
    if name == "components":  # three grids of different sizes + isolated rows, interleaved by the shuffle
        blocks = [oracle.poisson7(6, 5, 4).to_scipy(), oracle.poisson7(9, 1, 1).to_scipy(), oracle.poisson7(5, 5, 1).to_scipy(),
                  sp.identity(4, format="csr") * 2.0]
        return _shuffled(oracle, oracle.CSR.from_scipy(sp.block_diag(blocks, format="csr")), 4)
    if name == "many_components":  # more components than the search walks: the rest follows in index order
        blocks = [oracle.poisson7(2 + (k % 3), 2, 1).to_scipy() for k in range(80)]
        return _shuffled(oracle, oracle.CSR.from_scipy(sp.block_diag(blocks, format="csr")), 5)
    if name == "wide_levels":  # random sparse SPD graph: few, very wide levels (many tiles per level)
        n = 30000
        rng = np.random.default_rng(11)
        B = sp.csr_matrix((rng.random(3 * n), (rng.integers(0, n, 3 * n), rng.integers(0, n, 3 * n))), shape=(n, n))
        G = (B + B.T).tocsr()
        G.setdiag(0)
        G.eliminate_zeros()
        L = (sp.diags(np.asarray(G.sum(axis=1)).ravel() + 1.0) - G).tocsr()
        L.sort_indices()
        return oracle.CSR.from_scipy(L)
    raise KeyError(name)


NAMES = ["poisson_shuffled", "poisson_natural", "gr3030_shuffled", "dirichlet_rows", "components", "many_components", "wide_levels"]


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_order_is_the_oracles_and_products_match(S, oracle, name, rev):
    """rev: "reorder_reverse" (the default): the breadth-first order read backwards."""
    A = _cases(oracle, name)
    order, oinfo = oracle.cuthill_mckee(A, reverse=rev)
    assert np.array_equal(np.sort(order), np.arange(A.n))
    s = S.create("HIP", "")
    assert s.get_param("reorder_reverse") == 1  # the default
    s.set_parameters({"HIP": {"reorder": 1, "reorder_reverse": rev, "tolerance": 1e-9, "max_iter": 5000}})
    M = A.to_scipy()
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    perm, active = s.reorder_perm()
    assert active and s.get_param("reorder.active") == 1
    new_of_old = np.empty(A.n, np.int32)
    new_of_old[order] = np.arange(A.n, dtype=np.int32)
    assert np.array_equal(perm, new_of_old)  # bit-exact: the sequential definition, level by level on the device
    for k in ("levels", "components", "isolated", "leftover"):
        assert s.get_param("reorder." + k) == oinfo[k], k
    if name == "many_components":
        assert oinfo["components"] == 64 and oinfo["leftover"] > 0
    # the product in the caller's numbering: y = P^T (P A P^T) P x, row sums of the permuted matrix bit for bit
    B = oracle.permuted(A, order)
    x = oracle.splitmix_vector(A.n, 5)
    y = s.device_array(A.n)
    s.spmv_device(s.to_device(x), y)
    yo = np.empty(A.n)
    yo[order] = oracle.spmv(B, x[order])
    if s.get_param("spmv_rows_per_block") == 256:  # one thread per row: the additions in column order, as the oracle's loop
        assert np.array_equal(y.download(), yo)
    else:  # several threads per row (more than ~7 entries per row): partial sums folded by a butterfly
        assert np.abs(y.download() - yo).max() <= 1e-14 * np.abs(yo).max() * np.diff(A.rowptr).max()
    # the solve: the oracle's PCG on the permuted system
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    xg = np.zeros(A.n)
    s.solve(b, xg)
    xo_new, ito, _ = oracle.cg_eigen(B, b[order], tol=1e-9, max_iter=5000)
    xo = np.empty(A.n)
    xo[order] = xo_new
    info = s.get_info()
    assert abs(info["solver_iter"] - ito) <= 1
    assert np.abs(xg - xo).max() <= 1e-6 * np.abs(xo).max()
    assert info["true_residual"] < 1.5e-9
    # Jacobi through the wrapper
    r = oracle.splitmix_vector(A.n, 9)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    assert np.array_equal(z.download(), oracle.jacobi_setup(A) * r)
    # the forward order is idempotent: the search on the renumbered matrix finds the identity (a size-independent
    # property: the second search meets every vertex's children in ascending index = the order the first one appended
    # them)
    if not rev:
        s.analyze_pattern(B.to_scipy(), B.n)
        s.factorize(B.to_scipy())
        perm2, active2 = s.reorder_perm()
        assert active2 and np.array_equal(perm2, np.arange(A.n))


def test_initial_guess_refactorize_and_switching_off(S, oracle):
    A = _shuffled(oracle, oracle.poisson7(16, 12, 9), 7)
    M = A.to_scipy()
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 1, "tolerance": 1e-9}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    xs = oracle.splitmix_vector(A.n, 42)
    b = oracle.spmv(A, xs)
    x = np.zeros(A.n)
    s.solve(b, x)
    it0 = s.get_info()["solver_iter"]
    s.solve(b, x)  # the solution as the initial guess (Solver.hpp:119-127; tests/test_linear_solver.cpp:400-455)
    assert s.get_info()["solver_iter"] == 0
    # same pattern, new values (Newton.cpp:189-193): the order is kept, only the values are permuted again
    t_first = s.get_param("reorder.seconds")
    M2 = M.copy()
    M2.data = M2.data * 2.0
    s.factorize(M2)
    p1, _ = s.reorder_perm()
    x2 = np.zeros(A.n)
    s.solve(b, x2)
    assert np.abs(2.0 * x2 - x).max() <= 1e-7 * np.abs(x).max()
    assert s.get_param("reorder.levels") > 0 and t_first > 0
    # round 6: that second factorize did not sort the rows again -- it gathered the values through the map the first one left
    # (kept with the order).  Generic new values (every entry its own): the operator on the device is P M3 P^T bit for bit,
    # for the third factorize of the pattern as for a fresh handle's first
    d = 1.0 + 0.5 * np.random.default_rng(5).uniform(0, 1, A.n)
    M3 = sp.csr_matrix(sp.diags(d) @ M @ sp.diags(d))
    M3.sort_indices()
    assert np.array_equal(M3.indptr, sp.csr_matrix(M).indptr)
    s.factorize(M3)
    p3_, act = s.reorder_perm()
    assert act and np.array_equal(p3_, p1)
    ptr, col, val = s.matrix_to_host()
    new_of_old = p3_
    P = sp.csr_matrix((np.ones(A.n), (new_of_old, np.arange(A.n))), shape=(A.n, A.n))
    want = sp.csr_matrix(P @ M3 @ P.T)
    want.sort_indices()
    assert np.array_equal(ptr, want.indptr) and np.array_equal(col, want.indices) and np.array_equal(val, want.data)
    fresh = S.create("HIP", "")
    fresh.set_parameters({"HIP": {"reorder": 1, "tolerance": 1e-9}})
    fresh.factorize(M3)
    assert all(np.array_equal(a, b_) for a, b_ in zip(fresh.matrix_to_host(), (ptr, col, val)))
    x3m, xf = np.zeros(A.n), np.zeros(A.n)
    s.solve(b, x3m)
    fresh.solve(b, xf)
    assert np.array_equal(x3m, xf) and np.linalg.norm(M3 @ x3m - b) <= 1e-8 * np.linalg.norm(b)
    # another pattern: a new search
    A3 = _shuffled(oracle, oracle.poisson7(10, 10, 10), 8)
    s.analyze_pattern(A3.to_scipy(), A3.n)
    s.factorize(A3.to_scipy())
    p3, act3 = s.reorder_perm()
    o3, _ = oracle.cuthill_mckee(A3, reverse=True)
    assert act3 and np.array_equal(o3[p3], np.arange(A3.n))
    # off again: the caller's numbering, bit-equal to the oracle's loop
    s.set_parameters({"HIP": {"reorder": 0}})
    s.factorize(A3.to_scipy())
    assert s.reorder_perm() == (None, False)
    b3 = oracle.spmv(A3, oracle.splitmix_vector(A3.n, 42))
    x3 = np.zeros(A3.n)
    s.solve(b3, x3)
    xo, ito, _ = oracle.cg_eigen(A3, b3, tol=1e-9)
    assert abs(s.get_info()["solver_iter"] - ito) <= 1 and np.abs(x3 - xo).max() <= 1e-6 * np.abs(xo).max()
    assert it0 > 0


def test_auto_mode_leaves_a_grid_alone_and_reorders_a_shuffle(S, oracle):
    nat = oracle.poisson7(24, 24, 24)
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 2, "reorder_min_rows": 0}})
    s.analyze_pattern(nat.to_scipy(), nat.n)
    s.factorize(nat.to_scipy())
    assert s.get_param("reorder.active") == 0 and s.get_param("reorder.spread_before") < 2.0
    assert s.get_param("spmv_patterns") > 0  # the structured grid keeps its pattern dictionary
    shuf = _shuffled(oracle, nat, 3)
    s.analyze_pattern(shuf.to_scipy(), shuf.n)
    s.factorize(shuf.to_scipy())
    before, after = s.get_param("reorder.spread_before"), s.get_param("reorder.spread_after")
    assert s.get_param("reorder.active") == 1 and before > 4.0 and after < 0.6 * before
    b = oracle.spmv(shuf, oracle.splitmix_vector(shuf.n, 42))
    x = np.zeros(shuf.n)
    s.solve(b, x)
    assert s.get_info()["true_residual"] < 1.5e-8


def test_amg_on_the_reordered_system_is_the_oracles_hierarchy_of_the_permuted_matrix(S, oracle):
    A = _shuffled(oracle, oracle.poisson7(20, 20, 20), 12)
    order, _ = oracle.cuthill_mckee(A, reverse=True)
    B = oracle.permuted(A, order)
    amg = {"coarse_enough": 300, "cheb_degree": 3, "cheb_power_iters": 20, "aggregation_min_rows": 0}
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 1, "precond": "amg", "tolerance": 1e-10, "max_iter": 200, "amg": amg}})
    s.analyze_pattern(A.to_scipy(), A.n)
    s.factorize(A.to_scipy())
    ref = oracle.AMG(B, coarse_enough=300, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    assert s.get_info()["amg_levels"] == ref.num_levels
    for l in range(ref.num_levels):
        assert s.amg_level_info(l)[:2] == (ref.level(l).n, ref.level(l).nnz)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    xo_new, ito, _ = oracle.cg_amgcl(B, b[order], precond=ref, tol=1e-10, max_iter=200)
    xo = np.empty(A.n)
    xo[order] = xo_new
    assert abs(s.get_info()["num_iterations"] - ito) <= 1
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    # z = M^-1 r in the caller's numbering
    r = oracle.splitmix_vector(A.n, 3)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    zo = np.empty(A.n)
    zo[order] = ref.apply(r[order])
    assert np.abs(z.download() - zo).max() <= 1e-10 * np.abs(zo).max()


def test_block3_reorder_moves_whole_nodes(S, oracle):
    E = oracle.elasticity_q1(7)
    nb = E.n // 3
    rng = np.random.default_rng(21)
    pn = rng.permutation(nb)
    dof = (3 * pn[:, None] + np.arange(3)[None, :]).ravel().astype(np.int32)
    A = oracle.permuted(E, dof)  # nodes shuffled, xyz kept together
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 1, "block_size": 3, "precond": "amg", "tolerance": 1e-9, "max_iter": 500,
                              "amg": {"coarse_enough": 200, "cheb_degree": 3, "cheb_power_iters": 20}}})
    s.analyze_pattern(A.to_scipy(), A.n)
    s.factorize(A.to_scipy())
    perm, active = s.reorder_perm()
    assert active
    assert np.array_equal(perm[0::3] % 3, np.zeros(nb)) and np.array_equal(perm[1::3], perm[0::3] + 1) and \
        np.array_equal(perm[2::3], perm[0::3] + 2)
    # the node order is the oracle's order of the node graph
    Mb = A.to_scipy()
    coo = Mb.tocoo()
    G = sp.csr_matrix((np.ones(coo.nnz), (coo.row // 3, coo.col // 3)), shape=(nb, nb))
    G.sum_duplicates()
    G.sort_indices()
    node_order, _ = oracle.cuthill_mckee(oracle.CSR.from_scipy(G), reverse=True)
    assert np.array_equal(perm[0::3] // 3, np.argsort(node_order))
    assert s.get_param("bsr3_active") == 1
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    assert s.get_info()["true_residual"] < 1.5e-9
    order = np.argsort(perm).astype(np.int32)
    B = oracle.permuted(A, order)
    ref = oracle.AMG(B, coarse_enough=200, ncycle=1, cheb_degree=3, cheb_power_iters=20, block_size=3)
    _, ito, _ = oracle.cg_amgcl(B, b[order], precond=ref, tol=1e-9, max_iter=500)
    assert abs(s.get_info()["num_iterations"] - ito) <= 1


def test_device_entry_points_and_generated_rhs(S, oracle):
    """generate_poisson7_permuted (the bench's unstructured leg) with reorder: b = A x* by the caller's row index."""
    from polysolve_amd import HIPSolver
    N = 20
    s = HIPSolver("")
    s.set_parameters({"HIP": {"reorder": 1, "tolerance": 1e-9, "profile_spmv": 4}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=7)
    n = s.matrix_shape()[0]
    t = HIPSolver("")
    t.set_parameters({"HIP": {"tolerance": 1e-9}})
    t.generate_poisson7_permuted(N, N, N, mode=1, seed=7)
    b, xs = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b, xs)
    b2, xs2 = t.device_array(n), t.device_array(n)
    t.generate_rhs(42, b2, xs2)
    assert np.array_equal(xs.download(), xs2.download())
    assert np.abs(b.download() - b2.download()).max() <= 1e-13 * np.abs(b2.download()).max()
    x = s.device_array(n)
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    i = s.get_info()
    assert i["true_residual"] < 1.5e-9 and np.abs(x.download() - xs.download()).max() < 1e-6
    assert s.info_struct().spmv_samples > 0
    x2 = t.device_array(n)
    t.axpby_device(n, 0.0, b2, 0.0, x2)
    t.solve_device(b2, x2)
    assert abs(t.get_info()["solver_iter"] - i["solver_iter"]) <= 2
    assert s.time_spmv(b, x, reps=3) > 0


def test_default_is_auto_at_scale(S, oracle):
    """The defaults: a scattered numbering of a large system is renumbered under Jacobi / identity (PCG's iterates do
    not depend on the numbering) and amg (the hierarchy of the renumbered matrix), never under a preconditioner whose
    definition IS the numbering (ic, schwarz), never on small systems (every parity test against the oracle runs in
    the caller's numbering)."""
    from polysolve_amd import HIPSolver
    N = 64  # 262 144 rows >= reorder_min_rows
    s = HIPSolver("")
    assert s.get_param("reorder") == 2 and s.get_param("reorder_min_rows") == 131072
    s.set_parameters({"HIP": {"tolerance": 1e-8}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=3)
    assert s.get_param("reorder.active") == 1 and s.get_param("reorder.spread_after") < 2.5
    n = s.matrix_shape()[0]
    perm, _ = s.reorder_perm()
    assert np.array_equal(np.sort(perm), np.arange(n))  # a bijection at scale
    b, xs, x = s.device_array(n), s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b, xs)
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    it_r = s.get_info()["solver_iter"]
    assert s.get_info()["true_residual"] < 1.5e-8 and np.abs(x.download() - xs.download()).max() < 1e-5
    s.set_parameters({"HIP": {"reorder": 0}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=3)
    assert s.get_param("reorder.active") == 0
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    assert abs(s.get_info()["solver_iter"] - it_r) <= 2  # the same Krylov iterates up to rounding
    s.set_parameters({"HIP": {"reorder": 2, "precond": "amg", "amg": {"cheb_degree": 2, "cheb_power_iters": 20}}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=3)
    assert s.get_param("reorder.active") == 1
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    assert s.get_info()["true_residual"] < 1.5e-8 and s.get_info()["num_iterations"] < it_r / 4
    s.set_parameters({"HIP": {"precond": "schwarz"}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=3)
    assert s.get_param("reorder.active") == 0
    s.set_parameters({"HIP": {"precond": "jacobi"}})
    s.generate_poisson7(N)  # the grid's own numbering: nothing to gain, the dictionary stays
    assert s.get_param("reorder.active") == 0 and s.get_param("spmv_patterns") > 0
    t = HIPSolver("")
    t.generate_poisson7_permuted(40, 40, 40, mode=1, seed=3)  # 64 000 rows: small
    assert t.get_param("reorder.active") == 0


@pytest.mark.parametrize("kind", ["laplace", "elasticity"])
def test_unstructured_tet_mesh_parity(S, oracle, kind):
    """What PolyFEM hands over: P1 stiffness matrices of a Delaunay tetrahedralisation (tests/mesh_utils.py), hull nodes
    clamped by identity rows (FEMSolver.cpp:136-161), nodes in a random order.  In the caller's numbering the solve is the
    oracle's (Jacobi: Eigen's recurrence; amg: AMGCL's, block 3 for elasticity); renumbered, it is the oracle's solve of
    the permuted system, and the order is the oracle's order of the (node) graph."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mesh_utils as mu
    P, T, bd = mu.tet_mesh(13, seed=3)
    b3 = 3 if kind == "elasticity" else 1
    K = mu.p1_laplace(P, T, bd) if b3 == 1 else mu.p1_elasticity(P, T, bd)
    K, _ = mu.renumber_nodes(K, b3, seed=4)
    A = oracle.CSR.from_scipy(K)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    amg = {"coarse_enough": 150, "cheb_degree": 3, "cheb_power_iters": 20, "aggregation_min_rows": 0}
    for reorder in (0, 1):
        s = S.create("HIP", "")
        s.set_parameters({"HIP": {"reorder": reorder, "block_size": b3, "tolerance": 1e-9, "max_iter": 3000}})
        s.analyze_pattern(K, A.n)
        s.factorize(K)
        perm, active = s.reorder_perm()
        assert active == bool(reorder)
        order = np.argsort(perm).astype(np.int32) if active else np.arange(A.n, dtype=np.int32)
        B = oracle.permuted(A, order) if active else A
        if active:  # the oracle's order of the node graph
            coo = K.tocoo()
            G = sp.csr_matrix((np.ones(coo.nnz), (coo.row // b3, coo.col // b3)), shape=(A.n // b3, A.n // b3))
            G.sum_duplicates()
            G.sort_indices()
            node_order, info = oracle.cuthill_mckee(oracle.CSR.from_scipy(G), reverse=True)
            assert np.array_equal(order[0::b3] // b3, node_order) and info["isolated"] == int(bd.sum())
        x = np.zeros(A.n)
        s.solve(b, x)
        xo_new, ito, _ = oracle.cg_eigen(B, b[order], tol=1e-9, max_iter=3000)
        xo = np.empty(A.n)
        xo[order] = xo_new
        i = s.get_info()
        assert abs(i["solver_iter"] - ito) <= 1 and np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max() and i["true_residual"] < 1.5e-9
        s.set_parameters({"HIP": {"precond": "amg", "amg": amg}})
        s.factorize(K)
        ref = oracle.AMG(B, coarse_enough=150, ncycle=1, cheb_degree=3, cheb_power_iters=20, block_size=b3)
        assert s.get_info()["amg_levels"] == ref.num_levels
        xa = np.zeros(A.n)
        s.solve(b, xa)
        _, ita, _ = oracle.cg_amgcl(B, b[order], precond=ref, tol=1e-9, max_iter=3000)
        ia = s.get_info()
        assert abs(ia["num_iterations"] - ita) <= 1 and ia["true_residual"] < 1.5e-9 and ia["num_iterations"] < i["solver_iter"] / 2
        assert np.abs(xa - xo).max() <= 1e-6 * np.abs(xo).max()


@pytest.mark.parametrize("block", [1, 3])
def test_gather_spread_is_the_counted_figure(S, oracle, block):
    """The auto criterion is integer work too: distinct lines of eight consecutive unknowns (nodes: col // block) touched by
    the first 64 entries of each of 64 consecutive rows, over ceil(distinct unknowns / 8), summed over the groups --
    restated in numpy (every group is sampled while there are at most 4096 of them)."""
    A = oracle.elasticity_q1(8) if block == 3 else oracle.poisson7(17, 13, 9)
    nb = A.n // block
    pn = np.random.default_rng(9).permutation(nb)
    dof = (block * pn[:, None] + np.arange(block)[None, :]).ravel().astype(np.int32)
    A = oracle.permuted(A, dof)
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 1, "block_size": block}})
    s.analyze_pattern(A.to_scipy(), A.n)
    s.factorize(A.to_scipy())

    def spread(M):
        ideal = lines = 0
        for g0 in range(0, M.n, 64):
            cols = np.concatenate([M.col[M.rowptr[r]:min(M.rowptr[r + 1], M.rowptr[r] + 64)] for r in range(g0, min(g0 + 64, M.n))])
            nodes = np.unique(cols // block)
            ideal += (len(nodes) + 7) // 8
            lines += len(np.unique(nodes >> 3))
        return lines / ideal

    perm, _ = s.reorder_perm()
    assert s.get_param("reorder.spread_before") == spread(A)
    assert s.get_param("reorder.spread_after") == spread(oracle.permuted(A, np.argsort(perm).astype(np.int32)))


def test_golden_unstructured_fixture(S, golden_dir):
    """The committed fixture tests/golden/reorder_tets.npz: the device's order is the stored one (made by the oracle and
    checked against scipy's breadth-first search when the fixture was generated), the renumbered Jacobi / AMG solves take
    the stored iteration counts (+-1) and reach scipy's exact solution."""
    import json
    import os
    g = np.load(os.path.join(golden_dir, "reorder_tets.npz"))
    n = int(g["n"])
    M = sp.csr_matrix((g["val"], g["col"], g["rowptr"]), shape=(n, n))
    s = S.create("HIP", "")
    # (the fixture holds the breadth-first order and the iteration counts of THAT numbering: "reorder_reverse" off)
    s.set_parameters({"HIP": {"reorder": 1, "reorder_reverse": False, "tolerance": 1e-9, "max_iter": 2000}})
    s.analyze_pattern(M, n)
    s.factorize(M)
    perm, active = s.reorder_perm()
    assert active and np.array_equal(np.argsort(perm), g["order"]) and s.get_param("reorder.levels") == int(g["levels"])
    assert s.get_param("reorder.isolated") == int(g["isolated"])
    x = np.zeros(n)
    s.solve(g["b"], x)
    assert abs(s.get_info()["solver_iter"] - int(g["cg_jacobi_iters"])) <= 1
    assert np.abs(x - g["x_exact"]).max() <= 1e-7 * np.abs(g["x_exact"]).max()
    prm = json.loads(str(g["amg_params"]))
    s.set_parameters({"HIP": {"precond": "amg", "amg": dict(prm, aggregation_min_rows=0)}})
    s.factorize(M)
    xa = np.zeros(n)
    s.solve(g["b"], xa)
    i = s.get_info()
    assert i["amg_levels"] == int(g["amg_levels"]) and abs(i["num_iterations"] - int(g["cg_amg_iters"])) <= 1
    assert np.abs(xa - g["x_exact"]).max() <= 1e-7 * np.abs(g["x_exact"]).max()


def test_adopted_device_arrays_are_renumbered_into_a_copy(S, oracle):
    """psolve_hip_factorize_device (the caller's arrays already on the device, adopted without a copy): the renumbered
    matrix is a copy of the handle's, the caller's arrays are never written, and a second factorize of the same arrays
    with new values keeps the order."""
    from polysolve_amd import HIPSolver
    A = _shuffled(oracle, oracle.poisson7(15, 11, 10), 13)
    s = HIPSolver("")
    s.set_parameters({"HIP": {"reorder": 1, "tolerance": 1e-9}})
    ptr, col = s.to_device(A.rowptr), s.to_device(np.concatenate([A.col, np.zeros(4, np.int32)]))
    val = s.to_device(np.concatenate([A.val, np.zeros(4)]))
    s.factorize_device(A.n, A.nnz, ptr, col, val)
    assert s.get_param("reorder.active") == 1
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    db, dx = s.to_device(b), s.to_device(np.zeros(A.n))
    s.solve_device(db, dx)
    xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-9)
    assert abs(s.get_info()["solver_iter"] - ito) <= 2 and np.abs(dx.download() - xo).max() <= 1e-6 * np.abs(xo).max()
    assert np.array_equal(col.download()[:A.nnz], A.col) and np.array_equal(val.download()[:A.nnz], A.val)
    assert np.array_equal(ptr.download(), A.rowptr)
    val.upload(np.concatenate([3.0 * A.val, np.zeros(4)]))
    s.factorize_device(A.n, A.nnz, ptr, col, val)
    dx2 = s.to_device(np.zeros(A.n))
    s.solve_device(db, dx2)
    assert np.abs(3.0 * dx2.download() - xo).max() <= 2e-6 * np.abs(xo).max()


@pytest.mark.parametrize("case", ["one_by_one", "diagonal", "path3", "star"])
def test_tiny_and_degenerate_graphs(S, oracle, case):
    """The smallest inputs the boundary admits (the reference's tests start at 1 x 1 right-hand sides in spirit:
    tests/test_linear_solver.cpp builds 10 x 10 systems) under a forced renumbering: a single unknown, a diagonal matrix
    (every row isolated: no search at all), a path, a star (one level as wide as the graph)."""
    if case == "one_by_one":
        M = sp.csr_matrix(np.array([[2.0]]))
    elif case == "diagonal":
        M = sp.diags(np.arange(1.0, 8.0)).tocsr()
    elif case == "path3":
        M = sp.csr_matrix(np.array([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]]))
    else:
        n = 700
        M = sp.lil_matrix((n, n))
        M[0, 1:] = -1.0
        M[1:, 0] = -1.0
        M.setdiag(np.full(n, float(n)))
        M = M.tocsr()
    M.sort_indices()
    A = oracle.CSR.from_scipy(M)
    order, info = oracle.cuthill_mckee(A, reverse=True)
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"reorder": 1, "tolerance": 1e-12}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    perm, active = s.reorder_perm()
    assert active and np.array_equal(np.argsort(perm), order)
    assert s.get_param("reorder.levels") == info["levels"] and s.get_param("reorder.isolated") == info["isolated"]
    b = M @ np.arange(1.0, A.n + 1.0)
    x = np.zeros(A.n)
    s.solve(b, x)
    assert np.abs(x - np.arange(1.0, A.n + 1.0)).max() <= 1e-9 * A.n
