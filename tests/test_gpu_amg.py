"""GPU parity tests of the Chebyshev-smoothed aggregation AMG preconditioner against the CPU oracle's
restatement of the reference configuration (AMGCL.cpp:32-65)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

AMGCL_LIKE = dict(ncycle=2, cheb_degree=16, cheb_power_iters=100)  # the reference's W-cycle / degree 16


def _solver(S, M, amg, tol=1e-10, max_iter=1000, block_size=1, extra=None):
    s = S.create("HIP", "")
    s.set_parameters({"HIP": dict(extra or {}, precond="amg", tolerance=tol, max_iter=max_iter, block_size=block_size,
                                  amg=amg)})
    s.analyze_pattern(M, M.shape[0])
    s.factorize(M)
    return s


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


@pytest.mark.parametrize("case,ce", [("poisson12", 50), ("gr3030", 100), ("elasticity", 60), ("poisson_ragged", 30)])
@pytest.mark.parametrize("cfg", [AMGCL_LIKE, dict(ncycle=1, cheb_degree=3, cheb_power_iters=20)])
@pytest.mark.parametrize("kernel", ["auto", "dma-nt", "pipe", "sell", "pat"])  # every epilogue of the four product kernels
def test_vcycle_apply_matches_oracle(S, oracle, case, ce, cfg, kernel):
    A = {"poisson12": lambda: oracle.poisson7(12), "gr3030": oracle.gr_30_30,
         "elasticity": lambda: oracle.elasticity_q1(5), "poisson_ragged": lambda: oracle.poisson7(13, 7, 9)}[case]()
    ref = oracle.AMG(A, coarse_enough=ce, **cfg)
    # hand over exactly the arrays the oracle sees (the Q1 matrix is symmetric only to rounding, and
    # with eps_strong = 0 an entry that is 0 on one side and 1e-19 on the other changes the aggregates)
    extra = {"auto": {}, "dma-nt": {"spmv_kernel": 1, "spmv_nt": 1}, "pipe": {"spmv_kernel": 0},
             "sell": {"spmv_kernel": 2}, "pat": {"spmv_kernel": 3}}[kernel]
    s = _solver(S, A.to_scipy(), dict(coarse_enough=ce, sell=2 if kernel == "sell" else 0, **cfg), extra=extra)
    info = s.get_info()
    assert info["amg_levels"] == ref.num_levels
    for l in range(ref.num_levels):
        rows, nnz, rho = s.amg_level_info(l)
        assert (rows, nnz) == (ref.level(l).n, ref.level(l).nnz)
        assert np.isclose(rho, ref.level_scalars(l)["rho"], rtol=1e-9)
    r = oracle.splitmix_vector(A.n, 17)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    zo = ref.apply(r)
    assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo)


@pytest.mark.parametrize("name", ["poisson7_n8", "poisson7_n12", "poisson7_6x5x7", "gr_30_30", "elasticity_q1_m5"])
def test_amg_pcg_golden_parity(S, oracle, golden_dir, name):
    """AMGCL-default configuration: same hierarchy, same iteration count as the oracle's
    amgcl::solver::cg run that produced the fixture, solution equal to scipy's."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n = int(g["n"])
    M = sp.csr_matrix((g["val"], g["col"], g["rowptr"]), shape=(n, n))
    params = json.loads(str(g["amg_params"]))
    s = _solver(S, M, dict(params, **AMGCL_LIKE))
    assert s.get_info()["amg_levels"] == int(g["amg_levels"])
    x = np.zeros(n)
    s.solve(g["b"], x)
    info = s.get_info()
    assert abs(info["num_iterations"] - int(g["cg_amg_iters"])) <= 1
    assert np.linalg.norm(x - g["x_exact"]) / np.linalg.norm(g["x_exact"]) < 1e-8
    assert np.linalg.norm(M @ x - g["b"]) / np.linalg.norm(g["b"]) < 1e-7  # test_linear_solver.cpp:600-601


def test_amg_reference_inequalities(S, oracle):
    """`all` (:160-162) and `amgcl_initial_guess` (:400-455) with the AMG preconditioner."""
    A = oracle.poisson7(14)
    M = A.to_scipy().tocsc()
    b = np.random.default_rng(5).uniform(-1, 1, A.n)
    s = _solver(S, M, dict(coarse_enough=100))
    x = np.zeros(A.n)
    s.solve(b, x)
    assert s.get_info()["num_iterations"] > 0
    assert np.linalg.norm(M @ x - b) < 1e-8
    s2 = _solver(S, M, dict(coarse_enough=100), tol=2e-10)
    s2.solve(b, x)
    assert s2.get_info()["num_iterations"] == 0
    assert np.linalg.norm(M @ x - b) < 1e-8


@pytest.mark.parametrize("cfg", [dict(ncycle=1, cheb_degree=2), dict(ncycle=1, cheb_degree=4), dict(ncycle=2, cheb_degree=16)])
def test_amg_pcg_vs_oracle_midsize(S, oracle, cfg):
    """40^3 Poisson: the V-cycle variants the bench uses, iteration count and solution vs the oracle."""
    A = oracle.poisson7(40)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    cfg = dict(cfg, coarse_enough=500, cheb_power_iters=30)
    ref = oracle.AMG(A, **cfg)
    xo, ito, erro = oracle.cg_amgcl(A, b, precond=ref, tol=1e-8, max_iter=500)
    s = _solver(S, A.to_scipy(), cfg, tol=1e-8, max_iter=500)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert abs(info["num_iterations"] - ito) <= 1
    assert info["true_residual"] < 1.5e-8
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()


def test_amg_elasticity_config3_small(S, oracle):
    """BASELINE.json configs[2] in miniature: block-3 Q1 elasticity, Chebyshev-AMG PCG (scalar AMG)."""
    A = oracle.elasticity_q1(14)
    M = A.to_scipy()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    s = _solver(S, M, dict(coarse_enough=300, ncycle=1, cheb_degree=4, cheb_power_iters=30), tol=1e-8)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert info["solver_status"] == "Reach relative tolerance"
    assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1.5e-8
    sj = S.create("HIP", "")
    sj.factorize(M)
    xj = np.zeros(A.n)
    sj.solve(b, xj)
    assert info["num_iterations"] < sj.get_info()["num_iterations"] / 3  # AMG must beat Jacobi clearly


@pytest.mark.parametrize("cfg", [AMGCL_LIKE, dict(ncycle=1, cheb_degree=3, cheb_power_iters=20)])
def test_block3_vcycle_and_pcg_match_oracle(S, oracle, cfg):
    """AMGCL_Block<3> (AMGCL.cpp:243-302): block aggregation, block-smoothed P, block-Jacobi-scaled
    Chebyshev -- hierarchy, V-cycle action and PCG iteration count against the oracle."""
    A = oracle.elasticity_q1(9)
    M = A.to_scipy()
    ref = oracle.AMG(A, coarse_enough=200, block_size=3, **cfg)
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "block_size": 3, "tolerance": 1e-9, "max_iter": 500,
                              "amg": dict(coarse_enough=200, **cfg)}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    assert s.get_info()["amg_levels"] == ref.num_levels >= 2
    for l in range(ref.num_levels):
        rows, nnz, rho = s.amg_level_info(l)
        assert rows == ref.level(l).n
        if l > 0:
            assert nnz == ref.level(l).nnz
        assert np.isclose(rho, ref.level_scalars(l)["rho"], rtol=1e-9)
    r = oracle.splitmix_vector(A.n, 23)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    zo = ref.apply(r)
    assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-9, max_iter=500)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert abs(info["num_iterations"] - ito) <= 1
    assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1e-7  # test_linear_solver.cpp:663-664
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    # scalar vs block on the same system (the reference's crystm03 scalar/block-3 comparison, :604-665):
    # both reach the tolerance; the block hierarchy keeps ~3x more coarse unknowns
    s1 = S.create("HIP", "")
    s1.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-9, "max_iter": 500, "amg": dict(coarse_enough=200, **cfg)}})
    s1.factorize(M)
    x1 = np.zeros(A.n)
    s1.solve(b, x1)
    assert np.linalg.norm(M @ x1 - b) / np.linalg.norm(b) < 1e-7
    assert s.amg_level_info(1)[0] > 2 * s1.amg_level_info(1)[0]


def test_block_size_must_divide(S, oracle):
    A = oracle.poisson7(5)  # 125 rows
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "block_size": 3}})
    with pytest.raises(RuntimeError, match="block_size"):
        s.factorize(A.to_scipy())


@pytest.mark.parametrize("sell", [0, 2])  # 2: every operator on its SELL copy, refilled (not rebuilt) by a refresh
def test_amg_numeric_refresh_on_same_pattern(S, oracle, sell):
    """factorize() again with the SAME sparsity pattern and new values (what Newton does every
    iteration, Newton.cpp:189-193, and the reference's `pre_factor` test, :241-307): aggregates and all
    patterns are kept, omega / P / R / A P / R A P and the smoothers are recomputed by device kernels.
    The refreshed hierarchy must act like a from-scratch oracle setup on the new matrix."""
    cfg = dict(coarse_enough=60, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    base = oracle.poisson7(11, 9, 10)
    S0 = base.to_scipy()
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-10, "amg": dict(cfg, sell=sell),
                              "spmv_kernel": 2 if sell else -1}})
    s.analyze_pattern(S0, base.n)
    s.factorize(S0)
    assert s.get_param("amg.last_setup_reused") == 0
    rng = np.random.default_rng(42)
    for k in range(3):
        U = sp.triu(S0, k=1).tocoo()
        off = -rng.uniform(0.1, 5, U.nnz)
        U = sp.coo_matrix((off, (U.row, U.col)), shape=S0.shape)
        Mk = (U + U.T + sp.diags(rng.uniform(0.1, 5, S0.shape[0]) * 100)).tocsr()
        Mk.sort_indices()
        s.factorize(Mk)
        assert s.get_param("amg.last_setup_reused") == 1
        Ak = oracle.CSR.from_scipy(Mk)
        ref = oracle.AMG(Ak, **cfg)
        assert s.get_info()["amg_levels"] == ref.num_levels
        for l in range(ref.num_levels):
            rows, nnz, rho = s.amg_level_info(l)
            assert (rows, nnz) == (ref.level(l).n, ref.level(l).nnz)
            assert np.isclose(rho, ref.level_scalars(l)["rho"], rtol=1e-9)
        r = oracle.splitmix_vector(Ak.n, 3 + k)
        z = s.device_array(Ak.n)
        s.precond_apply_device(s.to_device(r), z)
        zo = ref.apply(r)
        assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo)
        b = rng.uniform(-1, 1, Ak.n)
        x = np.zeros(Ak.n)
        s.solve(b, x)
        assert np.linalg.norm(Mk @ x - b) < 1e-8  # pre_factor's assertion
        xo, ito, _ = oracle.cg_amgcl(Ak, b, precond=ref, tol=1e-10)
        assert abs(s.get_info()["num_iterations"] - ito) <= 1
    # a different pattern (or reuse switched off) goes through the full setup again
    other = oracle.poisson7(8).to_scipy()
    s.factorize(other)
    assert s.get_param("amg.last_setup_reused") == 0
    s.set_parameters({"HIP": {"amg": {"reuse": 0}}})
    s.factorize(other)
    assert s.get_param("amg.last_setup_reused") == 0


@pytest.mark.parametrize("cfg", [dict(ncycle=2, npre=2, npost=1, cheb_degree=2, cheb_power_iters=10),
                                 dict(ncycle=1, npre=0, npost=2, cheb_degree=4, cheb_power_iters=10),
                                 dict(ncycle=1, max_levels=2, cheb_degree=3, cheb_power_iters=10),
                                 dict(ncycle=1, eps_strong=0.08, cheb_degree=3, cheb_power_iters=10),
                                 dict(ncycle=1, estimate_spectral_radius=0, cheb_degree=3, cheb_power_iters=0)])
def test_cycle_variants_match_oracle(S, oracle, cfg):
    """W-cycle, asymmetric pre/post sweeps, a truncated hierarchy, a strength filter, omega = 2/3 and the
    Gershgorin smoother bound (power_iters 0): the device cycle against the oracle's amg::apply."""
    A = oracle.poisson7(13, 11, 12)
    cfg = dict(cfg, coarse_enough=40)
    ref = oracle.AMG(A, **cfg)
    s = _solver(S, A.to_scipy(), cfg, tol=1e-9)
    assert s.get_info()["amg_levels"] == ref.num_levels
    r = oracle.splitmix_vector(A.n, 31)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    zo = ref.apply(r)
    assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-9, max_iter=500)
    assert abs(s.get_info()["num_iterations"] - ito) <= 1
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


def _arrow_spd(n, seed=3):
    """SPD matrix with one hub row/column (row 0 touches every node) on top of a ring: the hub's
    aggregate swallows thousands of rows, so R and R (A P) get rows far wider than an LDS tile."""
    rng = np.random.default_rng(seed)
    i = np.arange(1, n)
    w = rng.uniform(0.1, 1.0, n - 1)
    ring = rng.uniform(0.1, 1.0, n - 1)
    rows = np.concatenate([np.zeros(n - 1, int), i, i, np.roll(i, 1)])
    cols = np.concatenate([i, np.zeros(n - 1, int), np.roll(i, 1), i])
    vals = -np.concatenate([w, w, ring, ring])
    M = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    M.sum_duplicates()
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 0.5
    return (M + sp.diags(d)).tocsr()


def _random_graph_spd(n, deg, seed):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), deg)
    c = rng.integers(0, n, n * deg)
    keep = r != c
    v = -rng.uniform(0.05, 1.0, keep.sum())
    M = sp.coo_matrix((v, (r[keep], c[keep])), shape=(n, n)).tocsr()
    M = (M + M.T).tocsr()
    M.sum_duplicates()
    d = np.asarray(abs(M).sum(axis=1)).ravel() + 0.1
    return (M + sp.diags(d)).tocsr()


@pytest.mark.parametrize("case", ["poisson", "poisson_eps", "ragged", "elasticity_scalar", "random_wide", "arrow",
                                  "gr3030_two_levels", "elasticity_block3", "elasticity_block3_eps", "gr3030_block2",
                                  "random_block3", "random_wide_hash", "arrow_hash"])
def test_device_setup_equals_host_hierarchy(S, oracle, case):
    """The hierarchy coarsened on the device (strength graph, row-set patterns, numeric kernels; only the
    greedy sweep on the host) is the all-host construction bit for bit: every A_l, P_l, R_l.  Wide rows of products with
    few columns take the LDS bitmap of amg_symbolic.hip; the "_hash" cases switch it off on their handle ("lab.symbolic_bitmap")
    so that the 256-lane and the HBM hash sets still see those rows."""
    hash_only = case.endswith("_hash")
    if hash_only:
        case = case[:-5]
    _device_setup_equals_host_hierarchy(S, oracle, case, {"lab.symbolic_bitmap": 0} if hash_only else None)


def _device_setup_equals_host_hierarchy(S, oracle, case, extra=None):
    from polysolve_amd import HostHierarchy
    amg = dict(coarse_enough=40, max_levels=5, aggregation_min_rows=0)  # the sweep as dependency rounds on the device
    bs = 1
    if case == "poisson":
        M = oracle.poisson7(24).to_scipy()
    elif case == "poisson_eps":
        M = oracle.poisson7(18, 11, 14).to_scipy()
        M = (M + sp.diags(np.linspace(0.0, 3.0, M.shape[0]))).tocsr()
        amg["eps_strong"] = 0.08
    elif case == "ragged":
        M = oracle.poisson7(13, 7, 9).to_scipy()
    elif case == "elasticity_scalar":
        M = oracle.elasticity_q1(8).to_scipy()  # 81 entries per row: the 64- and 256-lane tiers
        amg["coarse_enough"] = 20
    elif case == "random_wide":
        M = _random_graph_spd(6000, 40, 11)  # ~80 entries per row, unstructured: wide coarse rows
        amg["coarse_enough"] = 10
    elif case == "arrow":
        M = _arrow_spd(20000)  # hub: rows beyond every LDS tier (HBM hash set, HBM sort)
        amg["coarse_enough"] = 10
    elif case.startswith("elasticity_block3"):
        M = oracle.elasticity_q1(9).to_scipy()  # AMGCL_Block<3>: aggregation and smoothing on 3x3 blocks
        amg["coarse_enough"] = 60
        bs = 3
        if case.endswith("eps"):
            amg["eps_strong"] = 0.05
    elif case == "random_block3":
        # an unstructured block-3 operator with ~40 blocks per block row: coarse rows of a hundred blocks and more -- the
        # Galerkin products on blocks (amg_bspgemm.hip) park their output rows in several passes and meet rows of B longer
        # than their stage
        G = _random_graph_spd(1500, 20, 5)
        T = sp.csr_matrix(np.array([[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 1.0]]))
        M = sp.kron(G, T, format="csr")
        amg["coarse_enough"] = 10
        bs = 3
    elif case == "gr3030_block2":
        M = oracle.gr_30_30().to_scipy()  # the reference's block-2 run of gr_30_30 (test_linear_solver.cpp:541-602)
        amg["coarse_enough"] = 50
        bs = 2
    else:
        M = oracle.gr_30_30().to_scipy()
        amg["coarse_enough"] = 100
    M = sp.csr_matrix(M)
    M.sort_indices()
    n = M.shape[0]
    host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=amg["max_levels"],
                         coarse_enough=amg["coarse_enough"], eps_strong=amg.get("eps_strong", 0.0), block_size=bs)
    s = _solver(S, M, dict(amg, cheb_power_iters=5), block_size=bs, extra=extra)
    assert s.get_param("amg.device_setup") == 1
    assert s.get_param("amg.levels_aggregated_on_device") == host.num_levels - 1  # every coarsened level
    assert s.get_info()["amg_levels"] == host.num_levels
    assert host.num_levels >= 2
    for l in range(host.num_levels):
        for what, w in (("A", 0), ("P", 1), ("R", 2)):
            h = host.level(l, what)
            if h is None:
                assert l == host.num_levels - 1 and what != "A"
                continue
            shape, ptr, col, val = s.amg_level_matrix(l, w)
            assert shape == (h[0], h[1]), (case, l, what)
            assert np.array_equal(ptr, h[2]), (case, l, what)
            assert np.array_equal(col, h[3]), (case, l, what)
            assert np.array_equal(val, h[4]), (case, l, what)
    # and the all-host path is still selectable
    s0 = _solver(S, M, dict(amg, cheb_power_iters=5, device_setup=0), block_size=bs)
    assert s0.get_info()["amg_levels"] == host.num_levels
    b = np.ones(n)
    x, x0 = np.zeros(n), np.zeros(n)
    s.solve(b, x)
    s0.solve(b, x0)
    assert s.get_info()["num_iterations"] == s0.get_info()["num_iterations"]
    assert np.array_equal(x, x0)


def test_block3_numeric_refresh_on_same_pattern(S, oracle):
    """Newton on an elastic problem: the same block pattern, new values, at every factorize.  The block
    hierarchy is refreshed by kernels (block values, block-smoothed P, Galerkin products) and must equal a
    from-scratch setup on the new matrix; when the new values flip a strength flag the refresh is refused
    and the hierarchy is rebuilt."""
    from polysolve_amd import HostHierarchy
    A = oracle.elasticity_q1(8)
    M0 = sp.csr_matrix(A.to_scipy())
    M0.sort_indices()
    n = M0.shape[0]
    amg = dict(coarse_enough=60, ncycle=1, cheb_degree=3, cheb_power_iters=10)
    s = _solver(S, M0, amg, block_size=3)
    assert s.get_param("amg.last_setup_reused") == 0
    rng = np.random.default_rng(7)
    for k in range(2):
        # SPD perturbation that keeps the pattern: D M0 D with a smooth positive diagonal D (a "stiffness change")
        d = 1.0 + 0.3 * rng.uniform(0, 1, n // 3).repeat(3)
        Mk = M0.copy()  # scaled in place: the Q1 matrix stores explicit zeros, which scipy products would drop
        rows = np.repeat(np.arange(n), np.diff(M0.indptr))
        Mk.data = M0.data * d[rows] * d[M0.indices]
        s.factorize(Mk)
        assert s.get_param("amg.last_setup_reused") == 1
        host = HostHierarchy(n, Mk.indptr, Mk.indices, Mk.data, coarse_enough=60, block_size=3)
        assert s.get_info()["amg_levels"] == host.num_levels
        for l in range(host.num_levels):
            for what, w in (("A", 0), ("P", 1), ("R", 2)):
                h = host.level(l, what)
                if h is None:
                    continue
                shape, ptr, col, val = s.amg_level_matrix(l, w)
                assert np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]), (k, l, what)
                assert np.array_equal(val, h[4]), (k, l, what)
        b = rng.uniform(-1, 1, n)
        x = np.zeros(n)
        s.solve(b, x)
        assert np.linalg.norm(Mk @ x - b) / np.linalg.norm(b) < 1e-8
    # flip one strength flag: an off-diagonal block whose entries become explicit zeros (same pattern,
    # trace(A_ij A_ij) = 0 -> weak): the refresh is refused and the hierarchy rebuilt
    Mf = M0.copy()
    bi = 1
    cols = M0.indices[M0.indptr[3 * bi]:M0.indptr[3 * bi + 1]]
    bj = int(cols[cols >= 3 * (bi + 1)][0]) // 3
    for (p_, q_) in ((bi, bj), (bj, bi)):
        for r in range(3):
            lo, hi = Mf.indptr[3 * p_ + r], Mf.indptr[3 * p_ + r + 1]
            sel = (Mf.indices[lo:hi] // 3) == q_
            Mf.data[lo:hi][sel] = 0.0
    s.factorize(M0)
    assert s.get_param("amg.last_setup_reused") == 1
    s.factorize(Mf)
    assert s.get_param("amg.last_setup_reused") == 0
    x = np.zeros(n)
    b = np.ones(n)
    s.solve(b, x)
    assert np.linalg.norm(Mf @ x - b) / np.linalg.norm(b) < 1e-8


@pytest.mark.parametrize("M,ce,cfg", [(9, 200, dict(ncycle=1, cheb_degree=2, cheb_power_iters=20)),
                                      (16, 300, dict(ncycle=1, cheb_degree=3, cheb_power_iters=20)),
                                      (20, 200, dict(ncycle=2, npre=2, npost=1, cheb_degree=2, cheb_power_iters=10)),
                                      (14, 100, dict(ncycle=1, cheb_degree=1, cheb_power_iters=10))])
def test_block_operators_of_the_cycle_match_scalar_path_and_oracle(S, oracle, M, ce, cfg):
    """"amg.block_levels" (round 4; AMGCL_Block<3>'s value type end to end, AMGCL.cpp:243-302): A_l below the finest level,
    P_l and R_l multiply through 3x3-block copies -- prolongations with ~4 blocks per block row, restrictions with ~100 --
    and the block-scaled Chebyshev step is an epilogue of the block product.  Storage and fusion only: the cycle's action
    equals the scalar-CSR cycle's (round 3's path) up to the association of the row sums and the oracle's to 1e-9; PCG
    counts +-1; after a numeric refresh (new values, same pattern) the block copies carry the new numbers."""
    A = oracle.elasticity_q1(M)
    Msp = sp.csr_matrix(A.to_scipy())
    Msp.sort_indices()
    n = A.n
    ref = oracle.AMG(A, coarse_enough=ce, block_size=3, **cfg)
    r = oracle.splitmix_vector(n, 23)
    zo = ref.apply(r)
    zs = {}
    for bl in (1, 0):
        s = _solver(S, Msp, dict(coarse_enough=ce, block_levels=bool(bl), **cfg), tol=1e-9, max_iter=500, block_size=3)
        assert s.get_info()["amg_levels"] == ref.num_levels >= 2
        z = s.device_array(n)
        s.precond_apply_device(s.to_device(r), z)
        zs[bl] = z.download()
        assert np.linalg.norm(zs[bl] - zo) <= 1e-9 * np.linalg.norm(zo), bl
        if bl == 1:
            b = oracle.spmv(A, oracle.splitmix_vector(n, 42))
            xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-9, max_iter=500)
            x = np.zeros(n)
            s.solve(b, x)
            assert abs(s.get_info()["num_iterations"] - ito) <= 1
            assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
            # numeric refresh: D A D with a smooth positive diagonal D, constant per node
            rng = np.random.default_rng(M)
            d = 1.0 + 0.3 * rng.uniform(0, 1, n // 3).repeat(3)
            Mk = Msp.copy()
            rows = np.repeat(np.arange(n), np.diff(Msp.indptr))
            Mk.data = Msp.data * d[rows] * d[Msp.indices]
            s.factorize(Mk)
            assert s.get_param("amg.last_setup_reused") == 1
            refk = oracle.AMG(oracle.CSR.from_scipy(Mk), coarse_enough=ce, block_size=3, **cfg)
            s.precond_apply_device(s.to_device(r), z)
            zk = refk.apply(r)
            assert np.linalg.norm(z.download() - zk) <= 1e-9 * np.linalg.norm(zk)
    assert np.linalg.norm(zs[1] - zs[0]) <= 1e-12 * np.linalg.norm(zs[0])


def test_gr_30_30_scalar_vs_block2(S, oracle):
    """The reference's `gr_30_30` test (test_linear_solver.cpp:541-602): the 900 x 900 9-point Laplacian
    solved with the scalar backend and with block_size 2, both to a relative residual < 1e-7."""
    M = sp.csr_matrix(oracle.gr_30_30().to_scipy())
    b = np.ones(M.shape[0])
    its = {}
    for bs in (1, 2):
        s = _solver(S, M, dict(coarse_enough=100, **AMGCL_LIKE), block_size=bs)
        x = np.zeros(M.shape[0])
        s.solve(b, x)
        assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1e-7
        its[bs] = s.get_info()["num_iterations"]
        assert s.get_info()["amg_levels"] >= 2
    assert its[1] > 0 and its[2] > 0


def test_device_setup_random_graphs_property(S, oracle):
    """hypothesis: random symmetric diagonally dominant matrices (random degree, isolated rows, optional
    strength filter, scalar and 3x3-block value types) -- the device-built hierarchy equals the host-built
    one: same aggregates, same patterns, same numbers, on every level."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from polysolve_amd import HostHierarchy

    @settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck))
    @given(nb=st.integers(60, 1500), deg=st.integers(1, 30), seed=st.integers(0, 2 ** 31 - 1),
           eps=st.sampled_from([0.0, 0.0, 0.1]), bs=st.sampled_from([1, 1, 3]))
    def check(nb, deg, seed, eps, bs):
        rng = np.random.default_rng(seed)
        G = _random_graph_spd(nb, deg, seed % (2 ** 31))
        if bs == 3:  # couple the three components of every node: full 3x3 blocks
            B = rng.uniform(0.5, 1.5, (3, 3))
            B = B @ B.T + 3 * np.eye(3)
            M = sp.kron(G, B, format="csr")
        else:
            M = G
        k = int(rng.integers(0, nb))  # one isolated node (no strong connection: removed from the aggregation)
        M = sp.lil_matrix(M)
        for r in range(bs):
            row = k * bs + r
            M[row, :] = 0.0
            M[:, row] = 0.0
            M[row, row] = 2.0
        M = sp.csr_matrix(M)
        M.eliminate_zeros()
        M.sort_indices()
        n = M.shape[0]
        amg = dict(coarse_enough=12, max_levels=4, eps_strong=eps, cheb_power_iters=3,
                   aggregation_rounds=bool(rng.integers(0, 2)),  # dependency rounds or the waiting kernel ...
                   aggregation_min_rows=int(rng.integers(0, 2)) * 10 ** 9)  # ... or the host sweep
        host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=4, coarse_enough=12, eps_strong=eps,
                             block_size=bs)
        s = _solver(S, M, amg, block_size=bs)
        assert s.get_info()["amg_levels"] == host.num_levels
        for l in range(host.num_levels):
            for what, w in (("A", 0), ("P", 1), ("R", 2)):
                h = host.level(l, what)
                if h is None:
                    continue
                shape, ptr, col, val = s.amg_level_matrix(l, w)
                assert shape == (h[0], h[1]) and np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]), (l, what)
                assert np.array_equal(val, h[4]), (l, what)

    check()


@pytest.mark.parametrize("rounds", [False, True])  # the waiting kernel / the dependency rounds
@pytest.mark.parametrize("n,deg,seed", [(400, 2, 1), (3000, 3, 2), (2500, 8, 3)])
def test_device_aggregation_on_unsymmetric_patterns(S, oracle, n, deg, seed, rounds):
    """Unsymmetric strength patterns (not the SPD case, but the sweep is defined for them): seeds can be claimed
    by later seeds and aggregates can empty.  The device rounds work on the transposed graph and drop the emptied
    aggregates: same hierarchy as the host sweep (no solve here, only the setup is compared)."""
    from polysolve_amd import HostHierarchy
    from test_aggregation_closed_form import _random_pattern, closed_form_aggregates
    M = _random_pattern(n, deg, seed, symmetric=False)
    M = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel())).tocsr()  # keep the Chebyshev radii finite
    M.sort_indices()
    cnt, ids = closed_form_aggregates(M)
    host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=3, coarse_enough=20)
    assert host.level(1, "A")[0] == cnt  # the closed form, the host sweep ...
    s = _solver(S, M, dict(coarse_enough=20, max_levels=3, aggregation_min_rows=0, cheb_power_iters=3,
                           aggregation_rounds=rounds))
    assert s.get_param("amg.levels_aggregated_on_device") == host.num_levels - 1
    for l in range(host.num_levels):  # ... and the device rounds agree
        for what, w in (("A", 0), ("P", 1), ("R", 2)):
            h = host.level(l, what)
            if h is None:
                continue
            shape, ptr, col, val = s.amg_level_matrix(l, w)
            assert shape == (h[0], h[1]) and np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]), (l, what)
            assert np.array_equal(val, h[4]), (l, what)


@pytest.mark.parametrize("rounds,budget", [(True, 2000), (False, 1)])
def test_device_aggregation_gives_up_on_long_chains(S, oracle, rounds, budget):
    """A 1-D chain in natural order decides three vertices per dependency round: the device rounds notice the
    pace (the waiting kernel: its time limit of 10 us per allowed round), hand the level to the host sweep, and
    the hierarchy is the same."""
    from polysolve_amd import HostHierarchy
    n = 60000
    M = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
    host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=2, coarse_enough=100)
    s = _solver(S, M, dict(coarse_enough=100, max_levels=2, aggregation_min_rows=0, aggregation_max_rounds=budget,
                           aggregation_rounds=rounds, cheb_power_iters=3))
    assert s.get_param("amg.levels_aggregated_on_device") == 0
    shape, ptr, col, val = s.amg_level_matrix(1, 0)
    h = host.level(1, "A")
    assert np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]) and np.array_equal(val, h[4])


@pytest.mark.parametrize("bs", [1, 3])
def test_amg_matrix_fp32_option(S, oracle, bs):
    """amg.matrix_fp32: the cycle's operators stream single-precision values (arithmetic in double, PCG's own
    product on the original matrix).  The preconditioner changes by ~1e-7 relative, PCG reaches the same
    tolerance in (almost) the same number of iterations, and a same-pattern refactorize keeps working."""
    A = oracle.elasticity_q1(10) if bs == 3 else oracle.poisson7(36)
    M = sp.csr_matrix(A.to_scipy())
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    cfg = dict(coarse_enough=300, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    s64 = _solver(S, M, cfg, tol=1e-9, block_size=bs)
    s32 = _solver(S, M, dict(cfg, matrix_fp32=1), tol=1e-9, block_size=bs)
    r = oracle.splitmix_vector(A.n, 5)
    z64, z32 = s64.device_array(A.n), s32.device_array(A.n)
    s64.precond_apply_device(s64.to_device(r), z64)
    s32.precond_apply_device(s32.to_device(r), z32)
    d = np.linalg.norm(z64.download() - z32.download()) / np.linalg.norm(z64.download())
    assert 0 < d < 1e-5
    x64, x32 = np.zeros(A.n), np.zeros(A.n)
    s64.solve(b, x64)
    s32.solve(b, x32)
    assert abs(s32.get_info()["num_iterations"] - s64.get_info()["num_iterations"]) <= 1
    assert np.linalg.norm(M @ x32 - b) / np.linalg.norm(b) < 1.5e-9
    s32.factorize(M)  # same pattern: refreshed by kernels, the single-precision copies are rebuilt
    assert s32.get_param("amg.last_setup_reused") == 1
    x32b = np.zeros(A.n)
    s32.solve(b, x32b)
    assert np.array_equal(x32b, x32)


@pytest.mark.parametrize("case", ["elasticity_block3", "random_block3"])
def test_block_chebyshev_step_split_equals_fused(S, oracle, case):
    """Round 6: beyond the Infinity Cache the block-scaled Chebyshev step runs as the residual product plus a node-local update
    launch instead of one product with a fused epilogue ("lab.cheb_split" -1: by operator size, 768 MiB of blocks; 1 / 0 force
    it on / off).  The same operations in the same order: the preconditioner's action and the PCG iterates are bit-equal,
    whichever form every level takes."""
    M, bs, ce = _round5_case(oracle, case)
    M = _same_pattern_spd(M, bs, np.random.default_rng(1))
    n = M.shape[0]
    amg = dict(coarse_enough=ce, max_levels=4, ncycle=1, cheb_degree=3, cheb_power_iters=20, aggregation_min_rows=0)
    out = []
    for split in (0, 1, -1):
        s = _solver(S, M, amg, tol=1e-9, block_size=bs, extra={"lab.cheb_split": split, "lab.bsr3_kinds": 0})
        r = oracle.splitmix_vector(n, 5)
        z = s.device_array(n)
        s.precond_apply_device(s.to_device(r), z)
        b = M @ oracle.splitmix_vector(n, 42)
        x = np.zeros(n)
        s.solve(b, x)
        out.append((z.download(), x, s.get_info()["num_iterations"]))
    for z, x, it in out[1:]:
        assert np.array_equal(z, out[0][0]) and np.array_equal(x, out[0][1]) and it == out[0][2]


@pytest.mark.parametrize("case", ["elasticity_block3", "random_block3"])
@pytest.mark.parametrize("cfg", [dict(), dict(relax_type="damped_jacobi"), dict(aggregation="compact", direct_coarse=1)],
                         ids=["chebyshev", "damped_jacobi", "compact-direct"])
def test_matrix_fp32_keeps_the_block_hierarchy(S, oracle, case, cfg):
    """Round 6: under amg.matrix_fp32 a block-3 hierarchy keeps its 3 x 3-block copies of every operator of the cycle (A_l, P_l,
    R_l) -- with single-precision values, 40 bytes per block, streamed by the same LDS-DMA kernel (spmv_bsr3_dma<.., float>: chunks
    start on a multiple of four blocks, the tail of the value array by hand) including the fused Chebyshev step and the
    prolongation's add epilogue.  Against the same hierarchy in double: the cycle's action differs by single-precision rounding
    of the operators only (1e-8 ... 1e-5 relative), PCG takes the same count +- 1 to the same double-precision residual; a
    structured operator (Q1 elasticity on a grid) and an unstructured one with ~40 blocks per block row (several chunks per
    group, ragged ends)."""
    M, bs, ce = _round5_case(oracle, case)
    M = _same_pattern_spd(M, bs, np.random.default_rng(1))
    n = M.shape[0]
    base = dict(coarse_enough=ce, max_levels=4, ncycle=1, cheb_degree=2, cheb_power_iters=20, aggregation_min_rows=0)
    s64 = _solver(S, M, dict(base, **cfg), tol=1e-9, block_size=bs, extra={"lab.bsr3_kinds": 0})
    s32 = _solver(S, M, dict(base, matrix_fp32=1, **cfg), tol=1e-9, block_size=bs, extra={"lab.bsr3_kinds": 0})
    assert s32.get_info()["amg_levels"] == s64.get_info()["amg_levels"] >= 2
    r = oracle.splitmix_vector(n, 5)
    z64, z32 = s64.device_array(n), s32.device_array(n)
    s64.precond_apply_device(s64.to_device(r), z64)
    s32.precond_apply_device(s32.to_device(r), z32)
    d = np.linalg.norm(z64.download() - z32.download()) / np.linalg.norm(z64.download())
    assert 1e-9 < d < 1e-5, d
    # the operations of the cycle ran on block copies in both (timed on the hierarchy's own operators: the block kernel names)
    if "relax_type" not in cfg:
        t = s32.amg_time_level_ops(0, 1)
        assert t["cheb_step_us"] > 0 and t["restrict_us"] > 0 and t["prolong_us"] > 0
    b = M @ oracle.splitmix_vector(n, 42)
    x64, x32 = np.zeros(n), np.zeros(n)
    s64.solve(b, x64)
    s32.solve(b, x32)
    assert abs(s32.get_info()["num_iterations"] - s64.get_info()["num_iterations"]) <= 1
    assert np.linalg.norm(M @ x32 - b) / np.linalg.norm(b) < 1.5e-9 and np.linalg.norm(x32 - x64) <= 1e-6 * np.linalg.norm(x64)
    Mk = _same_pattern_spd(M, bs, np.random.default_rng(3))  # Newton's refactorize: the single-precision copies are refilled
    s32.factorize(Mk)
    s64.factorize(Mk)
    assert s32.get_param("amg.last_setup_reused") == 1
    x32[:] = 0
    x64[:] = 0
    s32.solve(b, x32)
    s64.solve(b, x64)
    assert abs(s32.get_info()["num_iterations"] - s64.get_info()["num_iterations"]) <= 1
    assert np.linalg.norm(Mk @ x32 - b) / np.linalg.norm(b) < 1.5e-9


@pytest.mark.parametrize("bs", [1, 3])
def test_unsorted_column_indices(S, oracle, bs):
    """Eigen does not promise sorted inner indices (makeCompressed keeps insertion order): rows with shuffled
    columns must give the same hierarchy shape and the same solution, through the Jacobi and the AMG path."""
    A = oracle.elasticity_q1(8) if bs == 3 else oracle.poisson7(20, 17, 15)
    M = sp.csr_matrix(A.to_scipy())
    M.sort_indices()
    rng = np.random.default_rng(9)
    U = M.copy()
    for i in range(M.shape[0]):
        lo, hi = M.indptr[i], M.indptr[i + 1]
        perm = rng.permutation(hi - lo)
        U.indices[lo:hi] = M.indices[lo:hi][perm]
        U.data[lo:hi] = M.data[lo:hi][perm]
    U.has_sorted_indices = False
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    sols, its, shapes = [], [], []
    for mat in (M, U):
        s = S.create("HIP", "")
        s.set_parameters({"HIP": {"precond": "amg", "block_size": bs, "tolerance": 1e-10,
                                  "amg": dict(coarse_enough=100, ncycle=1, cheb_degree=3, cheb_power_iters=20,
                                              aggregation_min_rows=0)}})
        # hand the arrays over as they are (scipy would sort them in some conversions)
        s._check(s._L.psolve_hip_factorize(s._h, mat.shape[0], mat.nnz, mat.indptr.ctypes.data, mat.indices.ctypes.data,
                                           mat.data.ctypes.data))
        s._n = mat.shape[0]
        x = np.zeros(A.n)
        s.solve(b, x)
        info = s.get_info()
        sols.append(x)
        its.append(info["num_iterations"])
        shapes.append([s.amg_level_info(l)[:2] for l in range(info["amg_levels"])])
        assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1.5e-10
    assert shapes[0] == shapes[1] and abs(its[0] - its[1]) <= 1
    assert np.abs(sols[0] - sols[1]).max() <= 1e-8 * np.abs(sols[0]).max()


@pytest.mark.parametrize("grid,ce", [((24, 22, 20), 60), ((40, 40, 40), 200)])
def test_renumbered_levels_are_the_oracle_hierarchy_permuted(S, oracle, grid, ce):
    """"amg.renumber": after the setup, levels >= 1 are renumbered for locality (the nodes of one coarse aggregate
    become consecutive).  It must be the SAME hierarchy -- the oracle's aggregates and operators under one
    permutation per level: A_l = Pi_l A Pi_l^T and P_l = Pi_l P Pi_{l+1}^T entry for entry (bit-equal right after the
    setup: the numbers are computed before the rows move), the cycle's action within rounding of the oracle's (row
    sums add in another order: 1e-11 relative, against 1e-9 for the tests above), PCG's count within one, and a
    numeric refresh on the same pattern keeps all of that (its numbers are recomputed IN the new order: 1e-12)."""
    cfg = dict(coarse_enough=ce, ncycle=1, cheb_degree=3, cheb_power_iters=20)

    def random_values(seed):
        """the grid's pattern with random M-matrix values (no entry of a Galerkin product cancels exactly, so a second
        matrix of this kind keeps the strength graphs and the refresh is accepted)"""
        rng = np.random.default_rng(seed)
        U = sp.triu(oracle.poisson7(*grid).to_scipy(), k=1).tocoo()
        U = sp.coo_matrix((-rng.uniform(0.5, 2, U.nnz), (U.row, U.col)), shape=U.shape)
        M = (U + U.T + sp.diags(rng.uniform(12.5, 14, U.shape[0]))).tocsr()
        M.sort_indices()
        return oracle.CSR.from_scipy(M)

    A = random_values(4)
    ref = oracle.AMG(A, **cfg)
    assert ref.num_levels >= 3
    # (the smaller case sweeps its aggregates on the device, the larger one on the host with the level's smoother
    # queued underneath: that smoother is redone after the renumbering)
    agg_min = 0 if ce == 60 else 100000
    s = _solver(S, A.to_scipy(), dict(cfg, renumber=1, renumber_min_rows=0, aggregation_min_rows=agg_min))
    assert s.get_info()["amg_levels"] == ref.num_levels

    def check(ref, tol, plain=None):
        """`plain`: a solver with the same hierarchy in the setup's numbering (its matrices must match bit for bit);
        without it the oracle's matrices to `tol`"""
        perms = [s.amg_level_perm(l) for l in range(ref.num_levels)]
        assert not perms[0][1] and not perms[-1][1] and all(f for _, f in perms[1:-1])  # finest / coarsest keep theirs
        for l, (p, _) in enumerate(perms):
            assert np.array_equal(np.sort(p), np.arange(p.size))
        for l in range(ref.num_levels):
            for what, key in ((0, "A"), (1, "P"), (2, "R")):
                if what and l == ref.num_levels - 1:
                    continue
                shape, ptr, col, val = s.amg_level_matrix(l, what)
                M = sp.csr_matrix((val, col, ptr), shape=shape)
                assert M.has_sorted_indices
                if plain is not None:
                    shape0, ptr0, col0, val0 = plain.amg_level_matrix(l, what)
                    O_ = sp.csr_matrix((val0, col0, ptr0), shape=shape0).tocoo()
                else:
                    O_ = ref.level(l, key).to_scipy().tocoo()
                pr = perms[l + 1][0] if what == 2 else perms[l][0]
                pc = perms[l + 1][0] if what == 1 else perms[l][0]
                E = sp.csr_matrix((O_.data, (pr[O_.row], pc[O_.col])), shape=O_.shape)
                E.sort_indices()
                assert np.array_equal(M.indptr, E.indptr) and np.array_equal(M.indices, E.indices), (l, key)
                if plain is not None:
                    assert np.array_equal(M.data, E.data), (l, key)
                else:
                    assert np.abs(M.data - E.data).max() <= tol * np.abs(E.data).max(), (l, key)
            assert np.isclose(s.amg_level_info(l)[2], ref.level_scalars(l)["rho"], rtol=1e-9)
        # the new order of level l is (new id of the node's aggregate on level l + 1, old id), 1 <= l <= levels - 2
        for l in range(1, ref.num_levels - 1):
            _, agg = oracle.plain_aggregates(ref.level(l), 0.0)  # the sweep's aggregates, in the setup's numbering
            assert agg.min() >= 0
            order = np.argsort(perms[l][0])                     # old id of the node at every new position
            key = perms[l + 1][0][agg[order]]
            assert np.all(np.diff(key) >= 0)
            same = np.diff(key) == 0
            assert np.all(np.diff(order)[same] > 0)
        r = oracle.splitmix_vector(A.n, 17)
        z = s.device_array(A.n)
        s.precond_apply_device(s.to_device(r), z)
        zo = ref.apply(r)
        assert np.linalg.norm(z.download() - zo) <= 1e-11 * np.linalg.norm(zo)

    plain = _solver(S, A.to_scipy(), dict(cfg, renumber=0, aggregation_min_rows=agg_min))
    check(ref, 0, plain)
    del plain
    check(ref, 1e-12)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-10)
    assert abs(s.get_info()["num_iterations"] - ito) <= 1 and np.abs(x - xo).max() < 1e-8
    # same pattern, new values: the refresh works on the renumbered structures
    A2 = random_values(5)
    s.factorize(A2.to_scipy())
    assert s.get_param("amg.last_setup_reused") == 1
    ref2 = oracle.AMG(A2, **cfg)
    A = A2
    check(ref2, 1e-12)
    # and the default threshold leaves small levels alone
    t = _solver(S, oracle.poisson7(*grid).to_scipy(), dict(cfg))
    assert not any(t.amg_level_perm(l)[1] for l in range(ref.num_levels))


def test_amgcl_params_block_builds_the_references_configuration(S, oracle):
    """`"solver": "HIP"` with the caller's `/AMGCL` block kept (/HIP/amgcl_params): the reference's AMGCL configuration
    -- W-cycle, Chebyshev-16, 100 power iterations, tol 1e-10 (AMGCL.cpp:32-65) -- patched by the caller's objects; the
    iteration count is the oracle's amgcl::solver::cg with that hierarchy, and /HIP keys still win."""
    A = oracle.poisson7(14, 13, 12)
    M = A.to_scipy()
    s = S.create({"solver": "HIP", "HIP": {"amgcl_params": True, "amg": {"coarse_enough": 60}},
                  "AMGCL": {"precond": {"relax": {"degree": 5}}}})
    assert s.get_param("precond") == 2 and s.get_param("tolerance") == 1e-10 and s.get_param("max_iter") == 1000
    assert s.get_param("amg.ncycle") == 2 and s.get_param("amg.cheb_degree") == 5 and s.get_param("amg.cheb_power_iters") == 100
    assert s.get_param("amg.coarse_enough") == 60
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    ref = oracle.AMG(A, coarse_enough=60, cheb_degree=5)  # the oracle's defaults are AMGCL.cpp:32-65
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-10, max_iter=1000)
    assert abs(s.get_info()["num_iterations"] - ito) <= 1 and np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()


# ---- round 5 ----------------------------------------------------------------------------------------------------------
def _round5_case(oracle, case):
    """(matrix, block size, coarse_enough) of the hierarchy-equality cases: stencils, wide unstructured rows, a hub row far
    beyond every LDS slot, block value types"""
    bs, ce = 1, 40
    if case == "poisson":
        M = oracle.poisson7(22, 19, 20).to_scipy()
    elif case == "random_wide":
        M, ce = _random_graph_spd(6000, 40, 11), 10
    elif case == "arrow":
        M, ce = _arrow_spd(20000), 10
    elif case == "elasticity_block3":
        M, bs, ce = oracle.elasticity_q1(9).to_scipy(), 3, 60
    elif case == "random_block3":
        G = _random_graph_spd(1500, 20, 5)
        T = sp.csr_matrix(np.array([[1.0, 0.2, 0.0], [0.2, 1.0, 0.1], [0.0, 0.1, 1.0]]))
        M, bs, ce = sp.kron(G, T, format="csr"), 3, 10
    elif case == "gr3030_block2":
        M, bs, ce = oracle.gr_30_30().to_scipy(), 2, 50
    else:
        raise ValueError(case)
    M = sp.csr_matrix(M)
    M.sort_indices()
    return M, bs, ce


def _same_pattern_spd(M0, bs, rng):
    """D M0 D with a positive diagonal that is constant per node: SPD, the same pattern (explicit zeros kept)"""
    n = M0.shape[0]
    d = (1.0 + 0.3 * rng.uniform(0, 1, n // bs)).repeat(bs)
    Mk = M0.copy()
    rows = np.repeat(np.arange(n), np.diff(M0.indptr))
    Mk.data = M0.data * d[rows] * d[M0.indices]
    return Mk


def _assert_hierarchy_equals_host(s, host, tag):
    assert s.get_info()["amg_levels"] == host.num_levels
    for l in range(host.num_levels):
        for what, w in (("A", 0), ("P", 1), ("R", 2)):
            h = host.level(l, what)
            if h is None:
                continue
            shape, ptr, col, val = s.amg_level_matrix(l, w)
            assert shape == (h[0], h[1]), (tag, l, what)
            assert np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]), (tag, l, what)
            assert np.array_equal(val, h[4]), (tag, l, what)


@pytest.mark.parametrize("case", ["poisson", "random_wide", "arrow", "elasticity_block3", "random_block3", "gr3030_block2"])
def test_numeric_refresh_equals_host_hierarchy(S, oracle, case):
    """The numeric refresh of a kept hierarchy (Newton.cpp:189-193 refactorizes a matrix of constant pattern every iteration):
    A P and R (A P) recomputed by the row-wise kernels sum every entry's terms in the order of the sequential host product, so
    the refreshed hierarchy equals the host construction on the new values BIT FOR BIT -- narrow rows, wide rows, 3 x 3 blocks
    (products on the block patterns), 2 x 2 blocks (scalar products on the expanded patterns).  (Round 5's kept product plans
    took this test over from round 4's; they measured slower and left the library in round 6.)"""
    from polysolve_amd import HostHierarchy
    M0, bs, ce = _round5_case(oracle, case)
    # (generic values from the start: the Galerkin operators of the uniform 7-point grid hold entries that cancel to exactly
    # zero, which a scaled matrix turns into nonzeros -- the strength graph of level 1 changes and the refresh is, rightly,
    # refused)
    M0 = _same_pattern_spd(M0, bs, np.random.default_rng(1))
    n = M0.shape[0]
    amg = dict(coarse_enough=ce, max_levels=5, aggregation_min_rows=0, ncycle=1, cheb_degree=2, cheb_power_iters=5)
    s = _solver(S, M0, amg, block_size=bs)
    assert s.get_info()["amg_levels"] >= 2
    rng = np.random.default_rng(5)
    for k in range(3):
        Mk = _same_pattern_spd(M0, bs, rng)
        s.factorize(Mk)
        assert s.get_param("amg.last_setup_reused") == 1
        host = HostHierarchy(n, Mk.indptr, Mk.indices, Mk.data, max_levels=5, coarse_enough=ce, block_size=bs)
        _assert_hierarchy_equals_host(s, host, (case, k, "refresh"))
        b = rng.uniform(-1, 1, n)
        x = np.zeros(n)
        s.solve(b, x)
        assert np.linalg.norm(Mk @ x - b) / np.linalg.norm(b) < 1e-8
    # a new pattern drops the kept hierarchy
    other = sp.csr_matrix(oracle.poisson7(9).to_scipy()) if bs == 1 else sp.csr_matrix(oracle.elasticity_q1(5).to_scipy())
    if bs == 2:
        other = sp.csr_matrix(oracle.poisson7(8).to_scipy())
    other.sort_indices()
    s.factorize(other)
    assert s.get_param("amg.last_setup_reused") == 0


@pytest.mark.parametrize("agg", ["parallel", "compact"])
@pytest.mark.parametrize("case", ["poisson", "random_wide", "arrow", "elasticity_block3", "random_block3", "gr3030_block2"])
def test_parallel_aggregation_on_the_device_equals_host_and_oracle(S, oracle, case, agg):
    """amg.aggregation = "parallel" (opt-in; the default stays AMGCL's sweep, AMGCL.cpp:32-65): the seeds are the distance-2
    maximal independent set by hashed priorities, found in synchronous rounds on the device (amg_aggregate.hip: mis_*).  Integer
    work: the device hierarchy equals the host construction with the same option bit for bit (and that one the oracle's,
    tests/test_amg_host.py), on every level -- small levels included, which the default mode would hand to the host sweep.
    Round 6, "compact": one-hop aggregates around two generations of such sets (compact_* kernels), the same guarantee."""
    from polysolve_amd import HostHierarchy
    M, bs, ce = _round5_case(oracle, case)
    M = _same_pattern_spd(M, bs, np.random.default_rng(1))  # (generic values: no exact cancellations in the Galerkin operators)
    n = M.shape[0]
    amg = dict(coarse_enough=ce, max_levels=5, cheb_power_iters=5, aggregation=agg)
    host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=5, coarse_enough=ce, block_size=bs, aggregation=agg)
    s = _solver(S, M, amg, block_size=bs)
    assert s.get_param("amg.aggregation") == {"parallel": 1, "compact": 2}[agg]
    assert s.get_param("amg.levels_aggregated_on_device") == host.num_levels - 1
    _assert_hierarchy_equals_host(s, host, case)
    # not the sweep's hierarchy
    h0 = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=5, coarse_enough=ce, block_size=bs)
    assert h0.level(1, "A")[0] != host.level(1, "A")[0] or case == "arrow"
    # the all-host construction and a numeric refresh go through the same aggregates
    s0 = _solver(S, M, dict(amg, device_setup=0), block_size=bs)
    b = np.ones(n)
    x, x0 = np.zeros(n), np.zeros(n)
    s.solve(b, x)
    s0.solve(b, x0)
    assert s.get_info()["num_iterations"] == s0.get_info()["num_iterations"] and np.array_equal(x, x0)
    Mk = _same_pattern_spd(M, bs, np.random.default_rng(3))
    s.factorize(Mk)
    assert s.get_param("amg.last_setup_reused") == 1
    hk = HostHierarchy(n, Mk.indptr, Mk.indices, Mk.data, max_levels=5, coarse_enough=ce, block_size=bs, aggregation=agg)
    _assert_hierarchy_equals_host(s, hk, (case, "refresh"))
    # switching the option is a new hierarchy, not a refresh
    s.set_parameters({"HIP": {"amg": {"aggregation": "amgcl"}}})
    s.factorize(Mk)
    assert s.get_param("amg.last_setup_reused") == 0


@pytest.mark.parametrize("name,bs", [("poisson", 1), ("elasticity", 3), ("tets", 1)])
def test_parallel_aggregation_pcg_matches_oracle_and_stays_close_to_the_default(S, oracle, golden_dir, name, bs):
    """PCG under the parallel aggregation: the oracle's count +- 1 (the oracle restates the same algorithm), and within 10 %
    (+ 1) of the default aggregation's count on the same system -- Poisson, Q1 elasticity on 3 x 3 blocks, the unstructured
    tetrahedral fixture."""
    cfg = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3)
    if name == "poisson":
        A, ce = oracle.poisson7(40), 300
    elif name == "elasticity":
        A, ce = oracle.elasticity_q1(16), 300
    else:
        d = np.load(os.path.join(golden_dir, "reorder_tets.npz"))
        A, ce = oracle.CSR(int(d["n"]), d["rowptr"].astype(np.int32), d["col"].astype(np.int32), d["val"].astype(np.float64)), 60
    M = sp.csr_matrix(A.to_scipy())
    M.sort_indices()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    its = {}
    for agg in ("amgcl", "parallel", "compact"):
        ref = oracle.AMG(A, coarse_enough=ce, block_size=bs, aggregation=agg, **cfg)
        xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-8, max_iter=500)
        s = _solver(S, M, dict(cfg, coarse_enough=ce, aggregation=agg), tol=1e-8, block_size=bs, extra=dict(reorder=0))
        assert s.get_info()["amg_levels"] == ref.num_levels
        for l in range(ref.num_levels):
            assert s.amg_level_info(l)[:2] == (ref.level(l).n, ref.level(l).nnz)
        x = np.zeros(A.n)
        s.solve(b, x)
        assert abs(s.get_info()["num_iterations"] - ito) <= 1, (agg, s.get_info()["num_iterations"], ito)
        assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
        its[agg] = s.get_info()["num_iterations"]
    assert its["parallel"] <= 1.1 * its["amgcl"] + 1, its
    assert its["compact"] <= 1.15 * its["amgcl"] + 2, its


@pytest.mark.parametrize("bs", [1, 3])
@pytest.mark.parametrize("cfg", [dict(relax_type="damped_jacobi"), dict(relax_type="damped_jacobi", damping=0.55, npre=2, npost=2),
                                 dict(relax_type="spai0"), dict(coarsening="aggregation"),
                                 dict(coarsening="aggregation", over_interp=1.2, relax_type="spai0"),
                                 dict(direct_coarse=1), dict(direct_coarse=1, coarsening="aggregation", relax_type="damped_jacobi"),
                                 dict(cheb_scale=0, cheb_power_iters=30), dict(cheb_power_iters=0, cheb_higher=1.0)],
                         ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()) if isinstance(c, dict) else str(c))
def test_amgcl_runtime_classes_match_oracle(S, oracle, cfg, bs):
    """The classes amgcl's runtime wrappers build when the reference forwards its free strings (linear-solver-spec.json:393-397
    relax `type`, :423-427 coarsening `type`, /AMGCL/precond/direct_coarse; AMGCL.cpp:67-92, 178-181), restated in
    oracle/amg_oracle.c from amgcl/relaxation/{damped_jacobi,spai0}.hpp, amgcl/coarsening/aggregation.hpp and amgcl/amg.hpp:
    same level sizes, the cycle's action to 1e-9, PCG counts +- 1 -- first setup AND numeric refresh, scalar and 3 x 3 blocks."""
    A = oracle.poisson7(14, 12, 13) if bs == 1 else oracle.elasticity_q1(9)
    M0 = sp.csr_matrix(A.to_scipy())
    M0.sort_indices()
    M0 = _same_pattern_spd(M0, bs, np.random.default_rng(2))
    base = dict(coarse_enough=60 if bs == 1 else 100, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    full = dict(base, **cfg)
    s = _solver(S, M0, dict(full, aggregation_min_rows=0), tol=1e-9, block_size=bs)
    rng = np.random.default_rng(9)
    for k, Mk in enumerate((M0, _same_pattern_spd(M0, bs, rng))):
        if k:
            s.factorize(Mk)
            assert s.get_param("amg.last_setup_reused") == 1
        Ak = oracle.CSR.from_scipy(Mk)
        ref = oracle.AMG(Ak, block_size=bs, **full)
        assert s.get_info()["amg_levels"] == ref.num_levels and ref.num_levels >= 2
        for l in range(ref.num_levels):
            assert s.amg_level_info(l)[:2] == (ref.level(l).n, ref.level(l).nnz), (k, l)
        if cfg.get("coarsening") == "aggregation":
            shape, ptr, col, val = s.amg_level_matrix(0, 1)
            assert set(np.unique(val)) <= {0.0, 1.0}  # the tentative prolongation
            Ac, Ao = s.amg_level_matrix(1, 0), ref.level(1).to_scipy()
            Ad = sp.csr_matrix((Ac[3], Ac[2], Ac[1]), shape=Ac[0])
            assert abs(Ad - Ao).max() <= 1e-13 * abs(Ao).max()
        r = oracle.splitmix_vector(Ak.n, 11 + k)
        z = s.device_array(Ak.n)
        s.precond_apply_device(s.to_device(r), z)
        zo = ref.apply(r)
        assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo), (k, cfg)
        b = oracle.spmv(Ak, oracle.splitmix_vector(Ak.n, 42))
        x = np.zeros(Ak.n)
        s.solve(b, x)
        xo, ito, _ = oracle.cg_amgcl(Ak, b, precond=ref, tol=1e-9, max_iter=1000)
        # (+- 1, or 5 % where a weak smoother needs many dozens of iterations: rounding differences of the fused kernels then
        # move the iteration at which the recurrence residual crosses the threshold by a few)
        assert abs(s.get_info()["num_iterations"] - ito) <= max(1, 0.05 * ito), (k, cfg, s.get_info()["num_iterations"], ito)
        assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


@pytest.mark.parametrize("bs", [1, 3])
@pytest.mark.parametrize("cfg", [dict(relax_type="gauss_seidel"), dict(relax_type="gauss_seidel", npre=2, npost=2, ncycle=2),
                                 dict(relax_type="ilu0"), dict(relax_type="ilu0", ilu_damping=0.8, npre=2),
                                 dict(relax_type="gauss_seidel", direct_coarse=1),
                                 dict(relax_type="gauss_seidel", precond_class="relaxation"),
                                 dict(relax_type="ilu0", precond_class="relaxation"),
                                 dict(relax_type="chebyshev", precond_class="relaxation"),
                                 dict(relax_type="spai0", precond_class="relaxation")],
                         ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()) if isinstance(c, dict) else str(c))
def test_ordered_relaxations_match_oracle(S, oracle, cfg, bs):
    """Round 6: amgcl's ORDERED relaxations and its single-level preconditioner class (linear-solver-spec.json:393-397 relax
    `type`, /AMGCL/precond/class; AMGCL.cpp:67-92) -- gauss_seidel (forward sweep before, backward sweep after the coarse
    correction), ilu0 (IKJ factorization on A's pattern, x += damping (LU)^-1 (rhs - A x)) and class = relaxation
    (amgcl::relaxation::as_preconditioner), restated in oracle/amg_oracle.c (gs_sweep / ilu0_factor / ilu0_solve) from
    amgcl/relaxation/{gauss_seidel,ilu0}.hpp and detail/ilu_solve.hpp.  On the device a sweep is ONE launch in which every row
    waits for the rows it depends on (amg_sweep.hip): the same operations in the same order as the serial loops, so a
    single-level gauss_seidel / ilu0 application is BIT-equal to the oracle's; cycles to 1e-9, PCG counts +- 1; first setup and
    numeric refresh, scalar and 3 x 3 blocks."""
    A = oracle.poisson7(14, 12, 13) if bs == 1 else oracle.elasticity_q1(9)
    M0 = sp.csr_matrix(A.to_scipy())
    M0.sort_indices()
    M0 = _same_pattern_spd(M0, bs, np.random.default_rng(2))
    base = dict(coarse_enough=60 if bs == 1 else 100, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    full = dict(base, **cfg)
    dev = dict(full, aggregation_min_rows=0)
    if "precond_class" in dev:
        dev["class"] = dev.pop("precond_class")
    s = _solver(S, M0, dev, tol=1e-9, block_size=bs, extra=dict(reorder=0))  # (a sweep's order is the numbering's)
    single = cfg.get("precond_class") == "relaxation"
    rng = np.random.default_rng(9)
    for k, Mk in enumerate((M0, _same_pattern_spd(M0, bs, rng))):
        if k:
            s.factorize(Mk)
        Ak = oracle.CSR.from_scipy(Mk)
        ref = oracle.AMG(Ak, block_size=bs, **full)
        assert s.get_info()["amg_levels"] == ref.num_levels and (ref.num_levels == 1 if single else ref.num_levels >= 2)
        r = oracle.splitmix_vector(Ak.n, 11 + k)
        z = s.device_array(Ak.n)
        s.precond_apply_device(s.to_device(r), z)
        zo = ref.apply(r)
        if single and cfg["relax_type"] in ("gauss_seidel", "ilu0"):
            assert np.array_equal(z.download(), zo), (k, cfg, np.abs(z.download() - zo).max())
        else:
            assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo), (k, cfg)
        b = oracle.spmv(Ak, oracle.splitmix_vector(Ak.n, 42))
        x = np.zeros(Ak.n)
        s.solve(b, x)
        xo, ito, _ = oracle.cg_amgcl(Ak, b, precond=ref, tol=1e-9, max_iter=1000)
        assert abs(s.get_info()["num_iterations"] - ito) <= max(1, 0.05 * ito), (k, cfg, s.get_info()["num_iterations"], ito)
        assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
        assert s.get_info()["true_residual"] <= 2e-9


def test_a_sweep_does_not_serialise_on_lines_that_are_no_multiple_of_a_ticket(S):
    """Round 6: a 64-row ticket of a 100^3 grid holds the end of one line and the start of the next.  While only a window of
    lanes behind a wave's first unfinished lane polled other waves' rows, the start of the next line waited behind the end of
    this one and the tickets ran one after the other: 1.9 s per sweep instead of 2.5 ms.  A generous bound (40 x the measured
    time) that only such a serialisation breaks."""
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "max_iter": 20, "amg": {"relax_type": "gauss_seidel", "class": "relaxation"}}})
    s.generate_poisson7(100)
    s.synchronize()
    ops = s.amg_time_level_ops(0, 2)
    assert ops["cheb_first_us"] < 100e3, ops


def test_ordered_relaxations_on_a_scattered_numbering_and_without_a_diagonal(S, oracle):
    """The sweeps of gauss_seidel / ilu0 follow the numbering they are given, and the sweep order IS the smoother: the automatic
    renumbering at factorize (`reorder` 2) leaves such a system alone, like one preconditioned by ic or schwarz, so a scattered
    numbering gets amgcl's sweeps in the caller's order -- the oracle's iteration counts; renumbered on request (`reorder` 1) it
    still converges, with another Gauss-Seidel.  A row without its diagonal is refused by ilu0 (amgcl: "No diagonal value in
    system matrix") instead of hanging the rows that wait for it."""
    A = oracle.poisson7(20, 18, 16)
    M = sp.csr_matrix(A.to_scipy())
    perm = np.random.default_rng(5).permutation(A.n)
    Mp = sp.csr_matrix(M[perm][:, perm])
    Mp.sort_indices()
    b = Mp @ oracle.splitmix_vector(A.n, 3)
    Ap = oracle.CSR.from_scipy(Mp)
    for rt in ("gauss_seidel", "ilu0"):
        cfg = dict(relax_type=rt, coarse_enough=200, ncycle=1)
        s = _solver(S, Mp, dict(cfg, aggregation_min_rows=0), tol=1e-9, extra=dict(reorder_min_rows=0))
        assert s.get_param("reorder.active") == 0
        x = np.zeros(A.n)
        s.solve(b, x)
        info = s.get_info()
        ref = oracle.AMG(Ap, **cfg)
        _, ito, _ = oracle.cg_amgcl(Ap, b, precond=ref, tol=1e-9, max_iter=500)
        assert info["true_residual"] <= 2e-9 and abs(info["num_iterations"] - ito) <= 1, (rt, info, ito)
        s = _solver(S, Mp, cfg, tol=1e-9, extra=dict(reorder=1))
        assert s.get_param("reorder.active") == 1
        x = np.zeros(A.n)
        s.solve(b, x)
        info = s.get_info()
        assert info["true_residual"] <= 2e-9 and info["num_iterations"] < 80, (rt, info)
    # ilu0 without a diagonal entry in one row
    C = M.tolil()
    C[7, 7] = 0.0
    C = sp.csr_matrix(C)
    C.eliminate_zeros()
    C.sort_indices()
    with pytest.raises(RuntimeError):
        _solver(S, C, {"relax_type": "ilu0", "class": "relaxation"}, tol=1e-9, extra=dict(reorder=0))


@pytest.mark.parametrize("case", ["poisson_1900", "elasticity_block3_1500", "ragged_last_block"])
def test_direct_coarse_blocked_inverse_matches_oracle(S, oracle, case):
    """Round 6: a coarsest level of more than 128 rows is inverted in BLOCKS of 32 columns (gj_pivot / gj_panels / gj_update,
    amg_relax.hip: n / 32 x 3 launches instead of n that each stream the matrix) -- the cycle with the directly solved coarsest
    level acts like the oracle's (amgcl's skyline LU restated) to 1e-9, PCG counts +- 1; a size that is no multiple of 32 and
    one just above the threshold included."""
    if case == "poisson_1900":
        A, bs, ce = oracle.poisson7(26, 25, 24), 1, 3000
    elif case == "elasticity_block3_1500":
        A, bs, ce = oracle.elasticity_q1(24), 3, 3000
    else:
        A, bs, ce = oracle.poisson7(11, 10, 13), 1, 300
    M = sp.csr_matrix(A.to_scipy())
    M.sort_indices()
    cfg = dict(coarse_enough=ce, max_levels=2, ncycle=1, cheb_degree=2, cheb_power_iters=20, direct_coarse=1)
    ref = oracle.AMG(A, block_size=bs, **cfg)
    nc = ref.level(1).n
    assert ref.num_levels == 2 and nc > 128 and (case != "ragged_last_block" or nc % 32)
    s = _solver(S, M, cfg, tol=1e-9, block_size=bs, extra=dict(reorder=0))
    assert s.get_info()["amg_levels"] == 2 and s.amg_level_info(1)[0] == nc
    r = oracle.splitmix_vector(A.n, 5)
    z = s.device_array(A.n)
    s.precond_apply_device(s.to_device(r), z)
    zo = ref.apply(r)
    assert np.linalg.norm(z.download() - zo) <= 1e-9 * np.linalg.norm(zo), (case, nc)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    x = np.zeros(A.n)
    s.solve(b, x)
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-9, max_iter=500)
    assert abs(s.get_info()["num_iterations"] - ito) <= 1 and np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


def test_direct_coarse_limits_and_errors(S, oracle):
    """direct_coarse inverts the coarsest operator densely: a coarsest level beyond kDirectCoarseMaxRows (4096) rows is refused
    with a message that names the remedy, a hierarchy of one level (the matrix itself under coarse_enough) is solved exactly."""
    A = oracle.poisson7(18)
    M = sp.csr_matrix(A.to_scipy())
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "amg": {"direct_coarse": True, "max_levels": 1}}})
    with pytest.raises(RuntimeError, match="direct_coarse"):
        s.factorize(M)  # 5832 rows on the only level
    A = oracle.poisson7(12)
    M = sp.csr_matrix(A.to_scipy())
    s = _solver(S, M, dict(direct_coarse=True, coarse_enough=3000), tol=1e-10)
    assert s.get_info()["amg_levels"] == 1
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 1))
    x = np.zeros(A.n)
    s.solve(b, x)
    assert s.get_info()["num_iterations"] <= 2 and np.linalg.norm(M @ x - b) <= 1e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("bs,warm_iters", [(1, 8), (3, 8), (3, 4)])
def test_newton_sequence_with_warm_started_refresh(S, oracle, bs, warm_iters):
    """The recommended refresh mode for Newton loops (round 6; Newton.cpp:189-193 refactorizes every iteration): ten successive
    Hessians of one pattern, each 5 % away from the last (a cumulative drift of 60 %), in the recommended cycle (Chebyshev on
    [0.1, 1.1] x the estimated radius: the tightest interval this backend recommends, i.e. the one a poor radius hurts most).
    A handle that continues its power iterations from the previous factorize's vector for `warm_iters` steps
    ("amg.refresh_power_iters") takes the iteration count of a handle that estimates from scratch (20 steps from amgcl's random
    vector) +- 1 at EVERY step, to the same residual.  Its radii are never smaller than the cold estimate's by more than 3 % and
    exceed it by at most 8 %: the power iteration approaches the radius from below, and the continued one has run longer -- the
    20-step cold estimate of the 7-point operator is 1.91 of a true 2, the warm one 1.97."""
    A = oracle.poisson7(30, 28, 26) if bs == 1 else oracle.elasticity_q1(14)
    M0 = sp.csr_matrix(A.to_scipy())
    M0.sort_indices()
    M0 = _same_pattern_spd(M0, bs, np.random.default_rng(4))
    amg = dict(coarse_enough=200, ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3)
    cold = _solver(S, M0, amg, tol=1e-8, block_size=bs)
    warm = _solver(S, M0, dict(amg, refresh_power_iters=warm_iters), tol=1e-8, block_size=bs)
    n = M0.shape[0]
    rng = np.random.default_rng(11)
    Mk = M0
    for k in range(10):
        d = (1.0 + 0.05 * rng.uniform(0, 1, n // bs)).repeat(bs)
        rows = np.repeat(np.arange(n), np.diff(Mk.indptr))
        Mn = Mk.copy()
        Mn.data = Mk.data * d[rows] * d[Mk.indices]
        Mk = Mn
        b = rng.uniform(-1, 1, n)
        its = []
        for s in (cold, warm):
            s.factorize(Mk)
            assert s.get_param("amg.last_setup_reused") == 1
            x = np.zeros(n)
            s.solve(b, x)
            assert np.linalg.norm(Mk @ x - b) <= 1.5e-8 * np.linalg.norm(b)
            its.append(s.get_info()["num_iterations"])
        assert abs(its[1] - its[0]) <= 1, (k, its)
        for l in range(cold.get_info()["amg_levels"]):
            rc, rw = cold.amg_level_info(l)[2], warm.amg_level_info(l)[2]
            assert 0.97 * rc <= rw <= 1.08 * rc, (k, l, rc, rw)


@pytest.mark.parametrize("bs", [1, 3])
def test_refresh_power_iterations_warm_start(S, oracle, bs):
    """amg.refresh_power_iters (opt-in, NOT amgcl's estimate): a factorize of the same pattern continues the smoothers' power
    iterations from the vector the previous factorize ended with instead of starting cheb_power_iters steps from the random
    vector -- Newton's next Hessian is close to the last one (Newton.cpp:189-193).  The radii stay within a few per cent of the
    cold estimate of the NEW matrix, PCG takes the same count +- 1, the hierarchy's operators are the refresh's (bit-equal);
    the default (-1) is untouched by the option's code path."""
    A = oracle.poisson7(16, 14, 15) if bs == 1 else oracle.elasticity_q1(9)
    M0 = sp.csr_matrix(A.to_scipy())
    M0.sort_indices()
    M0 = _same_pattern_spd(M0, bs, np.random.default_rng(4))
    amg = dict(coarse_enough=60 if bs == 1 else 100, ncycle=1, cheb_degree=2, cheb_power_iters=20, cheb_higher=1.2)
    cold = _solver(S, M0, amg, tol=1e-9, block_size=bs)
    warm = _solver(S, M0, dict(amg, refresh_power_iters=4), tol=1e-9, block_size=bs)
    keep = _solver(S, M0, dict(amg, refresh_power_iters=0), tol=1e-9, block_size=bs)
    levels = cold.get_info()["amg_levels"]
    for l in range(levels):  # the first factorize is the cold estimate in every mode
        assert warm.amg_level_info(l) == cold.amg_level_info(l) == keep.amg_level_info(l)
    rng = np.random.default_rng(8)
    n = M0.shape[0]
    Mk = M0
    for k in range(3):
        d = (1.0 + 0.05 * rng.uniform(0, 1, n // bs)).repeat(bs)  # a small change, as between Newton iterations
        rows = np.repeat(np.arange(n), np.diff(Mk.indptr))
        Mn = Mk.copy()
        Mn.data = Mk.data * d[rows] * d[Mk.indices]
        Mk = Mn
        for s in (cold, warm, keep):
            s.factorize(Mk)
            assert s.get_param("amg.last_setup_reused") == 1
        b = rng.uniform(-1, 1, n)
        its = []
        for s in (cold, warm, keep):
            x = np.zeros(n)
            s.solve(b, x)
            assert np.linalg.norm(Mk @ x - b) <= 1e-8 * np.linalg.norm(b)
            its.append(s.get_info()["num_iterations"])
        assert abs(its[1] - its[0]) <= 1 and abs(its[2] - its[0]) <= 2, its
        for l in range(levels):
            rc, rw = cold.amg_level_info(l)[2], warm.amg_level_info(l)[2]
            assert cold.amg_level_info(l)[:2] == warm.amg_level_info(l)[:2]
            assert abs(rw - rc) <= 0.05 * rc, (k, l, rc, rw)
            for what in (0, 1):
                if l + 1 == levels and what == 1:
                    continue
                a, bm = cold.amg_level_matrix(l, what), warm.amg_level_matrix(l, what)
                assert np.array_equal(a[3], bm[3])  # the operators do not depend on the smoothers
