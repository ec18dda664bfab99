"""GPU parity tests of the drop-in contract: the reference's own linear-solver tests
(/root/reference/tests/test_linear_solver.cpp), restated for Solver::create("HIP"), plus parity with
the CPU oracle (iteration counts and solutions) and with the committed golden fixtures.

Tolerances (fp64): the reference asserts ||Ax-b|| < 1e-8 with solver tolerance 1e-10 (:128,160-162)
and ||Ax-b||/||b|| < 1e-7 for the AMGCL legs (:600-601); against the oracle we require the same
iteration count +-1 (dot products are tree-reduced on the GPU, chunk-reduced in the oracle) and
|x_gpu - x_oracle| <= 1e-6 * |x|_inf on systems with cond <= 1e5."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


def _pre_factor_matrix(S0, rng):
    U = sp.triu(S0, k=1).tocoo()
    off = -rng.uniform(0.1, 5, U.nnz)
    U = sp.coo_matrix((off, (U.row, U.col)), shape=S0.shape)
    return (U + U.T + sp.diags(rng.uniform(0.1, 5, S0.shape[0]) * 100)).tocsc()


def test_all(S, oracle):
    """TEST_CASE("all") :103-164: name(), tolerance 1e-10, random b, x0 = 0, ||Ax-b|| < 1e-8."""
    A = oracle.elasticity_q1(7).to_scipy().tocsc()
    for name in S.available_solvers():
        solver = S.create(name, "")
        solver.set_parameters({name: {"tolerance": 1e-10}})
        rng = np.random.default_rng(0)
        b = rng.uniform(-1, 1, A.shape[0])
        x = np.zeros(A.shape[0])
        assert not solver.is_dense()
        solver.analyze_pattern(A, A.shape[0])
        solver.factorize(A)
        solver.solve(b, x)
        assert solver.name() == name
        info = solver.get_info()
        assert np.linalg.norm(A @ x - b) < 1e-8
        assert info["solver_iter"] > 0 and info["num_iterations"] == info["solver_iter"] + 1
        assert info["solver_status"] == "Reach relative tolerance"
        assert info["solver_error"] == info["final_res_norm"] < 1e-10


def test_jse_json_factory(S, oracle):
    """TEST_CASE("jse") / ("multi-solver") :52-101: create from json, priority list falls through."""
    A = oracle.poisson7(10).to_scipy().tocsc()
    b = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    for params in ({}, {"solver": ["Hypre", "HIP"]}, {"solver": "HIP", "precond": "Eigen::IdentityPreconditioner",
                                                    "HIP": {"tolerance": 1e-11, "max_iter": 500}}):
        solver = S.create(dict(params))
        if "HIP" not in params:
            solver.set_parameters({"HIP": {"tolerance": 1e-10}})
        x = np.zeros(A.shape[0])
        solver.analyze_pattern(A, A.shape[0])
        solver.factorize(A)
        solver.solve(b, x)
        assert np.linalg.norm(A @ x - b) < 1e-8
    with pytest.raises(RuntimeError):
        S.create({"solver": ["Hypre", "Pardiso"]})


def test_pre_factor(S, oracle):
    """TEST_CASE("pre_factor") :241-307: one analyze_pattern, 10 x (factorize new values + solve)."""
    S0 = oracle.poisson7(9, 8, 7).to_scipy()
    solver = S.create("HIP", "")
    solver.set_parameters({"HIP": {"tolerance": 1e-10}})
    solver.analyze_pattern(S0.tocsc(), S0.shape[0])
    rng = np.random.default_rng(42)
    for _ in range(10):
        At = _pre_factor_matrix(S0, rng)
        b = rng.uniform(-1, 1, At.shape[0])
        x = np.zeros(At.shape[0])
        solver.factorize(At)
        solver.solve(b, x)
        assert np.linalg.norm(At @ x - b) < 1e-8


def test_initial_guess_is_honoured(S, oracle):
    """TEST_CASE("amgcl_initial_guess") :400-455: a second solver started from the converged x
    reports num_iterations == 0 and leaves x alone (MAS's x := 0 work-around is NOT copied)."""
    A = oracle.poisson7(12).to_scipy().tocsc()
    b = np.random.default_rng(2).uniform(-1, 1, A.shape[0])
    x = np.zeros(A.shape[0])
    s1 = S.create("HIP", "")
    s1.set_parameters({"HIP": {"tolerance": 1e-10}})
    s1.analyze_pattern(A, A.shape[0])
    s1.factorize(A)
    s1.solve(b, x)
    assert s1.get_info()["num_iterations"] > 0
    x_first = x.copy()
    s2 = S.create("HIP", "")
    s2.set_parameters({"HIP": {"tolerance": 2e-10}})
    s2.analyze_pattern(A, A.shape[0])
    s2.factorize(A)
    s2.solve(b, x)
    assert s2.get_info()["num_iterations"] == 0
    assert np.array_equal(x, x_first)
    assert np.linalg.norm(A @ x - b) < 1e-8
    # and a poor guess converges to the same solution
    x2 = np.random.default_rng(3).uniform(-5, 5, A.shape[0])
    s2.solve(b, x2)
    assert np.linalg.norm(x2 - x) / np.linalg.norm(x) < 1e-8


def test_gr_30_30_b_ones(S, oracle):
    """:541-602 scalar leg: gr_30_30, b = 1, ||Ax-b||/||b|| < 1e-7, iterations > 0."""
    G = oracle.gr_30_30().to_scipy().tocsc()
    b = np.ones(G.shape[0])
    x = np.zeros(G.shape[0])
    s = S.create("HIP", "")
    s.analyze_pattern(G, G.shape[0])
    s.factorize(G)
    s.solve(b, x)
    assert s.get_info()["num_iterations"] > 0
    assert np.linalg.norm(G @ x - b) / np.linalg.norm(b) < 1e-7


@pytest.mark.parametrize("precond,oname", [("", "jacobi"), ("Eigen::IdentityPreconditioner", "none")])
@pytest.mark.parametrize("name", ["poisson7_n4", "poisson7_n8", "poisson7_n12", "poisson7_6x5x7", "gr_30_30",
                                  "elasticity_q1_m5"])
def test_golden_parity(S, oracle, golden_dir, name, precond, oname):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n = int(g["n"])
    M = sp.csr_matrix((g["val"], g["col"], g["rowptr"]), shape=(n, n)).tocsc()
    s = S.create("HIP", precond)
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000}})
    s.analyze_pattern(M, n)
    s.factorize(M)
    x = np.zeros(n)
    s.solve(g["b"], x)
    info = s.get_info()
    want = int(g["cg_jacobi_iters"] if oname == "jacobi" else g["cg_none_iters"])
    assert abs(info["solver_iter"] - want) <= 1
    xe = g["x_exact"]
    assert np.linalg.norm(x - xe) / np.linalg.norm(xe) < 1e-5
    if oname == "jacobi":
        assert np.abs(x - g["cg_jacobi_x"]).max() <= 1e-6 * np.abs(xe).max()
        assert np.isclose(info["solver_error"], float(g["cg_jacobi_err"]), rtol=1e-4) or info["solver_iter"] != want
    assert info["true_residual"] < 2e-8


@pytest.mark.parametrize("grid", [(20, 20, 20), (48, 48, 48), (64, 32, 16), (100, 3, 3)])
def test_oracle_parity_poisson(S, oracle, grid):
    A = oracle.poisson7(*grid)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    xo, ito, erro = oracle.cg_eigen(A, b, tol=1e-8, max_iter=5000)
    s = S.create("HIP", "")
    M = A.to_scipy()
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert abs(info["solver_iter"] - ito) <= 1
    if info["solver_iter"] == ito:
        assert np.isclose(info["solver_error"], erro, rtol=1e-5)
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    assert info["true_residual"] < 1.5e-8


def test_oracle_parity_elasticity(S, oracle):
    A = oracle.elasticity_q1(12)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    xo, ito, erro = oracle.cg_eigen(A, b, tol=1e-8, max_iter=5000)
    s = S.create("HIP", "")
    M = A.to_scipy()
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert abs(info["solver_iter"] - ito) <= max(2, ito // 100)
    assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1.5e-8
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-5


def test_zero_rhs_and_max_iter_and_abs_tol(S, oracle):
    A = oracle.poisson7(10)
    M = A.to_scipy()
    s = S.create("HIP", "")
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    x = np.ones(A.n)
    s.solve(np.zeros(A.n), x)  # Eigen: rhsNorm2 == 0 -> x = 0, 0 iterations, error 0
    info = s.get_info()
    assert not x.any() and info["solver_iter"] == 0 and info["solver_error"] == 0
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 1))
    s.set_parameters({"HIP": {"tolerance": 1e-14, "max_iter": 5}})
    x = np.zeros(A.n)
    s.solve(b, x)  # non-convergence is not an error (Eigen/AMGCL return; caller inspects get_info)
    info = s.get_info()
    assert info["solver_iter"] == 5 and info["solver_status"] == "Reach max iterations"
    xo, ito, erro = oracle.cg_eigen(A, b, tol=1e-14, max_iter=5)
    assert ito == 5 and np.isclose(info["solver_error"], erro, rtol=1e-9)
    assert np.allclose(x, xo, rtol=0, atol=1e-12)
    s.set_parameters({"HIP": {"tolerance": 0.0, "absolute_tolerance": 1e-5, "max_iter": 1000}})
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert info["solver_status"] == "Reach absolute tolerance"
    assert np.linalg.norm(M @ x - b) < 1.1e-5  # Newton's absolute acceptance test (Newton.cpp:156,207)


def test_error_behaviour(S, oracle):
    A = oracle.poisson7(6)
    M = A.to_scipy().tocsc()
    s = S.create("HIP", "")
    with pytest.raises(RuntimeError, match="Size mismatch|factorize"):  # MASSolver.cu:380-383
        s.solve(np.ones(A.n), np.zeros(A.n))
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    with pytest.raises(RuntimeError, match="Size mismatch"):
        s.solve(np.ones(A.n + 1), np.zeros(A.n + 1))
    bad = M.copy()
    bad.data = bad.data.copy()
    bad.data[bad.indptr[3]: bad.indptr[4]][bad.indices[bad.indptr[3]: bad.indptr[4]] == 3] = np.nan
    with pytest.raises(RuntimeError, match="non-finite"):  # -> std::runtime_error, caught by Newton.cpp:195
        s.factorize(bad)
    with pytest.raises(RuntimeError):
        s.solve(np.ones(A.n), np.zeros(A.n))  # failed factorize leaves the solver unfactorized
    with pytest.raises(RuntimeError):
        s.set_parameters({"HIP": {"no_such_key": 1}})
    with pytest.raises(RuntimeError, match="outside"):
        W = sp.csr_matrix((np.ones(3), np.array([0, 1, 7]), np.array([0, 1, 2, 3])), shape=(3, 8))
        s._check(s._L.psolve_hip_factorize(s._h, 3, 3, W.indptr.astype(np.int32).ctypes.data,
                                           W.indices.astype(np.int32).ctypes.data, W.data.ctypes.data))
    # a different pattern afterwards is fine (Newton refactorizes every iteration)
    B = oracle.poisson7(5, 4, 3).to_scipy().tocsc()
    s.set_parameters({"HIP": {"tolerance": 1e-10}})
    s.analyze_pattern(B, B.shape[0])
    s.factorize(B)
    b = np.ones(B.shape[0])
    x = np.zeros(B.shape[0])
    s.solve(b, x)
    assert np.linalg.norm(B @ x - b) < 1e-8
    # uncompressed / unsorted input is compressed by the adapter, like MAS (BSRMatrix.cu:444-452)
    C = sp.coo_matrix(B)
    C = sp.csc_matrix((np.concatenate([C.data * 0.5, C.data * 0.5]), (np.concatenate([C.row, C.row]),
                                                                     np.concatenate([C.col, C.col]))), shape=B.shape)
    s.factorize(C)
    x[:] = 0
    s.solve(b, x)
    assert np.linalg.norm(B @ x - b) < 1e-8


# the storages PCG's product can run on (round 6: each one has its full-size test again; "plain_csr" is the contract kernel
# spmv_csr_dma of the north_star -- what any matrix without repeating rows gets)
STORAGES = {"auto": ({}, "spmv_csr_slots"),
            "plain_csr": ({"spmv_kernel": 1, "spmv_value_dict": False}, "spmv_csr_dma"),
            "pattern_dictionary": ({"spmv_kernel": -1, "spmv_value_dict": False}, "spmv_csr_pat"),
            "register_staged": ({"spmv_kernel": 0, "spmv_value_dict": False}, "spmv_csr_pipe")}


@pytest.mark.parametrize("storage", list(STORAGES))
def test_full_size_solve_properties(S, oracle, storage):
    """BASELINE.json configs[1]: 256^3 Jacobi-PCG.  The oracle takes minutes at this size, so check
    size-independent properties: recomputed true residual, error against the known x*, and the
    iteration count against the sqrt(cond) bound -- on every storage the product can stream."""
    s = S.create("HIP", "")
    N = 256
    prm, kernel = STORAGES[storage]
    s.set_parameters({"HIP": prm})
    s.generate_poisson7(N)
    n, _, _ = s.matrix_shape()
    b, xs, x = s.device_array(n), s.device_array(n), s.to_device(np.zeros(n))
    s.generate_rhs(42, b, xs)
    s.solve_device(b, x)
    info = s.get_info()
    assert s.last_spmv_kernel().startswith(kernel), s.last_spmv_kernel()
    assert info["solver_status"] == "Reach relative tolerance"
    assert info["solver_error"] < 1e-8 and info["true_residual"] < 1.2e-8
    # independent residual through the plain SpMV + dot entry points
    r = s.device_array(n)
    s.spmv_device(x, r)
    s.axpby_device(n, 1.0, b, -1.0, r)
    res = np.sqrt(s.dot_device(n, r, r) / s.dot_device(n, b, b))
    assert res < 1.2e-8
    kappa = 4 * (N + 1) ** 2 / np.pi ** 2
    assert 100 < info["solver_iter"] < 0.5 * np.sqrt(kappa) * np.log(2 / 1e-8) * 1.1
    err = np.abs(x.download() - xs.download()).max()
    assert err < 1e-8 * kappa  # |x - x*| <= cond * relative residual (loose)
    # idempotence: solving again from the solution takes 0 iterations
    s.solve_device(b, x)
    assert s.get_info()["num_iterations"] == 0


def test_non_finite_input_stops_at_once(S, oracle):
    """NaN/Inf in b or x0: Eigen would iterate on NaNs until max_iter, MAS throws "Invalid initial
    residual" (MASSolver.cu:482-486); here solve() returns immediately (no exception: non-convergence
    is not an error), reports it in solver_status and leaves x at the last finite iterate."""
    A = oracle.poisson7(8)
    M = A.to_scipy()
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"max_iter": 100000}})
    s.factorize(M)
    b = np.ones(A.n)
    b[17] = np.nan
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert info["solver_status"] == "Non-finite residual" and info["num_iterations"] == 0
    assert not x.any()
    x0 = np.zeros(A.n)
    x0[3] = np.inf
    s.solve(np.ones(A.n), x0)
    assert s.get_info()["solver_status"] == "Non-finite residual"
    # and the solver is still usable afterwards
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 1))
    x = np.zeros(A.n)
    s.solve(b, x)
    assert s.get_info()["solver_status"] == "Reach relative tolerance"
    assert np.linalg.norm(M @ x - b) / np.linalg.norm(b) < 1.5e-8


def test_tiny_and_degenerate_systems(S, oracle):
    """1x1, diagonal and zero-diagonal systems (Eigen's DiagonalPreconditioner maps a zero diagonal to 1)."""
    s = S.create("HIP", "")
    s.factorize(sp.csr_matrix(np.array([[4.0]])))
    x = np.zeros(1)
    s.solve(np.array([2.0]), x)
    assert abs(x[0] - 0.5) < 1e-15
    D = sp.diags(np.arange(1.0, 301.0)).tocsr()
    s.factorize(D)
    b = np.ones(300)
    x = np.zeros(300)
    s.solve(b, x)
    assert np.allclose(x, 1.0 / np.arange(1.0, 301.0), rtol=1e-12)
    assert s.get_info()["num_iterations"] <= 2  # Jacobi is exact on a diagonal matrix
    # a structurally missing diagonal: invdiag = 1 there (EigenSolver / DiagonalPreconditioner semantics)
    A = sp.csr_matrix(np.array([[2.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 3.0]]))
    A.eliminate_zeros()
    s.factorize(A)
    r = np.array([1.0, 2.0, 3.0])
    z = s.device_array(3)
    s.precond_apply_device(s.to_device(r), z)
    assert np.array_equal(z.download(), np.array([0.5, 2.0, 1.0]))


@pytest.mark.parametrize("precond,bs", [("jacobi", 1), ("amg", 1), ("amg", 3)])
def test_cpp_host_matrix_market(oracle, tmp_path, precond, bs):
    """The reference's Matrix-Market path (loadSymmetric + b = 1, test_linear_solver.cpp:25-50, 541-665)
    through a C++ host of the C ABI: gr_30_30 (scalar) and a block-3 elasticity matrix, written as
    symmetric Matrix Market files, solved by examples/solve_mm."""
    import subprocess
    import scipy.io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    A = oracle.elasticity_q1(6) if bs == 3 else oracle.gr_30_30()
    M = A.to_scipy()
    M = ((M + M.T) * 0.5).tocoo()  # exactly symmetric for the symmetric MM writer
    path = tmp_path / "m.mtx"
    scipy.io.mmwrite(str(path), M, symmetry="symmetric")
    p = subprocess.run([os.path.join(root, "examples", "solve_mm"), str(path), "symmetric", precond, str(bs)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    fields = dict(kv.split("=") for kv in p.stdout.split())
    assert int(fields["num_iterations"]) > 0            # REQUIRE(num_iterations > 0)
    assert float(fields["host_residual"]) < 1e-7        # REQUIRE(err / b.norm() < 1e-7)


@pytest.mark.parametrize("reorder", [0, 1])
def test_factorize_of_the_same_pattern_uploads_values_only(S, oracle, reorder):
    """factorize(host arrays) recognises the pattern it still holds on the device (hash of the caller's arrays, computed
    by host threads while the values travel) and moves 8 nnz bytes instead of 12 nnz + 4 (n + 1): Newton's case
    (Newton.cpp:189-193; MAS keeps its partition the same way, MASSolver.cu:304-321).  A different pattern, a pattern
    with one column id changed, and a factorize from device arrays in between are all noticed; under "reorder" the
    caller's-numbering copy stays next to the renumbered one."""
    A = oracle.poisson7(12, 10, 9)
    M = sp.csr_matrix(A.to_scipy())
    n, nnz = M.shape[0], M.nnz
    b = oracle.spmv(A, oracle.splitmix_vector(n, 42))
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"tolerance": 1e-10, "reorder": reorder, "reorder_min_rows": 0}})
    stat = lambda k: int(s.get_param("stats." + k))
    s.analyze_pattern(M, n)
    s.factorize(M)
    assert stat("h2d_bytes") == 12 * nnz + 4 * (n + 1) and stat("pattern_uploads") == 1
    assert s.get_param("reorder.active") == reorder
    x0 = np.zeros(n)
    s.solve(b, x0)
    base = stat("h2d_bytes")
    M2 = M.copy()
    M2.data = M.data * 1.5
    s.factorize(M2)  # same pattern, new values
    assert stat("h2d_bytes") - base == 8 * nnz and stat("pattern_uploads") == 1 and stat("matrix_uploads") == 2
    x = np.zeros(n)
    s.solve(1.5 * b, x)
    assert np.abs(x - x0).max() <= 1e-9 * np.abs(x0).max()  # the new values are what is factorized
    # one column id changed (still sorted, same counts): a different pattern
    M3 = M.copy()
    M3.indices = M.indices.copy()
    row = 5
    lo, hi = M3.indptr[row], M3.indptr[row + 1]
    free = [c for c in range(n) if c not in set(M3.indices[lo:hi]) and c > M3.indices[hi - 1]]
    M3.indices[hi - 1] = free[0]
    base = stat("h2d_bytes")
    s.factorize(M3)
    assert stat("h2d_bytes") - base == 12 * nnz + 4 * (n + 1) and stat("pattern_uploads") == 2
    base = stat("h2d_bytes")
    s.factorize(M)  # back to the first pattern: it is no longer the one on the device
    assert stat("h2d_bytes") - base == 12 * nnz + 4 * (n + 1) and stat("pattern_uploads") == 3
    # a factorize from device arrays in between invalidates what the handle believes it holds
    g = S.create("HIP", "")
    g.set_parameters({"HIP": {"reorder": 0}})
    g.factorize(M)
    s.generate_poisson7(12, 10, 9)
    base = stat("h2d_bytes")
    s.factorize(M)
    assert stat("h2d_bytes") - base == 12 * nnz + 4 * (n + 1)
    x = np.zeros(n)
    s.solve(b, x)
    assert np.abs(x - x0).max() <= 1e-9 * np.abs(x0).max()


def test_one_handle_through_many_systems_equals_fresh_handles(S, oracle):
    """A handle keeps the device blocks it releases for its next allocations (AllocMeter's cache, common.hpp): one handle taken
    through systems of different sizes, block sizes and preconditioners -- every factorize a full setup whose buffers come out
    of what the previous ones left behind, handed over full of 0xFF bytes ("lab.alloc_cache_poison") -- gives, system by
    system, the iterates and counts of a fresh handle bit for bit.  (Nothing may depend on a new allocation reading as zero, or
    on what a recycled one held.)"""
    import scipy.sparse as sp
    amg = {"coarse_enough": 200, "cheb_degree": 3, "cheb_power_iters": 20, "aggregation_min_rows": 0}
    rnd = sp.random(30000, 30000, density=12 / 30000, random_state=3, format="csr")
    rnd = -abs(rnd + rnd.T)
    rnd = (rnd + sp.diags(np.asarray(abs(rnd).sum(axis=1)).ravel() + 0.5)).tocsr()
    systems = [("poisson48 amg", oracle.poisson7(48).to_scipy(), {"precond": "amg", "block_size": 1, "amg": amg}),
               ("elasticity14 block-3 amg", oracle.elasticity_q1(14).to_scipy(), {"precond": "amg", "block_size": 3, "amg": amg}),
               ("poisson 40x36x30 jacobi", oracle.poisson7(40, 36, 30).to_scipy(), {"precond": "jacobi", "block_size": 1}),
               ("random graph amg", rnd, {"precond": "amg", "block_size": 1, "amg": amg}),
               ("elasticity10 ic", oracle.elasticity_q1(10).to_scipy(), {"precond": "ic", "block_size": 1}),
               ("poisson56 amg", oracle.poisson7(56).to_scipy(), {"precond": "amg", "block_size": 1, "amg": amg}),
               ("elasticity14 scalar amg", oracle.elasticity_q1(14).to_scipy(), {"precond": "amg", "block_size": 1, "amg": amg})]

    def run(s, M, prm):
        M = sp.csr_matrix(M)
        s.set_parameters({"HIP": dict(prm, tolerance=1e-9, max_iter=2000)})
        s.analyze_pattern(M, M.shape[0])
        s.factorize(M)
        b = M @ np.linspace(-1.0, 1.0, M.shape[0])
        x = np.zeros(M.shape[0])
        s.solve(b, x)
        return x, s.get_info()["num_iterations"]

    fresh = []
    for _, M, prm in systems:
        fresh.append(run(S.create("HIP", ""), M, prm))
    one = S.create("HIP", "")
    try:
        one.set_parameters({"HIP": {"lab.alloc_cache_poison": 1}})
        cached_seen = 0.0
        for (name, M, prm), (xf, itf) in zip(systems, fresh):
            x, it = run(one, M, prm)
            cached_seen = max(cached_seen, one.get_param("stats.device_bytes_cached"))
            assert it == itf and np.array_equal(x, xf), name
        assert cached_seen > 0  # blocks did change hands
    finally:
        one.set_parameters({"HIP": {"lab.alloc_cache_poison": 0}})


def test_jacobi_selected_after_a_factorize_under_amg(S, oracle):
    """Round 6: Jacobi's inverse diagonal is computed at factorize only where Jacobi is the preconditioner (the pass reads every
    column index: 1.7 ms of configs[2]'s refresh).  A handle that factorized under AMG and is switched to Jacobi afterwards
    computes it at the first solve -- with the poisoned allocator cache of the session, a stale or never-written diagonal
    would show -- and gives a fresh Jacobi handle's iterates bit for bit."""
    M = sp.csr_matrix(oracle.poisson7(24, 20, 22).to_scipy())
    b = M @ np.linspace(-1.0, 1.0, M.shape[0])

    def solve(s):
        x = np.zeros(M.shape[0])
        s.solve(b, x)
        return x, s.get_info()["num_iterations"]

    fresh = S.create("HIP", "")
    fresh.set_parameters({"HIP": {"precond": "jacobi", "tolerance": 1e-9}})
    fresh.analyze_pattern(M, M.shape[0])
    fresh.factorize(M)
    xf, itf = solve(fresh)
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-9, "amg": {"coarse_enough": 200}}})
    s.analyze_pattern(M, M.shape[0])
    s.factorize(M)
    xa, ita = solve(s)
    assert ita < itf
    s.set_parameters({"HIP": {"precond": "jacobi"}})
    x, it = solve(s)
    assert it == itf and np.array_equal(x, xf)
    z = s.device_array(M.shape[0])
    s.precond_apply_device(s.to_device(b), z)
    assert np.allclose(z.download(), b / M.diagonal(), rtol=1e-15, atol=0)


@pytest.mark.parametrize("bs", [1, 3])
def test_kept_symbolic_work_carries_the_pattern_it_was_built_for(S, oracle, bs):
    """Round-4 advice: the pattern dictionary, the block graph of the BSR-3 copy and the IC ordering were kept while "the
    pattern is the one of the previous factorize call" -- but a call that FAILED after the pattern's identity was recorded
    lies between: X factorized; Y (same rows and entries, another pattern) fails on a non-finite diagonal; Newton retries Y
    (Newton.cpp:195 catches the runtime_error).  The retry must multiply by Y, not by Y's values on X's pattern."""
    X = sp.csr_matrix(oracle.poisson7(8, 4, 6).to_scipy())
    Y = sp.csr_matrix(oracle.poisson7(4, 8, 6).to_scipy())
    if bs == 3:
        T = sp.csr_matrix(np.array([[2.0, 0.3, 0.1], [0.3, 2.0, 0.2], [0.1, 0.2, 2.0]]))
        X, Y = sp.kron(X, T, format="csr"), sp.kron(Y, T, format="csr")
    for M in (X, Y):
        M.sort_indices()
    assert X.shape == Y.shape and X.nnz == Y.nnz and not np.array_equal(X.indices, Y.indices)
    n = X.shape[0]
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"block_size": bs, "reorder": 0}})
    s.analyze_pattern(X, n)
    s.factorize(X)
    assert (s.get_param("spmv_patterns") > 0) if bs == 1 else (s.get_param("bsr3_active") == 1)
    v = oracle.splitmix_vector(n, 5)
    y = s.device_array(n)
    s.spmv_device(s.to_device(v), y)
    assert np.allclose(y.download(), X @ v, rtol=1e-13, atol=1e-13)
    bad = Y.copy()
    bad.data = bad.data.copy()
    bad.data[bad.indptr[7]:bad.indptr[8]][bad.indices[bad.indptr[7]:bad.indptr[8]] == 7] = np.nan
    with pytest.raises(RuntimeError, match="non-finite"):
        s.factorize(bad)
    s.factorize(Y)  # the retry
    s.spmv_device(s.to_device(v), y)
    assert np.allclose(y.download(), Y @ v, rtol=1e-13, atol=1e-13)
    b = Y @ v
    x = np.zeros(n)
    s.solve(b, x)
    assert np.linalg.norm(Y @ x - b) / np.linalg.norm(b) < 1e-7
    # and the same pattern again is still recognised (the caches are kept where they are valid)
    s.factorize(Y)
    s.spmv_device(s.to_device(v), y)
    assert np.allclose(y.download(), Y @ v, rtol=1e-13, atol=1e-13)


def test_two_handles_hold_their_own_knobs_and_solve_concurrently(S, oracle):
    """instances are independent (the reference's MAS handle owns a private stream and pool,
    MASSolver.cu:186-196; nothing is process-wide).  Until round 5 the "lab.*" knobs were process-wide globals that
    psolve_hip_set_param on ANY handle wrote.  Three handles with different knobs -- row kinds by spmv_csr_kind
    ("lab.kind_slots" 0, 4 rows per lane), row kinds by spmv_csr_slots (the default), the plain CSR stream -- keep what each
    was given, whatever is set on the others afterwards, and solve from three host threads at once (ctypes releases the GIL:
    the three solves are in the library together) with the iteration counts and iterates of their own sequential solves."""
    import threading
    N = 72
    knobs = [({"lab.kind_slots": 0, "lab.kind_unroll": 4}, "spmv_csr_kind"), ({}, "spmv_csr_slots"),
             ({"spmv_kernel": 1, "spmv_value_dict": False}, "spmv_csr_dma")]
    hs = []
    for prm, _ in knobs:
        s = S.create("HIP", "")
        s.set_parameters({"HIP": dict(prm, tolerance=1e-9)})
        hs.append(s)
    hs[1].set_parameters({"HIP": {"lab.kind_unroll": 1, "lab.kind_slots": 1, "lab.kind_sched": 1}})  # (set AFTER the first handle got its own)
    sys_ = []
    for s in hs:
        s.generate_poisson7(N)
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        sys_.append((n, b, x))

    def solve(k, out):
        s, (n, b, x) = hs[k], sys_[k]
        for rep in range(4):
            s.axpby_device(n, 0.0, b, 0.0, x)
            s.solve_device(b, x)
            i = s.get_info()
            out.append((int(i["num_iterations"]), x.download(), s.last_spmv_kernel(), i["true_residual"]))

    seq = [[] for _ in hs]
    for k in range(len(hs)):
        solve(k, seq[k])
    par = [[] for _ in hs]
    th = [threading.Thread(target=solve, args=(k, par[k])) for k in range(len(hs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k, (_, kernel) in enumerate(knobs):
        assert len(par[k]) == 4
        for (it, x, kn, res), (it0, x0, kn0, _) in zip(par[k], seq[k]):
            assert kn.startswith(kernel) and kn0.startswith(kernel), (k, kn, kn0)
            assert it == it0 and np.array_equal(x, x0) and res < 1.5e-9, (k, it, it0)
    # the three storages agree with each other to rounding (different summation orders of p.q)
    assert abs(seq[0][0][0] - seq[2][0][0]) <= 1 and np.abs(seq[0][0][1] - seq[2][0][1]).max() < 1e-7
