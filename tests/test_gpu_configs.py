"""Every configuration BASELINE.json names, AT ITS SIZE, on one MI355X.

Where the CPU oracle still finishes in seconds (64^3) the comparison is against it; at full size the checks are
the size-independent properties of the domain: the recomputed true residual, the error against the known x*
(b = A x*), idempotence of a re-solve, iteration counts against the condition-number bound or against Jacobi,
and -- for the row-partitioned configuration -- agreement of the sharded solve with the single-device solve."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


def test_config0_poisson64_vs_oracle(S, oracle):
    """configs[0]: 3-D 7-point Poisson 64^3, Eigen::ConjugateGradient semantics -- through the HOST entry points,
    against oracle.cg_eigen: same iteration count (+-1: tree- vs chunk-reduced dots), same solution, same error."""
    A = oracle.poisson7(64)
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    s = S.create("HIP", "")
    s.set_parameters({"HIP": {"tolerance": 1e-8}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    xo, ito, erro = oracle.cg_eigen(A, b, tol=1e-8)
    assert abs(info["solver_iter"] - ito) <= 1
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    assert abs(info["solver_error"] - erro) <= 0.05 * erro or info["solver_iter"] != ito
    assert info["true_residual"] < 1.5e-8


def test_config2_elasticity_3m_dof_block_amg(S, oracle):
    """configs[2]: 3-D linear elasticity, Q1, M = 100 -> 3 000 000 DOF, ~2.4e8 nonzeros, block-3 Chebyshev-AMG PCG.
    The system is generated on the device (bit-equal to the oracle's generator at M = 6, checked first)."""
    small = S.create("HIP", "")
    small.generate_elasticity_q1(6)
    assert small.get_param("spmv_rows_per_block") < 256
    Ao = oracle.elasticity_q1(6)
    n, nnz, _ = small.matrix_shape()
    assert (n, nnz) == (Ao.n, Ao.nnz)
    xs = oracle.splitmix_vector(Ao.n, 3)
    y = small.device_array(n)
    small.spmv_device(small.to_device(xs), y)
    # same matrix as the oracle's generator (81-entry rows are summed by several lanes: rounding of the row sums
    # differs from the scalar loop, the entries do not)
    yo = oracle.spmv(Ao, xs)
    assert np.abs(y.download() - yo).max() <= 1e-14 * np.abs(Ao.to_scipy()).dot(np.abs(xs)).max()
    e = np.zeros(Ao.n)
    for probe in (7, Ao.n // 2 + 1, Ao.n - 2):  # single columns: entries themselves, bit for bit
        e[:] = 0
        e[probe] = 1.0
        small.spmv_device(small.to_device(e), y)
        assert np.array_equal(y.download(), oracle.spmv(Ao, e))
    del small

    M = 100
    s = S.create({"solver": "HIP", "HIP": {"precond": "amg", "block_size": 3, "tolerance": 1e-8, "max_iter": 20000,
                                           "amg": {"ncycle": 1, "cheb_degree": 2, "cheb_lower": 0.1, "cheb_power_iters": 20}}})
    s.generate_elasticity_q1(M)
    n, nnz, _ = s.matrix_shape()
    assert n == 3 * M ** 3 and nnz > 75 * n
    assert s.get_param("bsr3_active") == 1
    b, xs, x = s.device_array(n), s.device_array(n), s.to_device(np.zeros(n))
    s.generate_rhs(42, b, xs)
    s.solve_device(b, x)
    info = s.get_info()
    assert info["solver_status"] == "Reach relative tolerance" and info["amg_levels"] >= 3
    assert info["solver_error"] < 1e-8 and info["true_residual"] < 1.5e-8
    r = s.device_array(n)  # residual again through the plain product + dot entry points
    s.spmv_device(x, r)
    s.axpby_device(n, 1.0, b, -1.0, r)
    assert np.sqrt(s.dot_device(n, r, r) / s.dot_device(n, b, b)) < 1.5e-8
    amg_iters = info["num_iterations"]
    err = np.abs(x.download() - xs.download()).max()
    assert err < 1e-3  # cond ~ 1e5: |x - x*| <= cond * 1e-8 |x*|
    s.solve_device(b, x)  # idempotence
    assert s.get_info()["num_iterations"] <= 1
    # Jacobi on the same system, capped: AMG must need fewer than a third of ITS iterations
    s.set_parameters({"HIP": {"precond": "jacobi", "max_iter": 3 * amg_iters + 3}})
    x0 = s.to_device(np.zeros(n))
    s.solve_device(b, x0)
    assert s.get_info()["solver_status"] == "Reach max iterations"


def test_config2_with_shuffled_nodes_is_renumbered_backwards(S):
    """configs[2]'s stiffness matrix with its nodes in a pseudo-random order (what a mesh generator may leave): by default
    the system is renumbered at factorize by the breadth-first order of the node graph READ BACKWARDS ("reorder_reverse");
    AMGCL's aggregation sweep follows the numbering and builds more regular aggregates that way -- within a few
    iterations of the grid numbering's count (37), where the forward order needs 52."""
    from polysolve_amd import HIPSolver
    M = 100
    its = {}
    for rev in (True, False):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"precond": "amg", "block_size": 3, "tolerance": 1e-8, "max_iter": 20000, "reorder_reverse": rev,
                                  "amg": {"ncycle": 1, "cheb_degree": 2, "cheb_lower": 0.1, "cheb_higher": 1.1, "cheb_power_iters": 20,
                                          "sa_relax": 1.3}}})
        assert s.get_param("reorder") == 2
        s.generate_elasticity_q1_permuted(M, mode=1, seed=7)
        assert s.get_param("reorder.active") == 1 and s.get_param("bsr3_active") == 1
        n = s.matrix_shape()[0]
        b, xs, x = s.device_array(n), s.device_array(n), s.to_device(np.zeros(n))
        s.generate_rhs(42, b, xs)
        s.solve_device(b, x)
        info = s.get_info()
        assert info["true_residual"] < 1.5e-8 and np.abs(x.download() - xs.download()).max() < 1e-3
        its[rev] = info["num_iterations"]
        for a in (b, xs, x):
            a.free()
        del s
    assert its[True] <= 42 and its[True] < its[False]


@pytest.mark.parametrize("prm,kernel", [({}, "spmv_csr_slots"), ({"spmv_kernel": 1, "spmv_value_dict": False}, "spmv_csr_dma")],
                         ids=["auto", "plain_csr"])
def test_config3_poisson512_single_device_properties(S, prm, kernel):
    """configs[3]'s system, 512^3 = 134 M DOF (937 M nonzeros, 11 GB of CSR), on ONE device: it fits -- on the backend's own
    pick for this grid and on the plain CSR stream (the contract kernel)."""
    N = 512
    s = S.create("HIP", "")
    s.set_parameters({"HIP": dict({"tolerance": 1e-8, "max_iter": 20000}, **prm)})
    s.generate_poisson7(N)
    n, nnz, _ = s.matrix_shape()
    assert n == N ** 3 and nnz == 7 * N ** 3 - 6 * N ** 2
    b, xs, x = s.device_array(n), s.device_array(n), s.to_device(np.zeros(8))
    del x
    x = s.device_array(n)
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.generate_rhs(42, b, xs)
    s.axpby_device(n, 0.0, b, 0.0, x)  # x0 = 0
    s.solve_device(b, x)
    info = s.get_info()
    assert s.last_spmv_kernel().startswith(kernel), s.last_spmv_kernel()
    assert info["solver_status"] == "Reach relative tolerance"
    assert info["solver_error"] < 1e-8 and info["true_residual"] < 1.2e-8
    kappa = 4 * (N + 1) ** 2 / np.pi ** 2
    assert 300 < info["solver_iter"] < 0.5 * np.sqrt(kappa) * np.log(2 / 1e-8) * 1.1
    s.axpby_device(n, 1.0, xs, -1.0, x)  # x := x* - x
    assert np.sqrt(s.dot_device(n, x, x) / s.dot_device(n, xs, xs)) < 1e-8 * kappa


@pytest.mark.parametrize("storage", ["auto", "plain_csr"])
@pytest.mark.parametrize("single", [1, 0])
def test_config3_row_partition_4_shards_128(S, oracle, single, storage):
    """configs[3]'s algorithm -- rows 1-D partitioned, halo exchange overlapped with the interior rows, all-reduced
    recurrences -- with 4 shards at 128^3 (2.1 M DOF) against the single-device solve of the same system: same
    right-hand side bit for bit, iteration counts within 2, iterates within 1e-6.  Round 6: also on the plain CSR stream, the
    storage `bench.py --gpus N` times (interior / boundary row-block lists on spmv_csr_dma)."""
    from polysolve_amd import HIPSolver, LocalGroup
    N, world = 128, 4
    stor = {} if storage == "auto" else {"spmv_kernel": 1, "spmv_value_dict": False}
    ref = S.create("HIP", "")
    ref.set_parameters({"HIP": {"tolerance": 1e-8}})
    ref.generate_poisson7(N)
    n = N ** 3
    b, x = ref.device_array(n), ref.to_device(np.zeros(n))
    ref.generate_rhs(42, b)
    ref.solve_device(b, x)
    ri = ref.get_info()
    xb, bb = x.download(), b.download()
    cuts = [0, 32, 64, 96, 128]
    group = LocalGroup(world)
    out, errors = [None] * world, []

    def run(rank):
        try:
            s = HIPSolver("")
            s.comm_init_local(group, rank)
            s.set_parameters({"HIP": dict({"tolerance": 1e-8, "dist_single_reduction": bool(single), "profile_spmv": 8}, **stor)})
            s.generate_poisson7(N, N, N, cuts[rank], cuts[rank + 1])
            nl, _, nh = s.matrix_shape()
            lb, lx = s.device_array(nl), s.to_device(np.zeros(nl))
            s.generate_rhs(42, lb)
            s.solve_device(lb, lx)
            out[rank] = (lb.download(), lx.download(), s.get_info(), nh, s.last_spmv_kernel(), s.info_struct().spmv_ms_avg)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    assert np.array_equal(np.concatenate([o[0] for o in out]), bb)
    assert [o[3] for o in out] == [N * N, 2 * N * N, 2 * N * N, N * N]
    infos = [o[2] for o in out]
    assert len({i["solver_iter"] for i in infos}) == 1
    assert abs(infos[0]["solver_iter"] - ri["solver_iter"]) <= 2
    assert infos[0]["true_residual"] < 1.5e-8
    if storage == "plain_csr":
        assert all(o[4].startswith(("spmv_csr_dma", "spmv_csr_pipe")) for o in out), [o[4] for o in out]
    assert all(o[5] > 0 for o in out)  # (the sampled product of a shard: two launches around the halo wait, bracketed by events)
    xs = np.concatenate([o[1] for o in out])
    assert np.abs(xs - xb).max() <= 1e-6 * np.abs(xb).max()


def test_config3_global_amg_4_shards_128(S):
    """configs[3]'s partition with the AMG preconditioner: 4 shards at 128^3 through the in-process multi-device
    handle (host contract), the hierarchy built ON the shards (amg.dist_global 2, the default) -- the iteration count
    must stay within 1.3x of the single-device count (round 1's per-shard hierarchies: 13 -> 46), same solution, and a
    shard holds about a quarter of what the single device holds (no rank gathers the matrix)."""
    import oracle as O
    N = 128
    A = O.poisson7(N)
    M = A.to_scipy().tocsc()
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    amg = {"ncycle": 1, "cheb_degree": 2, "cheb_lower": 0.1, "cheb_power_iters": 20}
    res = {}
    for name, devices in (("one", [0]), ("four", [0, 0, 0, 0])):
        s = S.create({"solver": "HIP", "HIP": {"devices": devices, "precond": "amg", "tolerance": 1e-8, "amg": amg}})
        s.analyze_pattern(M, A.n)
        s.factorize(M)
        x = np.zeros(A.n)
        s.solve(b, x)
        res[name] = (x, s.get_info(), s.get_param("stats.device_bytes"))
    i1, i4 = res["one"][1], res["four"][1]
    assert res["four"][2] <= 0.45 * res["one"][2], (res["four"][2], res["one"][2])  # the largest shard's device bytes
    assert i4["solver_status"] == "Reach relative tolerance" and i4["true_residual"] < 1.5e-8
    assert i4["num_iterations"] <= 1.3 * i1["num_iterations"] and i4["num_iterations"] >= i1["num_iterations"] - 1
    assert abs(i4["amg_levels"] - i1["amg_levels"]) <= 1
    # (two different preconditioners, both stopped at ||r|| / ||b|| < 1e-8)
    assert np.abs(res["four"][0] - res["one"][0]).max() <= 5e-6 * np.abs(res["one"][0]).max()


def test_config4_newton_128_through_host_entry_points(S, oracle):
    """configs[4]: the Newton inner loop (Newton.cpp:173-214) on a 128^3 problem (2.1 M unknowns), Hessian solves
    through analyze_pattern / factorize / solve on HOST arrays, AMG preconditioner refreshed numerically while
    the pattern stays (Newton refactorizes every iteration, Newton.cpp:189-193)."""
    from test_gpu_newton import QuarticProblem, newton
    N = 128
    A = oracle.poisson7(N).to_scipy()
    rng = np.random.default_rng(0)
    b = rng.uniform(-1, 1, A.shape[0]) * 50
    problem = QuarticProblem(A, b, c=2.0)
    solver = S.create({"solver": "HIP", "HIP": {"precond": "amg", "tolerance": 1e-10, "absolute_tolerance": 1e-9,
                                                "max_iter": 5000, "amg": {"ncycle": 1, "cheb_degree": 3,
                                                                          "cheb_lower": 0.1, "cheb_power_iters": 20}}})
    x, its, infos = newton(problem, solver, np.zeros(A.shape[0]))
    assert np.linalg.norm(problem.gradient(x)) < 1e-7
    assert 2 <= its <= 15
    assert all(i["solver_status"] in ("Reach absolute tolerance", "Reach relative tolerance") for i in infos)
    assert all(i["num_iterations"] < 60 for i in infos)
    # every factorize after the first kept the aggregates and patterns (same sparsity, new diagonal)
    assert solver.get_param("stats.amg_setups") + solver.get_param("stats.amg_refreshes") == its
    assert solver.get_param("stats.amg_refreshes") >= its - 2
    assert solver.get_param("stats.matrix_uploads") == its
    # ... and crossed PCIe with its VALUES only: the pattern was recognised on the host (hash of the caller's arrays) and
    # stayed on the device -- 12 nnz + 4 (n + 1) bytes the first time, 8 nnz bytes per refactorize, + b and x per solve
    n, nnz = A.shape[0], A.nnz
    assert solver.get_param("stats.pattern_uploads") == 1
    assert solver.get_param("stats.h2d_bytes") == 4 * (n + 1) + 4 * nnz + its * 8 * nnz + its * 2 * 8 * n


def test_config1_under_a_scattered_numbering_is_renumbered(S):
    """configs[1] (256^3 Jacobi-PCG) as an unstructured mesh would hand it over: the same matrix under one pseudo-random
    renumbering of all rows.  At this size the checks are properties: the default renumbers it (a bijection; the gathers
    of 64 consecutive rows back within a few lines), the solve returns x* in the CALLER's numbering to the accuracy of
    the grid-numbered solve, with the iteration count of the caller's-numbering solve (Jacobi-PCG does not depend on the
    numbering), at several times its speed; a refactorize of the same pattern keeps the order."""
    from polysolve_amd import HIPSolver
    N = 256
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=7)
    assert s.get_param("reorder.active") == 1 and s.get_param("spmv_patterns") == 0
    assert s.get_param("reorder.spread_before") > 5.0 and s.get_param("reorder.spread_after") < 2.0
    assert s.get_param("reorder.levels") == 3 * (N - 1) + 1  # the eccentricity of a corner of the grid, + 1
    n = s.matrix_shape()[0]
    perm, _ = s.reorder_perm()
    assert np.array_equal(np.sort(perm), np.arange(n))
    b, xs, x = s.device_array(n), s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b, xs)
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    i1 = s.get_info()
    assert i1["true_residual"] < 1.5e-8 and np.abs(x.download() - xs.download()).max() < 1e-4
    assert s.get_param("stats.reorder_searches") == 1
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=7)  # the same pattern again: no new search
    assert s.get_param("reorder.active") == 1 and s.get_param("stats.reorder_searches") == 1
    s.set_parameters({"HIP": {"reorder": 0}})
    s.generate_poisson7_permuted(N, N, N, mode=1, seed=7)
    assert s.get_param("reorder.active") == 0
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x)
    i0 = s.get_info()
    assert abs(i0["solver_iter"] - i1["solver_iter"]) <= 2 and i0["true_residual"] < 1.5e-8
    assert i1["time_solve"] < 0.5 * i0["time_solve"]
