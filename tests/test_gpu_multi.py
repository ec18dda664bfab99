"""The in-process multi-device handle (psolve_hip_create_multi): ONE Solver object,
the reference's host contract (Solver.hpp:90-131: set_parameters / analyze_pattern / factorize / solve on
HOST arrays), the matrix row-partitioned over several device contexts behind it.

A gpurun box has one GPU, so the shards are put on the same device (`devices = [0, 0, ...]`): repeated ids
select the host-synchronised loopback group instead of the in-process RCCL clique -- same halo plan, column
remap, pack / exchange, overlap and all-reduced recurrences, on real kernels.  With >= 2 GPUs visible the
distinct-id leg runs too (ncclCommInitAll).

Tolerances as tests/test_gpu_solver.py: iteration count within 1 (Eigen's recurrence) or 2 (single-reduction
recurrences, the default on shards) of the oracle's, |x - x_oracle| <= 1e-6 |x|_inf."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from polysolve_amd import Solver
    return Solver


def _device_count():
    import ctypes as C
    from polysolve_amd import _lib
    c = C.c_int()
    _lib.load().psolve_hip_device_count(C.byref(c))
    return c.value


@pytest.mark.parametrize("devices,grid,precond,single", [([0, 0], (12, 10, 16), "", 1), ([0, 0, 0], (9, 8, 14), "", 0),
                                                         ([0, 0, 0, 0], (16, 16, 16), "Eigen::IdentityPreconditioner", 1),
                                                         ([0, 0], (40, 40, 24), "", 1)])
def test_host_contract_on_shards_matches_oracle(S, oracle, devices, grid, precond, single):
    """Solver::create(json) with params["HIP"]["devices"]: analyze_pattern / factorize / solve on host arrays,
    rows split inside the handle; solution, iteration count and residual against the oracle's global solve."""
    A = oracle.poisson7(*grid)
    M = A.to_scipy().tocsc()
    xs = oracle.splitmix_vector(A.n, 42)
    b = oracle.spmv(A, xs)
    s = S.create({"solver": "HIP", "precond": precond,
                  "HIP": {"devices": devices, "tolerance": 1e-8, "dist_single_reduction": bool(single)}})
    assert s.get_param("devices") == len(devices)
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    # contiguous shards that cover the rows, every one on the device asked for
    rows = [s.shard_rows(r) for r in range(len(devices))]
    assert rows[0][0] == 0 and rows[-1][1] == A.n
    assert all(rows[r][1] == rows[r + 1][0] and rows[r][0] < rows[r][1] for r in range(len(devices) - 1))
    assert [r[2] for r in rows] == devices
    nnz_shard = [int(A.rowptr[r1] - A.rowptr[r0]) for r0, r1, _ in rows]
    assert max(nnz_shard) < 1.25 * A.nnz / len(devices)  # balanced by nonzeros
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    pname = "none" if precond else "jacobi"
    xo, ito, _ = oracle.cg_eigen(A, b, precond=pname, tol=1e-8)
    assert abs(info["solver_iter"] - ito) <= (2 if single else 1)
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    assert info["true_residual"] < 1.5e-8 and np.linalg.norm(M @ x - b) < 1.5e-8 * np.linalg.norm(b)
    assert info["solver_status"] == "Reach relative tolerance"
    # x is the initial guess (Solver.hpp:119-127): a second solve from the solution does nothing
    s.solve(b, x)
    assert s.get_info()["num_iterations"] <= 1
    # a new matrix with the same pattern (Newton), then a different pattern, through the same handle
    M2 = (M * 2.0).tocsc()
    s.factorize(M2)
    x2 = np.zeros(A.n)
    s.solve(b, x2)
    assert np.abs(2 * x2 - xo).max() <= 2e-6 * np.abs(xo).max()
    B = oracle.poisson7(grid[0], grid[1], grid[2] + 3)
    MB = B.to_scipy().tocsc()
    bb = MB @ np.ones(B.n)
    s.analyze_pattern(MB, B.n)
    s.factorize(MB)
    xb = np.zeros(B.n)
    s.solve(bb, xb)
    assert np.abs(xb - 1.0).max() < 1e-5


def test_multi_device_amg_and_blocks(S, oracle):
    """precond = amg on shards (one hierarchy per shard, additive Schwarz) and block_size 3 (cuts at block
    multiples) through the host contract."""
    A = oracle.elasticity_q1(8)
    M = A.to_scipy().tocsc()
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, A.n)
    s = S.create({"solver": "HIP", "HIP": {"devices": [0, 0, 0], "precond": "amg", "block_size": 3, "tolerance": 1e-9,
                                           "amg": {"coarse_enough": 100, "aggregation_min_rows": 0}}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    for r in range(3):
        r0, r1, _ = s.shard_rows(r)
        assert r0 % 3 == 0 and r1 % 3 == 0
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert info["amg_levels"] >= 2
    assert np.linalg.norm(M @ x - b) < 1.5e-9 * np.linalg.norm(b)
    sj = S.create({"solver": "HIP", "HIP": {"devices": [0, 0, 0], "tolerance": 1e-9}})
    sj.factorize(M)
    xj = np.zeros(A.n)
    sj.solve(b, xj)
    assert info["num_iterations"] < sj.get_info()["num_iterations"] / 2


def test_multi_handle_errors(S, oracle):
    """Error behaviour of the reference contract on a multi-device handle: failures are exceptions with the
    shard named, nothing hangs, and the handle stays usable."""
    from polysolve_amd import HIPSolver
    A = oracle.poisson7(8)
    M = A.to_scipy().tocsc()
    s = HIPSolver("", devices=[0, 0])
    with pytest.raises(RuntimeError, match="solve before factorize|Size mismatch"):
        s.solve(np.ones(A.n), np.zeros(A.n))
    tiny = sp.identity(8, format="csc")
    with pytest.raises(RuntimeError, match="too small to partition"):
        s.factorize(tiny)
    # a NaN on the diagonal of the second shard only: factorize fails there, the first shard is woken up
    bad = M.copy().tolil()
    bad[A.n - 1, A.n - 1] = np.nan
    with pytest.raises(RuntimeError, match="shard 1.*non-finite diagonal"):
        s.factorize(bad.tocsc())
    # the same with the global AMG hierarchy selected: the bad shard must fail BEFORE the others enter the gather of
    # the matrix (a collective), and everybody must leave factorize together
    a = HIPSolver("", devices=[0, 0, 0])
    a.set_parameters({"HIP": {"precond": "amg", "amg": {"coarse_enough": 50, "aggregation_min_rows": 0}}})
    with pytest.raises(RuntimeError, match="shard 2.*non-finite diagonal"):
        a.factorize(bad.tocsc())
    a.factorize(M)
    xa = np.zeros(A.n)
    a.solve(M @ np.ones(A.n), xa)
    assert np.abs(xa - 1).max() < 1e-6 and a.get_info()["amg_levels"] >= 2
    # device-pointer entry points belong to one device
    with pytest.raises(RuntimeError, match="single-device handle"):
        s.device_array(16)
    with pytest.raises(RuntimeError, match="device id out of range|create_multi"):
        HIPSolver("", devices=[0, 99])
    s.factorize(M)  # still usable
    x = np.zeros(A.n)
    b = M @ np.ones(A.n)
    s.solve(b, x)
    assert np.abs(x - 1).max() < 1e-6


def test_adopted_arrays_survive_a_shard_factorize(S, oracle):
    """psolve_hip_factorize_device on a shard never writes the caller's arrays (the local-id remap goes to a
    private copy), so factorizing the SAME device arrays twice -- Newton with a constant pattern -- gives the
    same halo plan and the same solution the second time."""
    import threading
    from polysolve_amd import HIPSolver, LocalGroup
    world, (nx, ny, nz) = 2, (10, 9, 12)
    cuts = [0, 5, 12]
    group = LocalGroup(world)
    out, errors = [None] * world, []
    Ag = oracle.poisson7(nx, ny, nz)
    bg = oracle.spmv(Ag, oracle.splitmix_vector(Ag.n, 42))

    def run(rank):
        try:
            s = HIPSolver("")
            s.comm_init_local(group, rank)
            Al = oracle.poisson7(nx, ny, nz, cuts[rank], cuts[rank + 1])  # global column ids
            r0 = cuts[rank] * nx * ny
            s.set_partition(Ag.n, r0, r0 + Al.n)
            ptr, col, val = s.to_device(Al.rowptr), s.to_device(Al.col), s.to_device(Al.val)
            res = []
            for _ in range(2):
                s.factorize_device(Al.n, Al.nnz, ptr, col, val)
                assert np.array_equal(col.download(), Al.col)  # untouched: still global ids
                b, x = s.to_device(bg[r0:r0 + Al.n]), s.to_device(np.zeros(Al.n))
                s.solve_device(b, x)
                res.append((x.download(), s.get_info(), s.matrix_shape()[2]))
            out[rank] = res
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    xo, ito, _ = oracle.cg_eigen(Ag, bg, tol=1e-8)
    for k in range(2):
        x = np.concatenate([out[r][k][0] for r in range(world)])
        assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
        assert all(out[r][k][2] == nx * ny for r in range(world))
        assert abs(out[0][k][1]["solver_iter"] - ito) <= 2


@pytest.mark.rccl_multi  # (counted by conftest.py: "rccl_world_gt1_tests_executed")
@pytest.mark.skipif("_device_count() < 2")
def test_in_process_rccl_clique(S, oracle):
    """Distinct device ids: ncclCommInitAll inside one process, one host thread per device."""
    nd = min(_device_count(), 8)
    A = oracle.poisson7(32, 32, 16 * nd)
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    s = S.create({"solver": "HIP", "HIP": {"devices": list(range(nd)), "tolerance": 1e-8}})
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-8)
    assert abs(s.get_info()["solver_iter"] - ito) <= 2
    assert np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()


def _one_shard_fails_in_solve(S, oracle, devices):
    """A shard that leaves the collective sequence of a solve (here: an injected failure before its first collective)
    must not leave the others blocked for ever: the call returns an error that names the shard, and the handle works
    again afterwards WITHOUT a new factorize (an aborted RCCL clique is made again at the next call; what the shards
    hold does not live in the communicators)."""
    from polysolve_amd import HIPSolver
    A = oracle.poisson7(16, 16, 8 * len(devices))
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    s = HIPSolver("", devices=devices)
    s.set_parameters({"HIP": {"tolerance": 1e-8}})
    s.factorize(M)
    x = np.zeros(A.n)
    s.solve(b, x)
    it0 = s.get_info()["solver_iter"]
    s._set("fault.solve_rank", len(devices) - 1)
    with pytest.raises(RuntimeError, match=f"shard {len(devices) - 1}.*injected fault"):
        s.solve(b, np.zeros(A.n))
    x2 = np.zeros(A.n)
    s.solve(b, x2)  # the factorization survives (loopback group re-armed / RCCL clique made again)
    assert s.get_info()["solver_iter"] == it0 and np.array_equal(x, x2)
    # a failure every rank agrees on (a non-finite diagonal entry on ONE shard, Newton.cpp:191-202 catches it) leaves
    # the collective sequence aligned: no abort, and the next factorize / solve run on the same communicators
    Mbad = M.copy()
    Mbad.data = Mbad.data.copy()
    Mbad.data[Mbad.indptr[3]:Mbad.indptr[4]][Mbad.indices[Mbad.indptr[3]:Mbad.indptr[4]] == 3] = np.nan
    with pytest.raises(RuntimeError, match="non-finite diagonal"):
        s.factorize(Mbad)
    assert s.get_param("dist.comm_aborted") == 0
    with pytest.raises(RuntimeError, match="solve before factorize"):
        s.solve(b, np.zeros(A.n))
    s.factorize(M)
    x3 = np.zeros(A.n)
    s.solve(b, x3)
    assert s.get_info()["solver_iter"] == it0 and np.array_equal(x, x3)


def test_one_shard_failing_in_solve_frees_the_others_loopback(S, oracle):
    _one_shard_fails_in_solve(S, oracle, [0, 0, 0])


@pytest.mark.rccl_multi  # (counted by conftest.py: "rccl_world_gt1_tests_executed")
@pytest.mark.skipif("_device_count() < 2")
def test_one_shard_failing_in_solve_frees_the_others_rccl(S, oracle):
    _one_shard_fails_in_solve(S, oracle, list(range(min(_device_count(), 8))))


@pytest.mark.rccl_multi  # (counted by conftest.py: "rccl_world_gt1_tests_executed")
@pytest.mark.skipif("_device_count() < 2")
def test_bench_on_two_gpus(oracle):
    """`python bench.py --gpus 2` (its own launcher: one rank per GPU over RCCL) on the first multi-GPU box: one JSON
    line for 2 GPUs, the same system as N = 1 (iterations within 2: single-reduction recurrences on shards), both
    communicators of a rank in use (all-reduce on the main stream, halo exchange on the comm stream)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    lines = {}
    for n in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--grid", "96", "--steps",
                              "2", "--warmup", "1", "--no-cpu-baseline", "--no-north-star", "--no-extra"],
                             capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        js = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1
        lines[n] = json.loads(js[0])
    assert lines[2]["n_gpus"] == 2 and lines[2]["scaling"] == "strong" and lines[2]["config"]["partition"] == "2 z-slab(s)"
    assert abs(lines[2]["iterations"] - lines[1]["iterations"]) <= 2
    assert lines[2]["true_residual"] < 1.5e-8 and lines[2]["value"] > 0
    assert lines[2]["config"]["halo_per_gpu"] == 96 * 96


@pytest.mark.rccl_multi  # (counted by conftest.py: "rccl_world_gt1_tests_executed")
@pytest.mark.skipif("_device_count() < 2")
def test_one_process_per_gpu_vs_oracle(oracle, tmp_path):
    """The launcher's shape (one process per GPU, psolve_hip_comm_init from a shared unique id), without torch: two
    processes solve the row-partitioned 7-point system -- Jacobi with both reduction schemes, then AMG -- and the
    gathered solution is the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world, grid = 2, (24, 20, 28)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "rccl_worker.py"), str(r), str(world),
                               str(tmp_path)] + [str(g) for g in grid], cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    A = oracle.poisson7(*grid)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-8)
    for tag, slack in (("jacobi1", 2), ("jacobi2", 1)):
        x = np.concatenate([np.load(tmp_path / f"x_{tag}_{r}.npy") for r in range(world)])
        its = int(np.load(tmp_path / f"it_{tag}_0.npy"))
        assert abs(its - ito) <= slack and np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max(), tag
    x = np.concatenate([np.load(tmp_path / f"x_amg_{r}.npy") for r in range(world)])
    assert np.linalg.norm(A.to_scipy() @ x - b) <= 1.5e-8 * np.linalg.norm(b)


@pytest.mark.parametrize("devices,M,repl", [([0, 0, 0], 12, 200), ([0, 0], 10, 1)])
def test_block3_hierarchy_on_shards(S, oracle, devices, M, repl):
    """block_size 3 (AMGCL_Block<3>, AMGCL.cpp:243-302) on several devices: ONE block hierarchy built on the shards
    (amg.dist_global 2) -- node aggregation inside the shards, block-smoothed prolongation with the aggregates of the
    halo nodes, Galerkin products with exchanged halo rows, block Chebyshev smoothing -- instead of one hierarchy per
    shard.  Q1 elasticity through the host contract: the solution is the system's, the PCG count stays within 1.3x (+2)
    of the oracle's single-device block-3 count and below the per-shard (additive Schwarz) count."""
    A = oracle.elasticity_q1(M)
    Msp = A.to_scipy().tocsc()
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, A.n)
    cfg = dict(coarse_enough=100, ncycle=1, cheb_degree=3, cheb_power_iters=20)
    ref = oracle.AMG(A, block_size=3, **cfg)
    xo, ito, _ = oracle.cg_amgcl(A, b, precond=ref, tol=1e-9, max_iter=500)
    res = {}
    for mode in (2, 0):
        s = S.create({"solver": "HIP", "HIP": {"devices": devices, "precond": "amg", "block_size": 3, "tolerance": 1e-9,
                                               "amg": dict(cfg, aggregation_min_rows=0, dist_global=mode,
                                                           dist_replicate_rows=repl)}})
        s.analyze_pattern(Msp, A.n)
        s.factorize(Msp)
        x = np.zeros(A.n)
        s.solve(b, x)
        res[mode] = (x, s.get_info(), s.get_param("amg.distributed_levels"))
        assert np.linalg.norm(Msp @ x - b) < 1.5e-9 * np.linalg.norm(b)
    x2, i2, dl = res[2]
    assert dl >= 1 and i2["amg_levels"] >= 2
    assert np.abs(x2 - xo).max() <= 1e-6 * np.abs(xo).max()
    assert i2["num_iterations"] <= 1.3 * ito + 2, (i2["num_iterations"], ito)
    assert i2["num_iterations"] <= res[0][1]["num_iterations"]


@pytest.mark.parametrize("seed,n,devices,bs", [(1, 3000, [0, 0], 1), (2, 5000, [0, 0, 0, 0, 0], 1), (3, 2400, [0, 0, 0], 3),
                                                (4, 4000, [0, 0, 0], 1), (5, 1800, [0, 0], 2)])
def test_distributed_hierarchy_on_irregular_graphs(S, oracle, seed, n, devices, bs):
    """The hierarchy built on the shards (amg.dist_global 2) on matrices that are no grids: random sparse SPD M-matrices
    (graph Laplacians with a shift; seed 4: long-range couplings, so that every shard talks to every other one and halo
    rows come from several owners), scalar and with bs x bs node blocks, cut by nonzeros at uneven places.  Against the
    single-device solve of the same backend: same solution, iteration count within 1.3x (+3)."""
    rng = np.random.default_rng(seed)
    nn = n // bs
    if seed == 4:
        i, j = rng.integers(0, nn, 6 * nn), rng.integers(0, nn, 6 * nn)
    else:  # banded randomness: neighbours within +-40 nodes
        i = rng.integers(0, nn, 6 * nn)
        j = np.clip(i + rng.integers(-40, 41, 6 * nn), 0, nn - 1)
    keep = i != j
    G = sp.coo_matrix((rng.uniform(0.2, 1.0, keep.sum()), (i[keep], j[keep])), shape=(nn, nn)).tocsr()
    G = G + G.T
    Lap = sp.diags(np.asarray(G.sum(axis=1)).ravel()) - G + 0.05 * sp.identity(nn)
    if bs > 1:  # node blocks: an SPD bs x bs coupling per edge
        B = rng.uniform(-0.3, 0.3, (bs, bs))
        B = B @ B.T + np.eye(bs)
        M = sp.kron(Lap, B).tocsr()
    else:
        M = Lap.tocsr()
    M.sort_indices()
    M = M.tocsc()
    b = rng.uniform(-1, 1, M.shape[0])
    amg = dict(coarse_enough=40, ncycle=1, cheb_degree=3, cheb_power_iters=20, aggregation_min_rows=0,
               dist_replicate_rows=30)
    res = {}
    for name, dev in (("one", [0]), ("shards", devices)):
        s = S.create({"solver": "HIP", "HIP": {"devices": dev, "precond": "amg", "block_size": bs, "tolerance": 1e-9,
                                               "max_iter": 3000, "amg": amg}})
        s.analyze_pattern(M, M.shape[0])
        s.factorize(M)
        x = np.zeros(M.shape[0])
        s.solve(b, x)
        res[name] = (x, s.get_info(), s.get_param("amg.distributed_levels"))
        assert np.linalg.norm(M @ x - b) < 1.5e-9 * np.linalg.norm(b), name
    (x1, i1, _), (xs, is_, dl) = res["one"], res["shards"]
    assert dl >= 1
    assert np.abs(xs - x1).max() <= 1e-6 * np.abs(x1).max()
    assert is_["num_iterations"] <= 1.3 * i1["num_iterations"] + 3, (is_["num_iterations"], i1["num_iterations"])
    # Newton's case on shards: new values on the same pattern -> the distributed hierarchy keeps its aggregates, patterns,
    # halo links and row-exchange plans and recomputes the numbers; it must then act like one built from scratch
    M2 = M.copy()
    M2.data = M2.data * rng.uniform(0.9, 1.1, M2.nnz)
    M2 = ((M2 + M2.T) * 0.5 + 0.1 * sp.identity(M.shape[0])).tocsc()  # symmetric again, same pattern, still SPD
    M2.sort_indices()
    assert np.array_equal(M2.indptr, M.indptr) and np.array_equal(M2.indices, M.indices)
    s.factorize(M2)
    assert s.get_param("amg.last_setup_reused") == 1 and s.get_param("stats.amg_refreshes") >= 1
    xr = np.zeros(M.shape[0])
    s.solve(b, xr)
    ir = s.get_info()
    fresh = S.create({"solver": "HIP", "HIP": {"devices": devices, "precond": "amg", "block_size": bs, "tolerance": 1e-9,
                                               "max_iter": 3000, "amg": amg}})
    fresh.factorize(M2)
    xf = np.zeros(M.shape[0])
    fresh.solve(b, xf)
    assert abs(ir["num_iterations"] - fresh.get_info()["num_iterations"]) <= 1
    assert np.abs(xr - xf).max() <= 1e-7 * np.abs(xf).max()
    assert np.linalg.norm(M2 @ xr - b) < 1.5e-9 * np.linalg.norm(b)


@pytest.mark.parametrize("devices,precond", [([0, 0, 0], "jacobi"), ([0, 0], "amg"), ([0, 0, 0, 0], "jacobi")])
def test_scattered_numbering_is_renumbered_before_it_is_partitioned(S, oracle, devices, precond):
    """"reorder" on several devices: the multi-device handle searches the order of the WHOLE pattern (on device 0),
    partitions the renumbered rows -- contiguous ranges of a breadth-first order are slabs of the mesh --, every shard packs
    its rows and renames / sorts the columns on its device.  The caller's numbering outside; the halo of a shard drops
    from almost the whole vector to two frontiers of the search; the solution and the iteration count are the oracle's."""
    A0 = oracle.poisson7(20, 18, 16)
    rng = np.random.default_rng(5)
    A = oracle.permuted(A0, rng.permutation(A0.n).astype(np.int32))
    M = A.to_scipy().tocsc()
    xs = oracle.splitmix_vector(A.n, 42)
    b = oracle.spmv(A, xs)
    hip = {"devices": devices, "tolerance": 1e-9, "reorder": 1, "precond": precond,
           "amg": {"coarse_enough": 200, "cheb_degree": 3, "cheb_power_iters": 20, "aggregation_min_rows": 0, "dist_replicate_rows": 300}}
    s = S.create({"solver": "HIP", "HIP": hip})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    perm, active = s.reorder_perm()
    order, oinfo = oracle.cuthill_mckee(A, reverse=True)
    assert active and s.get_param("reorder.active") == 1 and np.array_equal(order[perm], np.arange(A.n))  # the oracle's order
    assert s.get_param("reorder.levels") == oinfo["levels"]
    halo_new = s.get_param("dist.n_halo")
    x = np.zeros(A.n)
    s.solve(b, x)
    info = s.get_info()
    assert info["true_residual"] < 1.5e-9 and np.linalg.norm(M @ x - b) < 1.5e-9 * np.linalg.norm(b)
    if precond == "jacobi":
        xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-9)
        assert abs(info["solver_iter"] - ito) <= 2 and np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    else:
        assert info["amg_levels"] >= 2 and info["num_iterations"] < 40
        assert np.abs(x - xs).max() <= 1e-6 * np.abs(xs).max()
    # the shards are balanced by the renumbered rows' entries and cover them
    rows = [s.shard_rows(r) for r in range(len(devices))]
    assert rows[0][0] == 0 and rows[-1][1] == A.n and all(rows[r][1] == rows[r + 1][0] for r in range(len(devices) - 1))
    # same pattern, new values: the order is kept (no new search), the answer scales
    assert s.get_param("stats.reorder_searches") == 1
    s.factorize((M * 2.0).tocsc())
    x2 = np.zeros(A.n)
    s.solve(b, x2)
    assert np.abs(2 * x2 - x).max() <= 1e-6 * np.abs(x).max() and s.get_param("stats.reorder_searches") == 1
    # the caller's numbering: every shard needs almost the whole vector
    s.set_parameters({"HIP": {"reorder": 0}})
    s.factorize(M)
    assert s.reorder_perm() == (None, False)
    halo_old = s.get_param("dist.n_halo")
    assert halo_new < 0.35 * halo_old and halo_new <= 2.5 * (20 * 18 + 18 * 16 + 20 * 16)
    x3 = np.zeros(A.n)
    s.solve(b, x3)
    assert np.linalg.norm(M @ x3 - b) < 1.5e-9 * np.linalg.norm(b)


def test_auto_renumbering_on_shards_and_block3_nodes(S, oracle):
    """auto (the default) on a multi-device handle: a grid keeps its numbering; a scattered block-3 system is renumbered on
    its node graph, whole nodes move, the partition still cuts between nodes."""
    E = oracle.elasticity_q1(16)  # (4096 nodes: a group of 64 rows must not reach most of the mesh for the figure to mean anything)
    nb = E.n // 3
    pn = np.random.default_rng(2).permutation(nb)
    dof = (3 * pn[:, None] + np.arange(3)[None, :]).ravel().astype(np.int32)
    A = oracle.permuted(E, dof)
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    s = S.create({"solver": "HIP", "HIP": {"devices": [0, 0, 0], "tolerance": 1e-9, "block_size": 3, "reorder_min_rows": 0,
                                           "max_iter": 5000}})
    s.analyze_pattern(M, A.n)
    s.factorize(M)
    perm, active = s.reorder_perm()
    assert active and s.get_param("reorder.spread_before") > 2.5 > s.get_param("reorder.spread_after")
    assert np.array_equal(perm[1::3], perm[0::3] + 1) and np.array_equal(perm[2::3], perm[0::3] + 2) and not (perm[0::3] % 3).any()
    assert all(s.shard_rows(r)[0] % 3 == 0 for r in range(3))
    x = np.zeros(A.n)
    s.solve(b, x)
    xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-9, max_iter=5000)
    assert abs(s.get_info()["solver_iter"] - ito) <= 3 and np.abs(x - xo).max() <= 1e-6 * np.abs(xo).max()
    G = oracle.poisson7(16, 16, 12)
    s2 = S.create({"solver": "HIP", "HIP": {"devices": [0, 0], "reorder_min_rows": 0}})
    s2.analyze_pattern(G.to_scipy().tocsc(), G.n)
    s2.factorize(G.to_scipy().tocsc())
    assert s2.get_param("reorder.active") == 0 and s2.get_param("reorder.spread_before") < 2.0


@pytest.mark.parametrize("kind,devices", [("laplace", [0, 0, 0]), ("elasticity", [0, 0])])
def test_unstructured_mesh_on_shards_newton_flow(S, oracle, kind, devices):
    """P1 Laplace / elasticity on Delaunay tetrahedra (tests/mesh_utils.py), nodes in a random order, through the
    multi-device handle: renumbered before partitioned, the AMG hierarchy built on the shards (block 3 for elasticity),
    three factorizations of the same pattern (Newton.cpp:189-193) -- the order is kept, the hierarchy refreshed -- and
    every solve meets the reference tests' inequality; a shard's halo stays a small fraction of its rows."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mesh_utils as mu
    b3 = 3 if kind == "elasticity" else 1
    P, T, bd = mu.tet_mesh(17, seed=11)
    K = mu.p1_laplace(P, T, bd) if b3 == 1 else mu.p1_elasticity(P, T, bd)
    K, _ = mu.renumber_nodes(K, b3, seed=12)
    n = K.shape[0]
    s = S.create({"solver": "HIP", "HIP": {"devices": devices, "precond": "amg", "block_size": b3, "tolerance": 1e-9,
                                           "reorder_min_rows": 0, "max_iter": 2000,
                                           "amg": {"coarse_enough": 300, "cheb_degree": 3, "cheb_power_iters": 20,
                                                   "aggregation_min_rows": 0, "dist_replicate_rows": 400}}})
    rng = np.random.default_rng(3)
    perm0 = None
    for k in range(3):
        H = (K + (0.1 * k) * sp.diags(K.diagonal())).tocsc()
        g = rng.uniform(-1, 1, n)
        s.analyze_pattern(H, n)
        s.factorize(H)
        perm, active = s.reorder_perm()
        assert active and s.get_param("reorder.active") == 1
        if perm0 is None:
            perm0 = perm
        else:
            assert np.array_equal(perm, perm0)
        x = np.zeros(n)
        s.solve(g, x)
        info = s.get_info()
        assert np.linalg.norm(H @ x - g) < 1e-7 * np.linalg.norm(g) and info["amg_levels"] >= 2 and info["num_iterations"] < 80
    assert s.get_param("dist.n_halo") < 0.5 * n / len(devices)


def _peer_vs_rccl(S, oracle, devices, grid, precond, single):
    A = oracle.poisson7(*grid)
    M = A.to_scipy().tocsc()
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    res = {}
    for mode in (0, 1):
        hip = {"devices": devices, "tolerance": 1e-9, "dist_collectives": mode, "dist_single_reduction": bool(single)}
        if precond == "amg":
            hip.update(precond="amg", amg={"coarse_enough": 200, "cheb_degree": 3, "cheb_power_iters": 20, "aggregation_min_rows": 0,
                                           "dist_replicate_rows": 300})
        s = S.create({"solver": "HIP", "HIP": hip})
        assert s.get_param("dist.peer_available") == 1
        s.analyze_pattern(M, A.n)
        s.factorize(M)
        assert s.get_param("dist.peer_in_use") == mode
        x = np.zeros(A.n)
        s.solve(b, x)
        i = s.get_info()
        x2 = np.zeros(A.n)
        s.factorize(M)  # the plan is prepared again, the epochs go on
        s.solve(b, x2)
        assert np.array_equal(x, x2)
        res[mode] = (x, i["num_iterations"], i["true_residual"])
    return A, b, res


@pytest.mark.parametrize("devices,grid,precond,single", [([0, 0], (12, 10, 16), "jacobi", 1), ([0, 0, 0], (9, 8, 21), "jacobi", 0),
                                                         ([0, 0, 0, 0], (16, 16, 24), "jacobi", 1), ([0, 0, 0], (20, 18, 24), "amg", 0)])
def test_peer_mapped_collectives_rehearsed_on_one_device(S, oracle, devices, grid, precond, single):
    """"dist_collectives" 1: the per-iteration all-reduce of the CG scalars and the halo exchange by stores into the peers'
    memory + epoch flags (dist_peer.hip) -- here with same-device "peers" (the same kernels, host-synchronised between
    posting and collecting).  Contributions are added in rank order, the halo entries are copies: the iterates are the
    RCCL-shaped (loopback) path's bit for bit, and the oracle's counts."""
    A, b, res = _peer_vs_rccl(S, oracle, devices, grid, precond, single)
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
    assert res[1][2] < 1.5e-9
    if precond == "jacobi":
        xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-9)
        assert abs(res[1][1] - ito) <= 2 and np.abs(res[1][0] - xo).max() <= 1e-6 * np.abs(xo).max()


def test_peer_halo_keeps_to_rccl_on_a_one_way_neighbour(S, oracle):
    """The staging parities of the peer-mapped halo exchange are safe between mutual neighbours only (dist_peer.hip,
    peer_prepare_halo): a stored entry that makes shard 0 read from shard 2 without shard 2 reading from shard 0 -- here
    an explicit zero at (0, n-1) -- sends every rank to the RCCL-shaped exchange ("dist.peer_in_use" 0 on all of them, not a
    mix), with the same iterates as without the entry."""
    A = oracle.poisson7(9, 8, 21)
    b = oracle.spmv(A, oracle.splitmix_vector(A.n, 42))
    sols = []
    for one_way in (False, True):
        H = A.to_scipy().tocoo()
        if one_way:
            H = sp.coo_matrix((np.concatenate([H.data, [0.0]]), (np.concatenate([H.row, [0]]), np.concatenate([H.col, [A.n - 1]]))),
                              shape=H.shape)
        H = H.tocsc()
        assert H.nnz == A.to_scipy().nnz + int(one_way)
        s = S.create({"solver": "HIP", "HIP": {"devices": [0, 0, 0], "tolerance": 1e-9, "dist_collectives": 1}})
        s.analyze_pattern(H, A.n)
        s.factorize(H)
        assert s.get_param("dist.peer_available") == 1 and s.get_param("dist.peer_in_use") == (0 if one_way else 1)
        x = np.zeros(A.n)
        s.solve(b, x)
        sols.append((x, s.get_info()["num_iterations"]))
    assert sols[0][1] == sols[1][1] and np.abs(sols[0][0] - sols[1][0]).max() <= 1e-12 * np.abs(sols[0][0]).max()


@pytest.mark.rccl_multi  # (counted by conftest.py: "rccl_world_gt1_tests_executed")
@pytest.mark.skipif("_device_count() < 2")
def test_peer_mapped_collectives_on_distinct_devices(S, oracle):
    """The same over real peers (xGMI peer mapping, no host synchronisation inside a solve): the RCCL path's iterates."""
    nd = min(_device_count(), 8)
    A, b, res = _peer_vs_rccl(S, oracle, list(range(nd)), (32, 32, 16 * nd), "jacobi", 1)
    assert res[0][1] == res[1][1] and np.abs(res[0][0] - res[1][0]).max() <= 1e-12 * np.abs(res[0][0]).max()
    xo, ito, _ = oracle.cg_eigen(A, b, tol=1e-9)
    assert abs(res[1][1] - ito) <= 2 and np.abs(res[1][0] - xo).max() <= 1e-6 * np.abs(xo).max()
