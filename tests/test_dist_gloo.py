"""Multi-process CPU cover of the N>1 path (gloo, 127.0.0.1): partition -> halo plan -> request
exchange -> column remap -> distributed PCG, against the global oracle solve."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, script, args):
    pytest.importorskip("torch")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)]
                                      + [str(g) for g in args], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
    return outs[0]


@pytest.mark.parametrize("world,grid", [(2, (6, 5, 8)), (3, (4, 4, 9))])
def test_row_partitioned_pcg_gloo(world, grid):
    assert "DIST_OK" in _launch(world, "dist_worker.py", grid)


@pytest.mark.parametrize("world,args", [(2, ("elasticity", 7)), (2, ("poisson", 7, 6, 10)), (3, ("elasticity", 8)), (3, ("poisson", 5, 5, 11))])
def test_sharded_amg_level_and_block3_pcg_gloo(world, args):
    """Round 6 (the only CPU cover of N > 1 used to be Jacobi on a scalar grid): a block-3 operator row-partitioned by node
    planes -- distributed Jacobi-PCG against the oracle -- and level 1 of a sharded smoothed-aggregation hierarchy built rank
    by rank as DistAmg::setup does (amg_dist.hpp), with the halo plans of level 0, of level 1 (a non-slab partition of the
    aggregates) and of P's columns all made by the product's planner, against the same construction done globally."""
    out = _launch(world, "dist_worker_amg.py", args)
    assert "DIST_AMG_OK" in out and f"bs={3 if args[0] == 'elasticity' else 1}" in out
