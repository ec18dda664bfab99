"""Block-3 elasticity through the multi-device handle (loopback shards on ONE GPU): hierarchy on the shards (2) vs per
shard (0) vs one device.  Counts and bytes are meaningful, timings are not multi-GPU timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import Solver
M = int(os.environ.get("M", "48"))
A = O.elasticity_q1(M)
Msp = A.to_scipy().tocsc()
b = np.random.default_rng(3).uniform(-1, 1, A.n)
amg = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)
for devices, mode in (([0], 2), ([0, 0, 0, 0], 2), ([0, 0, 0, 0], 0), ([0] * 8, 2), ([0] * 8, 0)):
    s = Solver.create({"solver": "HIP", "HIP": {"devices": devices, "precond": "amg", "block_size": 3, "tolerance": 1e-8,
                                                 "amg": dict(amg, dist_global=mode)}})
    s.analyze_pattern(Msp, A.n)
    t = time.perf_counter(); s.factorize(Msp); tf = time.perf_counter() - t
    x = np.zeros(A.n)
    t = time.perf_counter(); s.solve(b, x); ts = time.perf_counter() - t
    i = s.get_info()
    print(f"M={M} ({A.n} dof) shards={len(devices)} dist_global={mode}: its={i['num_iterations']} levels={i['amg_levels']} "
          f"distributed_levels={int(s.get_param('amg.distributed_levels'))} res={np.linalg.norm(Msp@x-b)/np.linalg.norm(b):.2e} "
          f"device MiB/shard={s.get_param('stats.device_bytes')/2**20:.0f} factorize {tf:.2f} s solve {ts:.2f} s", flush=True)
    del s
