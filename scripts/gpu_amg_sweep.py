"""AMG-PCG on the 256^3 (or BIGN^3) Poisson system: setup time, solve time, iterations per configuration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver

N = int(os.environ.get("BIGN", "256"))
cfgs = [dict(ncycle=1, cheb_degree=2), dict(ncycle=1, cheb_degree=3), dict(ncycle=1, cheb_degree=4),
        dict(ncycle=1, cheb_degree=6), dict(ncycle=1, cheb_degree=3, cheb_lower=1/30), dict(ncycle=1, cheb_degree=2, cheb_lower=1/10),
        dict(ncycle=1, cheb_degree=3, cheb_lower=1/10), dict(ncycle=2, cheb_degree=16)]
for cfg in cfgs:
    s = HIPSolver("")
    amg = dict(cfg, cheb_power_iters=20)
    s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "max_iter": 500, "amg": amg}})
    t = time.time(); s.generate_poisson7(N); tf = time.time() - t
    n, nnz, _ = s.matrix_shape()
    b = s.device_array(n); x = s.to_device(np.zeros(n))
    s.generate_rhs(42, b)
    s.solve_device(b, x)           # warm-up
    x.upload(np.zeros(n))
    t = time.time(); s.solve_device(b, x); ts = time.time() - t
    i = s.get_info()
    lv = [s.amg_level_info(l) for l in range(i["amg_levels"])]
    print(f"N={N} {cfg}: setup {tf:.2f}s solve {ts*1e3:.1f} ms iters={i['num_iterations']} true={i['true_residual']:.2e} "
          f"DOF/s={n/ts:.3e} levels={[(r, z) for r, z, _ in lv]}", flush=True)
    del s
