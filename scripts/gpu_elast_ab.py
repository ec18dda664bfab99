import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
base = dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))
for name, extra in [("auto", {}), ("pipe only", dict(spmv_kernel=0)), ("dma only", dict(spmv_kernel=1)), ("auto 5/cu", dict(spmv_blocks_per_cu=5)), ("pipe 5/cu", dict(spmv_kernel=0, spmv_blocks_per_cu=5))]:
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(base, tolerance=1e-8, max_iter=20000, **extra)})
    s.generate_elasticity_q1(M)
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); best = min(best, time.time() - t)
    i = s.get_info()
    lv = [s.amg_level_info(l)[:2] for l in range(i["amg_levels"])]
    print(f"M={M} {name:12s} solve {best*1e3:8.1f} ms its={i['num_iterations']:4d} levels={lv}", flush=True)
    del s
