import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = 256
for pol in (15, 7, 13, 5, 11, 3, 14, 6, 9, 1, 0):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, profile_spmv=8, vec_policy=pol)})
    s.generate_poisson7(N)
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.info_struct()
    print(f"vec_policy={pol:2d} (loads nt={pol&1} r={pol>>1&1} x={pol>>2&1} p={pol>>3&1}): {best*1e3:7.2f} ms {best*1e3/i.num_iterations:.4f} ms/it spmv {i.spmv_ms_avg:.4f}", flush=True)
