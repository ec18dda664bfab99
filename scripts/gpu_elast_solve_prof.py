"""Q1 elasticity M = 100 block-3 AMG-PCG: one setup, three solves (for rocprofv3 per-iteration breakdowns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3))})  # bench.py AMG_RECOMMENDED
s.generate_elasticity_q1(M); s.synchronize()
n = s.matrix_shape()[0]
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
for _ in range(3):
    s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
    t = time.time(); s.solve_device(b, x); dt = time.time() - t
print(f"solve {dt*1e3:.1f} ms its={s.get_info()['num_iterations']}")
