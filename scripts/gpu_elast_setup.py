"""configs[2]: Q1 elasticity M = 100, block-3 AMG: first factorize with the timing lines, refactorize, solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20,
                                                                     aggregation_rounds=bool(int(os.environ.get("ROUNDS", "0"))), aggregation_max_rounds=int(os.environ.get("MAXR", "10000"))))})
s.generate_elasticity_q1(M); s.synchronize()
s.set_parameters({"HIP": dict(amg=dict(reuse=False))})
os.environ["PSOLVE_TIMING"] = "1"
t = time.time(); s.generate_elasticity_q1(M); s.synchronize(); print(f"TOTAL full setup (incl. generator) {time.time()-t:.4f} s", flush=True)
del os.environ["PSOLVE_TIMING"]
s.set_parameters({"HIP": dict(amg=dict(reuse=True))})
s.generate_elasticity_q1(M); s.synchronize()
t = time.time(); s.generate_elasticity_q1(M); s.synchronize(); print(f"refactorize {time.time()-t:.4f} s reused={s.get_param('amg.last_setup_reused')}", flush=True)
n = s.matrix_shape()[0]
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
for _ in range(2):
    s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
    t = time.time(); s.solve_device(b, x); dt = time.time() - t
print(f"solve {dt*1e3:.1f} ms its={s.get_info()['num_iterations']}")
