"""One first factorize + 6 numeric refreshes of the Q1 elasticity M = 100 block-3 hierarchy (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))})
s.generate_elasticity_q1(M); s.synchronize()
for k in range(6):
    t = time.time(); s.generate_elasticity_q1(M); s.synchronize()
    print(f"refactorize {k}: {time.time()-t:.4f} s reused={s.get_param('amg.last_setup_reused')}", flush=True)
