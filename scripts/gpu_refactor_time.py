import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("BIGN", "256"))
s = HIPSolver("")
s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "amg": dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)}})
t = time.time(); s.generate_poisson7(N); print(f"first factorize (host hierarchy) {time.time()-t:.3f}s reused={s.get_param('amg.last_setup_reused')}")
for _ in range(2):
    t = time.time(); s.generate_poisson7(N); print(f"refactorize same pattern {time.time()-t:.3f}s reused={s.get_param('amg.last_setup_reused')}")
n, nnz, _ = s.matrix_shape()
b, x = s.device_array(n), s.to_device(np.zeros(n))
s.generate_rhs(42, b); s.solve_device(b, x); x.upload(np.zeros(n))
t = time.time(); s.solve_device(b, x); print(f"solve {1e3*(time.time()-t):.1f} ms iters={s.get_info()['num_iterations']} true={s.get_info()['true_residual']:.2e}")
