"""Numeric refresh (same pattern, new values: what Newton does every iteration) against the first factorize."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "256"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))})
t = time.time(); s.generate_poisson7(N); s.synchronize(); print(f"first factorize (incl. generator) {time.time()-t:.4f} s", flush=True)
for k in range(3):
    if k == 2:
        os.environ["PSOLVE_TIMING"] = "1"
    t = time.time(); s.generate_poisson7(N); s.synchronize()
    print(f"refactorize {k}: {time.time()-t:.4f} s reused={s.get_param('amg.last_setup_reused')}", flush=True)
