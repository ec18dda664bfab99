import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "216"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, reuse=False, aggregation_rounds=bool(int(os.environ.get("ROUNDS", "0")))))})
s.generate_poisson7(N); s.synchronize()
os.environ["PSOLVE_TIMING"] = "1"
t = time.time(); s.generate_poisson7(N); s.synchronize(); print(f"TOTAL second setup {time.time()-t:.4f} s", flush=True)
