"""One AMG-PCG solve (for rocprofv3): env NS_N, XCD_MAP, CHEB (degree), NCYCLE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("NS_N", "216"))
s = HIPSolver("")
deg = int(os.environ.get("CHEB", "2"))
amg = dict(ncycle=int(os.environ.get("NCYCLE", "1")), cheb_degree=deg, cheb_power_iters=20)
if deg <= 4: amg["cheb_lower"] = 0.1
s.set_parameters({"HIP": dict(spmv_xcd_map=int(os.environ.get("XCD_MAP", "2")), precond="amg", tolerance=1e-8, max_iter=2000, amg=amg)})
s.generate_poisson7(N)
n, nnz, _ = s.matrix_shape()
b, x = s.device_array(n), s.to_device(np.zeros(n))
s.generate_rhs(42, b)
for rep in range(3):
    x.upload(np.zeros(n))
    t = time.time(); s.solve_device(b, x); dt = time.time() - t
print(f"N={N} map={os.environ.get('XCD_MAP','2')}: {dt*1e3:.1f} ms its={s.get_info()['num_iterations']}")
