import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polysolve_amd import HIPSolver
def used(): 
    f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))})
for it in range(24):
    N = 96 if it % 3 else 128
    s.set_parameters({"HIP": dict(amg=dict(reuse=bool(it % 2)))})
    s.generate_poisson7(N); n = N ** 3
    b, x = s.device_array(n), s.device_array(n); s.generate_rhs(42, b); s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x); s.synchronize(); b.free(); x.free()
    if it % 4 == 3: print(it, round(used()), "MiB", flush=True)
