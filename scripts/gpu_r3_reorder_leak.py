"""Device-memory stability of "reorder" over repeated factorize / solve with changing sizes, numberings and preconditioners;
and the resident bytes of a renumbered handle against the caller's-numbering one (one copy of the matrix, not two)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from polysolve_amd import HIPSolver
def used():
    f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
s = HIPSolver("")
for it in range(24):
    N = 96 if it % 3 else 128
    pre = "amg" if it % 2 else "jacobi"
    s.set_parameters({"HIP": dict(precond=pre, tolerance=1e-8, reorder=(it % 3), amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, reuse=bool(it % 4)))})
    s.generate_poisson7_permuted(N, N, N, mode=1 + (it % 2), seed=it % 5); n = N ** 3
    b, x = s.device_array(n), s.device_array(n); s.generate_rhs(42, b); s.axpby_device(n, 0.0, b, 0.0, x)
    s.solve_device(b, x); s.synchronize(); assert s.get_info()["true_residual"] < 1.5e-8; b.free(); x.free()
    if it % 4 == 3: print(it, round(used()), "MiB  handle:", round(s.get_param("stats.device_bytes") / 2**20), "MiB", flush=True)
for reorder in (0, 1):
    t = HIPSolver("")
    t.set_parameters({"HIP": dict(reorder=reorder)})
    t.generate_poisson7_permuted(256, 256, 256, mode=1, seed=7)
    print("256^3 random numbering, reorder", reorder, "resident", round(t.get_param("stats.device_bytes") / 2**20), "MiB, peak", round(t.get_param("stats.device_bytes_peak") / 2**20), "MiB")
    del t
