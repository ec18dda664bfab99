"""AMG-PCG through the shards' code path with a one-rank RCCL communicator (global hierarchy, level 0 'distributed'):
what the distributed cycle costs without the wire, next to the single-GPU path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "216"))
for dist in (0, 1):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))})
    if dist:
        s.comm_init(0, 1, HIPSolver.comm_unique_id())
    t = time.time(); s.generate_poisson7(N); s.synchronize(); tf = time.time() - t
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    print(f"N={N} dist={dist}: setup {tf:.3f} s, solve {best*1e3:.1f} ms, {i['num_iterations']} iterations, true {i['true_residual']:.2e}", flush=True)
    del s
