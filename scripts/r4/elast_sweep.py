"""configs[2]: smoother-degree / interval sweep with the round-4 cycle (block operators, fused block Chebyshev step)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
base = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3)
cases = [dict(), dict(cheb_degree=1), dict(cheb_degree=1, cheb_lower=0.3), dict(cheb_degree=1, cheb_lower=0.5), dict(cheb_degree=1, cheb_lower=0.7),
         dict(cheb_degree=1, npre=2, npost=2), dict(cheb_degree=3, cheb_lower=0.05), dict(sa_relax=1.5), dict(sa_relax=1.0), dict(coarse_enough=1000), dict(max_levels=3)]
for c in cases:
    amg = dict(base, **c)
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, max_iter=400, amg=amg)})
    s.generate_elasticity_q1(M); s.synchronize()
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
    info = s.get_info()
    print(json.dumps(dict(case=c, solve_ms=round(best * 1e3, 1), its=info["num_iterations"], res=info["true_residual"], levels=info["amg_levels"])), flush=True)
    b.free(); x.free(); del s
