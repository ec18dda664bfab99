"""A/B of "amg.coarse_dense" (the relaxed coarsest level applied as one dense operator): reference configuration (W-cycle,
Chebyshev-16, 100 power iterations) and recommended configuration at N^3, interleaved solves.  env N (216), REPS (3)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
N = int(os.environ.get("N", "216")); REPS = int(os.environ.get("REPS", "3"))
REF = dict(ncycle=2, cheb_degree=16, cheb_power_iters=100)
out = {}
for name, amg in (("reference", REF), ("recommended", dict(AMG_RECOMMENDED))):
    for cd in (0, 1024):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "amg": dict(amg, coarse_dense=cd)}})
        s.generate_poisson7(N); s.synchronize()
        t = time.perf_counter(); s.generate_poisson7(N); s.synchronize(); t_refresh = time.perf_counter() - t
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        ts = []
        for _ in range(REPS + 1):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); ts.append(time.perf_counter() - t)
        i = s.get_info()
        out[f"{name}_coarse_dense_{cd}"] = dict(solve_ms=[round(v * 1e3, 2) for v in ts[1:]], iterations=i["num_iterations"],
                                                true_residual=i["true_residual"], refresh_ms=round(t_refresh * 1e3, 2),
                                                coarsest_rows=s.amg_level_info(int(i["amg_levels"]) - 1)[0])
        del s
print(json.dumps(out))
