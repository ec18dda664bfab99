"""16-bit columns on the cycle's CSR operators ("spmv_col16") with the row kinds at level 0: 216^3 AMG-PCG, both configurations"""
import json, sys, time
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
REF = dict(ncycle=2, cheb_degree=16, cheb_lower=0.008333333333, cheb_higher=2.0, cheb_power_iters=100, sa_relax=1.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 216
for name, amg in (("recommended", AMG_RECOMMENDED), ("reference", REF)):
    for c16 in (False, True, False, True):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"precond": "amg", "tolerance": 1e-8, "max_iter": 2000, "spmv_col16": c16, "amg": dict(amg)}})
        s.generate_poisson7(N); s.synchronize()
        t = time.perf_counter(); s.generate_poisson7(N); s.synchronize(); t_refresh = time.perf_counter() - t
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e9
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        print(json.dumps({"N": N, "config": name, "col16": c16, "refresh_s": t_refresh, "solve_s": best, "iters": int(i["num_iterations"]),
                          "res": i["true_residual"]}), flush=True)
