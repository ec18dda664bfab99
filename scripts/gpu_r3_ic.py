"""precond = ic on the 7-point Poisson system: host factorization time, apply time, PCG time against Jacobi."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
for N in (64, 128, 192):
    for pre in ("", "ic"):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 5000, "precond": pre or "jacobi"}})
        t = time.perf_counter(); s.generate_poisson7(N); s.synchronize(); tf = time.perf_counter() - t
        n = s.matrix_shape()[0]
        b, x, z = s.device_array(n), s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        s.precond_apply_device(b, z); s.synchronize()
        t = time.perf_counter()
        for _ in range(5): s.precond_apply_device(b, z)
        s.synchronize(); ta = (time.perf_counter() - t) / 5
        best = 1e9
        for _ in range(2):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        print(f"N={N} precond={pre or 'jacobi':6s} factorize {tf:.2f} s apply {ta*1e3:.3f} ms solve {best*1e3:.1f} ms its={i['num_iterations']} "
              f"res={i['true_residual']:.1e} levels={int(s.get_param('ic.levels'))}", flush=True)
        del s
