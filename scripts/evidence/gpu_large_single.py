"""384^3 and 512^3 Jacobi-PCG on ONE GPU (vectors beyond the Infinity Cache)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
for N in (384, 512):
    for k in (-1, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, spmv_kernel=k)})
        s.generate_poisson7(N)
        n = N ** 3
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); dt = time.time() - t
        i = s.get_info()
        print(f"{N}^3 Jacobi-PCG spmv_kernel={k:2d} patterns={int(s.get_param('spmv_patterns'))}: {dt:.3f} s, {i['num_iterations']} iterations, "
              f"{dt*1e3/i['num_iterations']:.3f} ms/it, {n/dt/1e6:.1f} M DOF/s, true {i['true_residual']:.2e}", flush=True)
        b.free(); x.free()
        del s
