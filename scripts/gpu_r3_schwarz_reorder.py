"""Does a breadth-first numbering give the 64-unknown Schwarz domains what the grid's x-lines do not?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for precond, levels in (("jacobi", 1), ("schwarz", 1), ("schwarz", 2), ("schwarz", 3)):
    for reorder in (0, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 5000, "precond": precond, "reorder": reorder, "schwarz": {"levels": levels}}})
        s.generate_poisson7(N)
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        for _ in range(2):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); dt = time.perf_counter() - t
        i = s.get_info()
        print(precond, levels, "reorder", reorder, "iters", i["num_iterations"], f"solve {dt*1e3:.1f} ms", flush=True)
        del s
