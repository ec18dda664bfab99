import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
for N in (64, 128):
    for name, hip in [("jacobi", {})] + [(f"schwarz L{l}", {"precond": "schwarz", "schwarz": {"levels": l}}) for l in (1, 2, 3, 4)]:
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(hip, tolerance=1e-8)})
        s.generate_poisson7(N)
        n = N ** 3
        b, x = s.device_array(n), s.to_device(np.zeros(n))
        s.generate_rhs(42, b)
        t = time.time(); s.solve_device(b, x); dt = time.time() - t
        i = s.get_info()
        print(f"poisson {N}^3 {name:12s} its={i['num_iterations']:4d} {dt*1e3:7.1f} ms", flush=True)
