import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("BIGN", "64"))
s = HIPSolver("")
s.set_parameters({"HIP": {"use_graph": int(os.environ.get("GRAPH", "1"))}})
s.generate_poisson7(N)
n, nnz, _ = s.matrix_shape()
b, x = s.device_array(n), s.to_device(np.zeros(n))
s.generate_rhs(42, b)
for _ in range(3):
    x.upload(np.zeros(n)); t = time.time(); s.solve_device(b, x); dt = time.time() - t
i = s.get_info()
print(f"N={N} graph={os.environ.get('GRAPH','1')} solve {dt*1e3:.2f} ms iters={i['num_iterations']} {dt*1e6/i['num_iterations']:.1f} us/iter grid={s.get_param('grid')} spmv_grid={s.get_param('spmv_grid')}")
