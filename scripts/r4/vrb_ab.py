"""Row-blocks packed to the LDS tile ("lab.var_row_blocks") against fixed-height row-blocks, same process: N^3 Poisson
AMG-PCG first setup, numeric refresh and solve; the iterates are bit-equal."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
N = int(os.environ.get("N", "256"))
xs = []
for rep in range(2):
    for flag in (0, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(AMG_RECOMMENDED), **{"lab.var_row_blocks": flag})})
        t = time.time(); s.generate_poisson7(N, N, N); s.synchronize(); t_setup = time.time() - t
        t = time.time(); s.generate_poisson7(N, N, N); s.synchronize(); t_refresh = time.time() - t
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        ts = []
        for _ in range(5):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.time(); s.solve_device(b, x); s.synchronize(); ts.append(time.time() - t)
        xs.append(x.download())
        print(json.dumps(dict(N=N, packed=flag, ops=s.get_param("amg.packed_row_block_operators"), setup_s=round(t_setup, 4), refresh_s=round(t_refresh, 4),
                              solve_ms=round(min(ts) * 1e3, 2), its=s.get_info()["num_iterations"], reused=s.get_param("amg.last_setup_reused"))))
        del s, b, x
print("bit-equal", all(np.array_equal(xs[0], v) for v in xs[1:]),
      {f"{i}{j}": (bool(np.array_equal(xs[i], xs[j])), float(np.max(np.abs(xs[i] - xs[j])))) for i in range(4) for j in range(i + 1, 4)})
