"""2.1 M-row shard (256 x 256 x 32): cache policies of the matrix stream and of the vector kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
nz = int(os.environ.get("NZ", "32"))
for dist in (0, 1):
    for name, prm in (("default", {}), ("matrix nt, vectors plain", dict(spmv_nt=1, vec_policy=0)),
                      ("matrix nt, vector loads nt", dict(spmv_nt=1, vec_policy=1)), ("all nt", dict(spmv_nt=1, vec_policy=7)),
                      ("dma kernel (no dictionary)", dict(spmv_kernel=1)), ("pipe kernel", dict(spmv_kernel=0))):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=300)})
        if dist:
            s.comm_init(0, 1, HIPSolver.comm_unique_id())
        s.generate_poisson7(256, 256, nz)
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e9
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        print(f"nz={nz} dist={dist} {name:28s}: {best*1e6/i['num_iterations']:7.1f} us per iteration", flush=True)
        del s
