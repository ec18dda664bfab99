"""A/B of launch parameters on the AMG-PCG solve time (V-cycle Chebyshev-2 and the AMGCL-default W/16)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("NS_N", "216"))
variants = [dict(), dict(spmv_xcd_map=0), dict(spmv_blocks_per_cu=4), dict(spmv_chunk_rows=2048), dict(spmv_chunk_rows=32768),
            dict(blocks_per_cu=4), dict(spmv_xcd_map=0, spmv_blocks_per_cu=4)]
for amg in (dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20), dict(ncycle=2, cheb_degree=16, cheb_power_iters=100)):
    for v in variants:
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(v, precond="amg", tolerance=1e-8, max_iter=2000, amg=amg)})
        s.generate_poisson7(N)
        n, nnz, _ = s.matrix_shape()
        b, x = s.device_array(n), s.to_device(np.zeros(n))
        s.generate_rhs(42, b)
        s.solve_device(b, x)
        best = 1e9
        for rep in range(3):
            x.upload(np.zeros(n))
            t = time.time(); s.solve_device(b, x); best = min(best, time.time() - t)
        print(f"N={N} cheb{amg['cheb_degree']} {v}: {best*1e3:.1f} ms its={s.get_info()['num_iterations']}", flush=True)
        del s
