"""BSR-3 product on the Q1 elasticity matrix (M = 100): plain / nt loads, serial / eight-lane row sums."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
for nt in (0, 1):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(block_size=3, spmv_nt=nt, precond="amg", tolerance=1e-8,
                                  amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))})
    s.generate_elasticity_q1(M)
    n, nnz, _ = s.matrix_shape()
    x, y = s.device_array(n), s.device_array(n)
    s.generate_rhs(7, x)
    ms = min(s.time_spmv(x, y, 20) for _ in range(3))
    nb = n // 3
    nnzb = int(s.get_param("bsr3_blocks")) if False else None
    b, z = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, z); s.synchronize()
        t = time.time(); s.solve_device(b, z); best = min(best, time.time() - t)
    print(f"PD={os.environ.get('PSOLVE_BSR_PD','0')} nt={nt}: bsr3 product {ms:.4f} ms; AMG-PCG solve {best*1e3:.1f} ms its={s.get_info()['num_iterations']} true={s.get_info()['true_residual']:.2e}", flush=True)
    del s
