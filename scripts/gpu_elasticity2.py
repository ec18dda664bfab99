"""configs[2] on the device-side generator: M = 100 Q1 elasticity, block-3 AMG-PCG; kernel breakdown via PSOLVE env."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
for name, prm in [][:0] + [("blk3 V cheb2 lo.1", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))),
                  ("blk3 V cheb3 lo.1", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=3, cheb_lower=0.1, cheb_power_iters=20))),
                  ("blk3 amgcl W16", dict(precond="amg", block_size=3, amg=dict(ncycle=2, cheb_degree=16, cheb_power_iters=100))),
                  ("jacobi bsr3", dict(block_size=3)),
                  ("schwarz L3 bs3", dict(precond="schwarz", block_size=3)), ("schwarz L1 bs3", dict(precond="schwarz", block_size=3, schwarz=dict(levels=1)))]:
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=20000)})
    t = time.time(); s.generate_elasticity_q1(M); s.synchronize(); tf = time.time() - t
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); best = min(best, time.time() - t)
    i = s.get_info()
    print(f"M={M} {name:20s} setup {tf:.3f}s solve {best*1e3:8.1f} ms its={i['num_iterations']:4d} true={i['true_residual']:.2e} DOF/s={n/best:.3e}", flush=True)
    del s
