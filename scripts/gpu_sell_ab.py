"""amg.sell on / off: configs[2] (Q1 elasticity M = 100, block-3 AMG-PCG), the same system as scalar CSR, and
the 256^3 Poisson AMG-PCG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
amg = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)
def run(name, gen, prm):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=20000)})
    t = time.time(); gen(s); s.synchronize(); tf = time.time() - t
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); best = min(best, time.time() - t)
    i = s.get_info()
    print(f"{name:34s} setup {tf:.3f}s solve {best*1e3:8.1f} ms its={i['num_iterations']:4d} true={i['true_residual']:.2e}", flush=True)
    del s
for sell in (True, False):
    run(f"elast M={M} blk3 amg sell={sell}", lambda s: s.generate_elasticity_q1(M), dict(precond="amg", block_size=3, amg=dict(amg, sell=int(sell))))
for k in (-1, 1):
    run(f"elast M={M} scalar jacobi kernel={k}", lambda s: s.generate_elasticity_q1(M), dict(spmv_kernel=k))
    run(f"elast M={M} scalar amg kernel={k}", lambda s: s.generate_elasticity_q1(M), dict(precond="amg", spmv_kernel=k, amg=dict(amg, sell=int(k < 0))))
for sell in (True, False):
    run(f"poisson 256 amg sell={sell}", lambda s: s.generate_poisson7(256), dict(precond="amg", amg=dict(amg, sell=int(sell))))
    run(f"poisson 216 amg sell={sell}", lambda s: s.generate_poisson7(216), dict(precond="amg", amg=dict(amg, sell=int(sell))))
