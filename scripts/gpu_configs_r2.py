"""Round-2 measurements of the BASELINE configs on one MI355X (device-resident inputs unless noted)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver

def solve(label, gen, prm, reps=2):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=20000)})
    gen(s)  # warm-up (code objects, allocations)
    s.set_parameters({"HIP": {"amg": {"reuse": False}}})
    t = time.time(); gen(s); s.synchronize(); tf = time.time() - t
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(reps):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); best = min(best, time.time() - t)
    i = s.get_info()
    print(f"CFG {label:44s} n={n:10d} setup {tf:6.3f} s  solve {best*1e3:8.1f} ms  its={i['num_iterations']:4d}  true={i['true_residual']:.2e}  {n/best/1e6:7.1f} M DOF/s", flush=True)

V = dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))
W = dict(precond="amg", amg=dict(ncycle=2, cheb_degree=16, cheb_power_iters=100))
for N in (64, 128, 216, 256, 512):
    solve(f"poisson {N}^3 jacobi-pcg", lambda s: s.generate_poisson7(N), {})
    solve(f"poisson {N}^3 amg-pcg V cheb2", lambda s: s.generate_poisson7(N), V, reps=2)
    if N in (216, 256):
        solve(f"poisson {N}^3 amg-pcg AMGCL config (W, cheb16)", lambda s: s.generate_poisson7(N), W, reps=1)
M = 100
solve(f"elasticity M={M} jacobi-pcg (bsr3)", lambda s: s.generate_elasticity_q1(M), dict(block_size=3))
solve(f"elasticity M={M} block-3 amg-pcg V cheb2", lambda s: s.generate_elasticity_q1(M), dict(V, block_size=3))
solve(f"elasticity M={M} block-3 amg-pcg AMGCL config", lambda s: s.generate_elasticity_q1(M), dict(W, block_size=3), reps=1)
solve(f"elasticity M={M} schwarz (1 level, block 3)", lambda s: s.generate_elasticity_q1(M), dict(precond="schwarz", block_size=3), reps=1)
