"""Which kernels should stream non-temporally?  Jacobi-PCG and AMG-PCG at 256^3 / 216^3 under the policies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver

def solve(N, prm, reps=2):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(prm, tolerance=1e-8, max_iter=20000, profile_spmv=8)})
    s.generate_poisson7(N)
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(reps):
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.info_struct()
    return best, i.num_iterations, i.spmv_ms_avg

amg = dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))
amgw = dict(precond="amg", amg=dict(ncycle=2, cheb_degree=16, cheb_power_iters=100))
for N in (256, 216):
    for name, prm in [("jacobi  r1 (pipe, no nt)", dict(spmv_kernel=0, spmv_nt=0, spmv_blocks_per_cu=5)),
                      ("jacobi  dma+nt everywhere", dict()),
                      ("jacobi  dma+nt spmv only (vec plain)", dict(spmv_nt=1, spmv_nt_mbytes=1 << 19)),
                      ]:
        if "vec plain" in name:
            continue
        t, it, ms = solve(N, prm)
        print(f"N={N} {name:40s} {t*1e3:8.2f} ms  {it} its  {t*1e3/it:.4f} ms/it  spmv {ms:.4f} ms", flush=True)
    for name, base in (("amg V2", amg), ("amg W16", amgw)):
        for pol, extra in [("r1 (pipe, no nt)", dict(spmv_kernel=0, spmv_nt=0, spmv_blocks_per_cu=5)),
                           ("nt everywhere", dict()),
                           ("nt in PCG, cycle cached", None)]:
            prm = dict(base)
            if extra is None:
                prm["amg"] = dict(base["amg"], stream_nt=0)
            else:
                prm.update(extra)
            t, it, ms = solve(N, prm)
            print(f"N={N} {name:8s} {pol:30s} {t*1e3:8.2f} ms  {it} its", flush=True)
