"""BASELINE.json configs[2]: 3-D linear-elasticity block-3 SPD (Q1, M^3 nodes), Chebyshev-AMG PCG vs Jacobi-PCG."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import HIPSolver

M = int(os.environ.get("M", "100"))
t = time.time(); A = O.elasticity_q1(M); S = A.to_scipy(); print(f"generate M={M}: n={A.n} nnz={A.nnz} {time.time()-t:.1f}s", flush=True)
b = O.spmv(A, O.splitmix_vector(A.n, 42))
for name, params in [("jacobi", {}), ("jacobi bsr3", dict(block_size=3)),
                     ("amg d2 lo.1", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))),
                     ("amg d3 lo1/30", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=3, cheb_lower=1/30, cheb_power_iters=20))),
                     ("amg d4 lo1/120", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=4, cheb_power_iters=20))),
                     ("amg d3 lo.1", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=3, cheb_lower=0.1, cheb_power_iters=20))),
                     ("blk3 d2 lo.1", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))),
                     ("blk3 d2 lo.1 fp32-values", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, matrix_fp32=1))),
                     ("blk3 d3 lo.1", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=3, cheb_lower=0.1, cheb_power_iters=20))),
                     ("blk3 d3 lo1/30", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=3, cheb_lower=1/30, cheb_power_iters=20))),
                     ("blk3 d4 lo1/30", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=4, cheb_lower=1/30, cheb_power_iters=20))),
                     ("blk3 d6 lo1/30", dict(precond="amg", block_size=3, amg=dict(ncycle=1, cheb_degree=6, cheb_lower=1/30, cheb_power_iters=20))),
                     ("blk3 amgcl W d16", dict(precond="amg", block_size=3, amg=dict(ncycle=2, cheb_degree=16, cheb_power_iters=100)))]:
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(params, tolerance=1e-8, max_iter=20000)})
    t = time.time(); s.analyze_pattern(S, A.n); s.factorize(S); tf = time.time() - t
    db, dx = s.to_device(b), s.to_device(np.zeros(A.n))
    s.solve_device(db, dx); dx.upload(np.zeros(A.n))
    t = time.time(); s.solve_device(db, dx); ts = time.time() - t
    i = s.get_info()
    lv = [s.amg_level_info(l)[:2] for l in range(i["amg_levels"])]
    print(f"{name}: factorize {tf:.2f}s solve {ts*1e3:.1f} ms iters={i['num_iterations']} true={i['true_residual']:.2e} status={i['solver_status']} "
          f"DOF/s={A.n/ts:.3e} R={s.get_param('spmv_rows_per_block')} levels={lv}", flush=True)
    if name in ("jacobi", "blk3 d2 lo.1"):
        dy = s.device_array(A.n)
        ms = s.time_spmv(dx, dy, 20)
        print(f"   spmv {ms:.4f} ms -> {(12*A.nnz+20*A.n)/ms/1e6:.0f} GB/s alg")
    del s
