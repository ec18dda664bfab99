"""Q1 elasticity M = 100 (BASELINE.json configs[2]) block-3 AMG-PCG: smoother / cycle sweep, one line per setting."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
M = int(os.environ.get("M", "100"))
base = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)
cfgs = [("r2 default", {}),
        ("deg3", dict(cheb_degree=3)), ("deg4", dict(cheb_degree=4)),
        ("deg3 lower .05", dict(cheb_degree=3, cheb_lower=0.05)), ("deg4 lower .05", dict(cheb_degree=4, cheb_lower=0.05)),
        ("deg2 lower .2", dict(cheb_lower=0.2)), ("deg2 lower .05", dict(cheb_lower=0.05)),
        ("deg3 lower .2", dict(cheb_degree=3, cheb_lower=0.2)),
        ("deg2 npost0", dict(npost=0)), ("deg3 npost0", dict(cheb_degree=3, npost=0)), ("deg4 npost0", dict(cheb_degree=4, npost=0)),
        ("deg2 npre0", dict(npre=0)), ("deg4 npre0", dict(cheb_degree=4, npre=0)),
        ("W deg2", dict(ncycle=2)), ("deg2 higher1.1", dict(cheb_higher=1.1)), ("deg3 higher1.1", dict(cheb_degree=3, cheb_higher=1.1)),
        ("deg2 fp32", dict(matrix_fp32=1)), ("deg3 fp32", dict(cheb_degree=3, matrix_fp32=1)),
        ("deg2 eps .08", dict(eps_strong=0.08)), ("deg3 eps .08", dict(cheb_degree=3, eps_strong=0.08)),
        ("deg2 relax .7", dict(sa_relax=0.7)), ("deg2 relax 1.3", dict(sa_relax=1.3))]
for name, extra in cfgs:
    s = HIPSolver("")
    amg = dict(base, **extra)
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, max_iter=2000, amg=amg)})
    t = time.perf_counter(); s.generate_elasticity_q1(M); s.synchronize(); ts = time.perf_counter() - t
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    lv = [s.amg_level_info(l)[0] for l in range(i["amg_levels"])]
    print(f"{name:18s} solve {best*1e3:7.1f} ms its={i['num_iterations']:4d} ms/it={best*1e3/max(i['num_iterations'],1):.2f} res={i['true_residual']:.1e} levels={lv} setup {ts:.2f}", flush=True)
    del s
