"""cheb_higher (AMGCL's safety factor on the power-iteration estimate of rho, default 2) on configs[2] and on Poisson."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
def run(tag, gen, prm, amg):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(prm, precond="amg", tolerance=1e-8, max_iter=500, amg=amg)})
    gen(s); s.synchronize()
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(2):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    rho = [round(s.amg_level_info(l)[2], 4) for l in range(i["amg_levels"])]
    print(f"{tag:44s} solve {best*1e3:7.1f} ms its={i['num_iterations']:4d} res={i['true_residual']:.1e} rho={rho}", flush=True)
    del s
base = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20)
E = lambda s: s.generate_elasticity_q1(100)
for extra in ({}, dict(cheb_higher=1.3), dict(cheb_higher=1.2), dict(cheb_higher=1.1), dict(cheb_higher=1.05), dict(cheb_higher=1.0),
              dict(cheb_higher=1.1, cheb_power_iters=50), dict(cheb_higher=1.1, sa_relax=1.3), dict(cheb_higher=1.2, sa_relax=1.3),
              dict(cheb_higher=1.1, cheb_lower=0.15), dict(cheb_higher=1.1, cheb_lower=0.07), dict(cheb_higher=1.1, cheb_degree=3),
              dict(cheb_higher=1.1, matrix_fp32=1), dict(cheb_higher=1.1, sa_relax=1.3, matrix_fp32=1)):
    run("elast100 " + str(extra), E, dict(block_size=3), dict(base, **extra))
for N in (216, 256):
    P = lambda s: s.generate_poisson7(N)
    for extra in ({}, dict(cheb_higher=1.2), dict(cheb_higher=1.1), dict(cheb_higher=1.05), dict(cheb_higher=1.1, cheb_lower=0.15),
                  dict(cheb_higher=1.1, cheb_lower=0.07), dict(cheb_higher=1.1, cheb_degree=3), dict(cheb_higher=1.1, sa_relax=1.3)):
        run(f"poisson{N} " + str(extra), P, {}, dict(base, **extra))
