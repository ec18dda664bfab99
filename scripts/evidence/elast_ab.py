"""configs[2] (Q1 elasticity M = 100, block-3 AMG-PCG, bench.py's AMG_RECOMMENDED): "amg.block_levels" 1 against 0
(round 3's cycle) -- setup, numeric refresh, solve, iterations -- on one box, one process."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import BoxSampler, box_static
M = int(os.environ.get("M", "100"))
AMG = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3)
out = {"box": box_static()}
for bl in (0, 1, 0, 1):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(AMG, block_levels=bool(bl)))})
    s.generate_elasticity_q1(M); s.synchronize()
    t = time.perf_counter(); s.generate_elasticity_q1(M); s.synchronize(); t_refresh = time.perf_counter() - t
    s.set_parameters({"HIP": {"amg": {"reuse": False}}})
    t = time.perf_counter(); s.generate_elasticity_q1(M); s.synchronize(); t_setup = time.perf_counter() - t
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    with BoxSampler() as box:
        for _ in range(4):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
    info = s.get_info()
    rec = dict(block_levels=bl, setup_s=t_setup, refresh_s=t_refresh, solve_ms=best * 1e3, iterations=info["num_iterations"],
               true_residual=info["true_residual"], reused=s.get_param("amg.last_setup_reused"), box=box.summary())
    print(json.dumps(rec), flush=True)
    out.setdefault(str(bl), []).append(rec)
    b.free(); x.free(); del s
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "r04_elast_ab.json"), "w"), indent=1)
