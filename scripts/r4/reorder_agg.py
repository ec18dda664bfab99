"""configs[2] with its nodes renumbered pseudo-randomly (bench.py's elasticity.unstructured leg) and 216^3 Poisson under a
random numbering: iterations / solve time under "reorder_reverse" 0 / 1, and the grid numbering."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED, time_solves
M = int(os.environ.get("M", "100")); N = int(os.environ.get("N", "216"))
out = []
def run(tag, gen, block, extra):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(dict(tolerance=1e-8, max_iter=20000, precond="amg", block_size=block, amg=dict(AMG_RECOMMENDED)), **extra)})
    gen(s); s.synchronize()
    t = time.perf_counter(); gen(s); s.synchronize(); t_ref = time.perf_counter() - t
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        dt, its, ms, smp, info = time_solves(s, b, x, n)
        best = min(best, dt)
    rec = dict(tag=tag, extra=extra, iterations=its, solve_ms=best * 1e3, refresh_s=t_ref, reordered=bool(s.get_param("reorder.active")),
               levels=[s.amg_level_info(l)[0] for l in range(int(info["amg_levels"]))], spread_after=s.get_param("reorder.spread_after"))
    print(json.dumps(rec), flush=True); out.append(rec)
    b.free(); x.free(); del s
for extra in (dict(reorder=2), dict(reorder=2, reorder_reverse=1)):
    run("elasticity grid", lambda s: s.generate_elasticity_q1(M), 3, extra)
    run("elasticity random nodes", lambda s: s.generate_elasticity_q1_permuted(M, mode=1, seed=7), 3, extra)
    run("poisson random", lambda s: s.generate_poisson7_permuted(N, N, N, mode=2, window=4096, seed=7), 1, extra)
    run("poisson windows", lambda s: s.generate_poisson7_permuted(N, N, N, mode=1, window=4096, seed=7), 1, extra)
run("poisson grid", lambda s: s.generate_poisson7(N, N, N), 1, dict(reorder=2))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_reorder_agg.json"), "w"), indent=1)
