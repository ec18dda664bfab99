"""Round 6, item 4: first factorize and solve under amg.aggregation amgcl | parallel | compact (x direct_coarse) on configs[2]
(Q1 elasticity M = 100, block 3), on the same matrix under a random node numbering, and on Poisson 216^3 (north_star)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
CASES = os.environ.get("CASES", "elast,elast_random,poisson216").split(",")
VARIANTS = json.loads(os.environ.get("VARIANTS", "null")) or [
    {"aggregation": "amgcl"}, {"aggregation": "amgcl", "direct_coarse": True}, {"aggregation": "parallel"},
    {"aggregation": "compact"}, {"aggregation": "compact", "direct_coarse": True},
    {"aggregation": "compact", "coarse_enough": 500, "direct_coarse": True}]
for case in CASES:
    for v in VARIANTS:
        s = HIPSolver("")
        amg = dict(AMG_RECOMMENDED, **v)
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 1 if case.startswith("poisson") else 3,
                                  "amg": amg}})
        if case == "elast": gen = lambda: s.generate_elasticity_q1(100)
        elif case == "elast_random": gen = lambda: s.generate_elasticity_q1_permuted(100, mode=1, seed=7)
        else: gen = lambda: s.generate_poisson7(int(case[7:]))
        try:
            gen(); s.synchronize()                       # warm-up (code objects, first-touch allocations)
            s.set_parameters({"HIP": {"amg": {"reuse": False}}})
            setups = []
            for _ in range(3):
                t = time.perf_counter(); gen(); s.synchronize(); setups.append(time.perf_counter() - t)
            n = s.matrix_shape()[0]
            b, x = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, b)
            best = 1e30
            for _ in range(3):
                s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
                t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
            i = s.get_info()
            lv = [s.amg_level_info(l)[0] for l in range(int(i["amg_levels"]))]
            print(json.dumps({"case": case, "amg": v, "generate_plus_setup_ms": [round(t * 1e3, 1) for t in setups], "solve_ms": round(best * 1e3, 2),
                              "iterations": int(i["num_iterations"]), "levels": lv, "true_residual": i["true_residual"]}), flush=True)
            b.free(); x.free()
        except Exception as e:
            print(json.dumps({"case": case, "amg": v, "failed": str(e)[:300]}), flush=True)
        del s
