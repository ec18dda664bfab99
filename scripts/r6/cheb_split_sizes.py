"""where does the split Chebyshev step pay?  level-0 cheb_step us, fused against split, by size (random node numbering)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
for M in (20, 30, 40, 50, 64, 80, 100):
    row = {"M": M, "rows": 3 * M ** 3}
    for split in (0, 1, -1):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 3, "reorder": 1, "lab.cheb_split": split, "amg": dict(AMG_RECOMMENDED)}})
        s.generate_elasticity_q1_permuted(M, mode=1, seed=7); s.synchronize()
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e30
        for _ in range(4):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
        row[f"split{split}"] = {"solve_ms": round(best * 1e3, 3), "it": int(s.get_info()["num_iterations"]), "cheb_step_us": round(s.amg_time_level_ops(0, 10)["cheb_step_us"], 1)}
        b.free(); x.free(); del s
    print(json.dumps(row), flush=True)
