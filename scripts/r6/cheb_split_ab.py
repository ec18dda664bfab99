"""fused block Chebyshev step against residual product + update launch on level 0 (configs[2] under a random node numbering)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
hs = []
for split in (0, 1000000, 0, 1000000):
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 3, "lab.cheb_split": split, "amg": dict(AMG_RECOMMENDED)}})
    s.generate_elasticity_q1_permuted(100, mode=1, seed=7); s.synchronize()
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    hs.append((split, s, b, x, []))
for r in range(4):
    for split, s, b, x, acc in hs:
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); acc.append(time.perf_counter() - t)
for split, s, b, x, acc in hs:
    i = s.get_info()
    print(json.dumps({"cheb_split": split, "solve_ms": [round(a * 1e3, 2) for a in acc], "iterations": int(i["num_iterations"]), "ops_level0_us": s.amg_time_level_ops(0, 10)}), flush=True)
