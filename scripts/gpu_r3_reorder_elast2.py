"""Where does the first factorize of the node-shuffled elasticity system spend its time under "reorder"?"""
import sys, time, json
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for reorder in (0, 1):
    for precond in ("jacobi", "amg"):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"precond": precond, "block_size": 3, "reorder": reorder, "amg": dict(AMG_RECOMMENDED, reuse=False)}})
        s.generate_elasticity_q1_permuted(M, mode=1, seed=7); s.synchronize()
        for rep in range(2):
            t = time.perf_counter(); s.generate_elasticity_q1_permuted(M, mode=1, seed=7 + rep); s.synchronize(); dt = time.perf_counter() - t
            print(f"reorder {reorder} {precond} rep {rep}: generate+factorize {dt:.4f} s  reorder.seconds {s.get_param('reorder.seconds'):.4f} levels {s.get_param('reorder.levels')}"
                  f" on-device aggregation levels {s.get_param('amg.levels_aggregated_on_device')} amg levels {s.get_info()['amg_levels']}", flush=True)
        del s
