"""PSOLVE_TIMING laps of one full setup (the second on a handle) under an option set: env KIND, N, AMG (JSON merged into /HIP/amg)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
KIND = os.environ.get("KIND", "poisson"); N = int(os.environ.get("N", "216"))
amg = dict(AMG_RECOMMENDED); amg.update(json.loads(os.environ.get("AMG", "{}")))
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 1 if KIND == "poisson" else 3, "amg": amg}})
gen = (lambda: s.generate_poisson7(N)) if KIND == "poisson" else (lambda: s.generate_elasticity_q1(N))
gen(); s.synchronize()
s.set_parameters({"HIP": {"amg": {"reuse": False}}})
gen(); s.synchronize()
os.environ["PSOLVE_TIMING"] = "1"
t = time.perf_counter(); gen(); s.synchronize(); print(f"setup {time.perf_counter() - t:.4f} s", file=sys.stderr)
