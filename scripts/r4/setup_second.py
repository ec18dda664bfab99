"""the north_star block's way of timing a setup: a SECOND full setup on one handle ("amg.reuse" off), PSOLVE_TIMING laps on"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
N = int(os.environ.get("N", "216"))
s = HIPSolver("")
KIND = os.environ.get("KIND", "poisson")
gen = (lambda: s.generate_poisson7(N)) if KIND == "poisson" else (lambda: s.generate_elasticity_q1(N))
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 3 if os.environ.get("KIND", "poisson") != "poisson" else 1, "amg": dict(AMG_RECOMMENDED), "lab.alloc_cache_mb": int(os.environ.get("CACHE_MB", "16384"))}})
gen(); s.synchronize()
s.set_parameters({"HIP": {"amg": {"reuse": os.environ.get("REUSE", "0") == "1"}}})
for rep in range(4):
    if rep == 3 and os.environ.get("LAPS"): os.environ["PSOLVE_TIMING"] = "1"
    print(f"== second setup, rep {rep}", file=sys.stderr, flush=True)
    t = time.perf_counter(); gen(); s.synchronize(); print(f"setup {time.perf_counter()-t:.4f} s  cached {s.get_param('stats.device_bytes_cached')/2**20:.0f} MiB  in use {s.get_param('stats.device_bytes')/2**20:.0f} MiB", file=sys.stderr, flush=True)
