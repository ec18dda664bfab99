"""Round 6: first factorize of configs[2] against the cap of the handle's block cache (lab.alloc_cache_mb)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
for case in ("elast", "elast_random"):
    for mb in (4096, 16384, 4096, 16384):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED), "lab.alloc_cache_mb": mb}})
        gen = (lambda: s.generate_elasticity_q1(100)) if case == "elast" else (lambda: s.generate_elasticity_q1_permuted(100, mode=1, seed=7))
        gen(); s.synchronize()
        s.set_parameters({"HIP": {"amg": {"reuse": False}}})
        ts = []
        for _ in range(3):
            t = time.perf_counter(); gen(); s.synchronize(); ts.append(round((time.perf_counter() - t) * 1e3, 1))
        s.set_parameters({"HIP": {"amg": {"reuse": True}}})
        tr = []
        for _ in range(3):
            t = time.perf_counter(); gen(); s.synchronize(); tr.append(round((time.perf_counter() - t) * 1e3, 1))
        print(json.dumps({"case": case, "alloc_cache_mb": mb, "generate_plus_setup_ms": ts, "generate_plus_refresh_ms": tr,
                          "cached_mb": s.get_param("stats.device_bytes_cached") / 2 ** 20}), flush=True)
        del s
