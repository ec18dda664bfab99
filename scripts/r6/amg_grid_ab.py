"""blocks_per_cu (the persistent grid the AMG levels' vector kernels and the setup kernels are fitted under) on AMG-PCG."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
for case in ("poisson216", "poisson256", "elast_random", "elast"):
    for bpc in (8, 4, 2):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "block_size": 1 if case.startswith("poisson") else 3,
                                  "blocks_per_cu": bpc, "amg": dict(AMG_RECOMMENDED)}})
        if case == "elast": gen = lambda: s.generate_elasticity_q1(100)
        elif case == "elast_random": gen = lambda: s.generate_elasticity_q1_permuted(100, mode=1, seed=7)
        else: gen = lambda: s.generate_poisson7(int(case[7:]))
        gen(); s.synchronize()
        s.set_parameters({"HIP": {"amg": {"reuse": False}}})
        setups = []
        for _ in range(3):
            t = time.perf_counter(); gen(); s.synchronize(); setups.append(time.perf_counter() - t)
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e30
        for _ in range(4):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        print(json.dumps({"case": case, "blocks_per_cu": bpc, "setup_ms": round(min(setups) * 1e3, 1), "solve_ms": round(best * 1e3, 2),
                          "iterations": int(i["num_iterations"])}), flush=True)
        b.free(); x.free(); del s
