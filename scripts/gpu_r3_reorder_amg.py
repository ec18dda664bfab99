"""Round 3: AMG-PCG (recommended configuration) on the bench matrix under pseudo-random renumberings, in the caller's
numbering ("reorder" 0) and renumbered at factorize ("reorder" 1): setup, solve, iterations."""
import sys, time, json
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED

N = int(sys.argv[1]) if len(sys.argv) > 1 else 216
out = {}
for name, mode in (("natural", 0), ("windowed_4096", 2), ("random", 1)):
    for reorder in (0, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "reorder": reorder, "amg": dict(AMG_RECOMMENDED)}})
        gen = (lambda: s.generate_poisson7(N)) if mode == 0 else (lambda: s.generate_poisson7_permuted(N, N, N, mode=mode, window=4096, seed=7))
        gen(); s.synchronize()
        s.set_parameters({"HIP": {"amg": {"reuse": False}}})
        t = time.perf_counter(); gen(); s.synchronize(); t_setup = time.perf_counter() - t
        s.set_parameters({"HIP": {"amg": {"reuse": True}}})
        gen(); s.synchronize()
        t = time.perf_counter(); gen(); s.synchronize(); t_refresh = time.perf_counter() - t
        n, nnz, _ = s.matrix_shape()
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e9
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x)
            s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        r = {"setup_s": t_setup, "refresh_s": t_refresh, "solve_s": best, "its": int(i["num_iterations"]), "levels": int(i["amg_levels"]),
             "dof_per_s": n / best, "true_res": i["true_residual"], "reorder_active": s.get_param("reorder.active"),
             "level_rows": [s.amg_level_info(l)[0] for l in range(int(i["amg_levels"]))]}
        out[f"{name}/reorder{reorder}"] = r
        print(name, reorder, json.dumps(r), flush=True)
        b.free(); x.free(); del s
json.dump(out, open("gpurun_out/r03_reorder_amg.json", "w"), indent=1)
