"""Round 3: "reorder" on the bench's unstructured legs (256^3 Poisson under pseudo-random renumberings):
search time, gather spread before / after, in-loop SpMV and whole solves, with and without the renumbering."""
import sys, time, json
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = {}
for name, mode in (("natural", 0), ("windowed_4096", 2), ("random", 1)):
    for reorder in (0, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "profile_spmv": 8, "reorder": reorder}})
        gen = (lambda: s.generate_poisson7(N)) if mode == 0 else (lambda: s.generate_poisson7_permuted(N, N, N, mode=mode, window=4096, seed=7))
        gen(); s.synchronize()
        t = time.perf_counter(); gen(); s.synchronize(); t_fact2 = time.perf_counter() - t   # same pattern: order kept
        n, nnz, _ = s.matrix_shape()
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        for _ in range(2):
            s.axpby_device(n, 0.0, b, 0.0, x)
            s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); dt = time.perf_counter() - t
        i = s.info_struct()
        by = 12 * nnz + 20 * n
        r = {"solve_s": dt, "its": i.num_iterations, "spmv_ms": i.spmv_ms_avg, "frac": by / (i.spmv_ms_avg * 1e-3) / 8e12 if i.spmv_ms_avg else 0,
             "dof_per_s": n / dt, "refactorize_s": t_fact2, "true_res": s.get_info()["true_residual"], "patterns": s.get_param("spmv_patterns")}
        if reorder:
            r.update({k: s.get_param("reorder." + k) for k in ("active", "levels", "spread_before", "spread_after", "seconds")})
            s2 = HIPSolver("")
            s2.set_parameters({"HIP": {"reorder": 1}})
            t = time.perf_counter(); 
            (s2.generate_poisson7(N) if mode == 0 else s2.generate_poisson7_permuted(N, N, N, mode=mode, window=4096, seed=7)); s2.synchronize()
            r["first_factorize_s"] = time.perf_counter() - t
            r["first_reorder_s"] = s2.get_param("reorder.seconds")
            del s2
        out[f"{name}/reorder{reorder}"] = r
        print(name, reorder, json.dumps(r), flush=True)
        b.free(); x.free(); del s
json.dump(out, open("gpurun_out/r03_reorder.json", "w"), indent=1)
