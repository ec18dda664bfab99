"""Round 3: what the 16-bit columns give -- the same runs with "spmv_col16" on / off."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
out = {}
def run(tag, gen, hip, reps=3):
    for c16 in (1, 0):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict({"tolerance": 1e-8, "max_iter": 20000, "profile_spmv": 4, "spmv_col16": bool(c16)}, **hip)})
        gen(s); s.synchronize()
        n, nnz, _ = s.matrix_shape()
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e9
        for _ in range(reps):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.info_struct()
        r = {"solve_ms": best * 1e3, "its": i.num_iterations, "spmv_us": i.spmv_ms_avg * 1e3, "col16_active": s.get_param("col16_active"),
             "bytes_per_launch": (10 if s.get_param("col16_active") else 12) * nnz + 20 * n}
        r["tbs"] = r["bytes_per_launch"] / (r["spmv_us"] * 1e-6) / 1e12 if r["spmv_us"] else 0
        out[f"{tag}/col16={c16}"] = r
        print(tag, c16, json.dumps(r), flush=True)
        b.free(); x.free(); del s
run("poisson256 random numbering, jacobi (renumbered)", lambda s: s.generate_poisson7_permuted(256, 256, 256, mode=1, seed=7), {})
run("poisson256 windows, jacobi (renumbered)", lambda s: s.generate_poisson7_permuted(256, 256, 256, mode=2, seed=7), {})
run("poisson256 grid, amg", lambda s: s.generate_poisson7(256), {"precond": "amg", "amg": dict(AMG_RECOMMENDED)})
run("poisson216 grid, amg", lambda s: s.generate_poisson7(216), {"precond": "amg", "amg": dict(AMG_RECOMMENDED)})
run("poisson216 random, amg (renumbered)", lambda s: s.generate_poisson7_permuted(216, 216, 216, mode=1, seed=7), {"precond": "amg", "amg": dict(AMG_RECOMMENDED)})
run("elasticity M=100, block-3 amg", lambda s: s.generate_elasticity_q1(100), {"precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED)})
json.dump(out, open("gpurun_out/r03_col16.json", "w"), indent=1)
