"""spmv_bsr3_dma chain variants (VAR 0..3, -1 = by epilogue) on configs[2]: PCG's product alone and the whole AMG-PCG solve."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
M = int(os.environ.get("M", "100"))
out = []
for rnd in range(2):
    for var in (-1, 0, 1):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, bsr3_variant=var, amg=dict(AMG_RECOMMENDED))})
        s.generate_elasticity_q1(M); s.synchronize()
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        ms = min(s.time_spmv(b, x, 40) for _ in range(2))
        best = 1e9
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
        rec = dict(var=var, dot_ms=round(ms, 4), solve_ms=round(best * 1e3, 2), its=s.get_info()["num_iterations"])
        print(json.dumps(rec), flush=True)
        out.append(rec)
        b.free(); x.free(); del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_bsr_variants.json"), "w"), indent=1)
