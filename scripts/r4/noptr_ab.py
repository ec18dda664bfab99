"""spmv_csr_pat without row pointers ("lab.pat_noptr"): bit-equality and time against the kernel that reads them, N^3 Poisson."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
out = []
for N in [int(v) for v in os.environ.get("NS", "256,200").split(",")]:
    xs = {}
    for rep in range(2):
        for flag in (0, 1):
            s = HIPSolver("")
            s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "lab.pat_noptr": flag, "profile_spmv": 8}})
            s.generate_poisson7(N, N, N); s.synchronize()
            n = s.matrix_shape()[0]
            b, x = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, b)
            ms = min(s.time_spmv(b, x, reps=20) for _ in range(3))
            best = 1e9
            for _ in range(3):
                s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
                t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
            i = s.info_struct()
            xs[flag] = x.download()
            rec = dict(N=N, noptr=flag, spmv_ms=round(ms, 4), in_loop_ms=round(i.spmv_ms_avg, 4), solve_ms=round(best * 1e3, 2), its=i.num_iterations, mdofs=round(n / best / 1e6, 2))
            # AMG on the same system
            s.set_parameters({"HIP": {"precond": "amg", "amg": dict(AMG_RECOMMENDED)}})
            s.generate_poisson7(N, N, N); s.synchronize()
            best = 1e9
            for _ in range(3):
                s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
                t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
            rec.update(amg_ms=round(best * 1e3, 2), amg_its=s.get_info()["num_iterations"])
            print(json.dumps(rec), flush=True); out.append(rec)
            b.free(); x.free(); del s
    print("bit-equal solutions:", bool(np.array_equal(xs[0], xs[1])), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_noptr_ab.json"), "w"), indent=1)
