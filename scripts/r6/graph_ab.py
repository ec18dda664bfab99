"""Does sampling (profile_spmv > 0: eager launches + events) cost the timed solves anything against the captured loop?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
N = 256; n = N ** 3
for storage in ({"spmv_kernel": 1, "spmv_value_dict": False}, {}):
    hs = []
    for prof, graph in ((8, 1), (0, 1), (0, 0), (64, 1)):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(dict(tolerance=1e-8, max_iter=20000, profile_spmv=prof, use_graph=bool(graph)), **storage)})
        s.generate_poisson7(N)
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        s.axpby_device(n, 0.0, b, 0.0, x); s.solve_device(b, x)
        hs.append((prof, graph, s, b, x, []))
    for r in range(3):
        for prof, graph, s, b, x, acc in hs:
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); acc.append(time.perf_counter() - t)
    for prof, graph, s, b, x, acc in hs:
        i = s.info_struct()
        print(json.dumps({"storage": storage, "profile_spmv": prof, "use_graph": graph, "solve_ms": [round(a * 1e3, 2) for a in acc],
                          "iterations": int(i.num_iterations), "spmv_ms_avg": i.spmv_ms_avg, "kernel": s.last_spmv_kernel()}), flush=True)
