"""Block-row kinds A/B on BASELINE configs[2] (Q1 elasticity M = 100, block-3 AMG-PCG): setup, refresh, solve; products."""
import json, sys, time
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for kinds in (0, 1, 0, 1):
    s = HIPSolver("")
    s.set_parameters({"HIP": {"precond": "amg", "block_size": 3, "tolerance": 1e-8, "max_iter": 5000, "amg": dict(AMG_RECOMMENDED),
                              "lab.bsr3_kinds": kinds}})
    s.generate_elasticity_q1(M); s.synchronize()
    t = time.perf_counter(); s.generate_elasticity_q1(M); s.synchronize(); t_refresh = time.perf_counter() - t
    n = s.matrix_shape()[0]
    b, x, y = s.device_array(n), s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    for _ in range(5): s.spmv_device(b, y)
    s.synchronize(); t = time.perf_counter()
    for _ in range(50): s.spmv_device(b, y)
    s.synchronize(); t_spmv = (time.perf_counter() - t) / 50
    print(json.dumps({"M": M, "kinds_knob": kinds, "row_kinds": s.get_param("bsr3_row_kinds"), "blocks": s.get_param("bsr3_kind_blocks"),
                      "generate_plus_refresh_s": t_refresh, "solve_s": best, "iters": int(i["num_iterations"]), "res": i["true_residual"],
                      "spmv_us": t_spmv * 1e6, "kernel": s.last_spmv_kernel()}), flush=True)
s.set_parameters({"HIP": {"lab.bsr3_kinds": 1}})
