"""One unstructured leg for the profilers: 256^3 Poisson under a random renumbering, factorize (+ search) and solves.
usage: gpu_r3_reorder_one.py <reorder 0|1|2> [max_iter]"""
import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
reorder = int(sys.argv[1]); max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
N = 256
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": max_iter, "reorder": reorder, "profile_spmv": 8}})
s.generate_poisson7_permuted(N, N, N, mode=1, seed=7)
n, nnz, _ = s.matrix_shape()
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
for _ in range(2):
    s.axpby_device(n, 0.0, b, 0.0, x)
    s.synchronize(); t = time.perf_counter(); s.solve_device(b, x); dt = time.perf_counter() - t
i = s.info_struct()
print(json.dumps({"reorder": reorder, "active": s.get_param("reorder.active"), "solve_s": dt, "its": i.num_iterations, "spmv_ms": i.spmv_ms_avg,
                  "frac": (12 * nnz + 20 * n) / (i.spmv_ms_avg * 1e-3) / 8e12, "search_s": s.get_param("reorder.seconds")}))
