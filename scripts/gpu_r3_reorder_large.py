"""One large single-device system under a random numbering (384^3 = 56.6 M rows, 3.95e8 entries): the renumbering at its size."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for reorder in (2, 0):
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "reorder": reorder, "profile_spmv": 8}})
    t = time.perf_counter(); s.generate_poisson7_permuted(N, N, N, mode=1, seed=7); s.synchronize(); tf = time.perf_counter() - t
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b); s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
    t = time.perf_counter(); s.solve_device(b, x); dt = time.perf_counter() - t
    i = s.info_struct()
    print(json.dumps({"N": N, "reorder": reorder, "active": s.get_param("reorder.active"), "generate_plus_factorize_s": tf, "search_s": s.get_param("reorder.seconds"),
                      "levels": s.get_param("reorder.levels"), "solve_s": dt, "its": i.num_iterations, "true_residual": s.get_info()["true_residual"],
                      "spmv_ms": i.spmv_ms_avg, "frac": (12 * nnz + 20 * n) / (i.spmv_ms_avg * 1e-3) / 8e12, "dof_per_s": n / dt,
                      "device_GiB_peak": s.get_param("stats.device_bytes_peak") / 2**30}), flush=True)
    b.free(); x.free(); del s
