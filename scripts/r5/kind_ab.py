"""Row kinds (spmv_value_dict) A/B: Jacobi-PCG at 256^3 and the recommended AMG configuration at 216^3, on / off."""
import json, sys, time
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED


def run(N, precond, vd, grid=None, nt=None, unroll=None, chunk=None, sched=None, bpc=None):
    hip = {"tolerance": 1e-8, "max_iter": 20000, "spmv_value_dict": bool(vd)}
    if grid:
        hip["spmv_grid"] = grid
    if nt is not None:
        hip["spmv_nt"] = nt
    if unroll:
        hip["lab.kind_unroll"] = unroll
    if chunk:
        hip["spmv_chunk_rows"] = chunk
    if sched is not None:
        hip["lab.kind_sched"] = sched
    if bpc:
        hip["spmv_blocks_per_cu"] = bpc
    if precond == "amg":
        hip.update(precond="amg", amg=dict(AMG_RECOMMENDED))
    s = HIPSolver("")
    s.set_parameters({"HIP": hip})
    s.generate_poisson7(N)
    s.synchronize()
    t0 = time.perf_counter()
    s.generate_poisson7(N)
    s.synchronize()
    tf = time.perf_counter() - t0
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.synchronize()
        t0 = time.perf_counter()
        s.solve_device(b, x)
        best = min(best, time.perf_counter() - t0)
    i = s.get_info()
    return {"N": N, "precond": precond, "value_dict": vd, "grid": grid, "nt": nt, "unroll": unroll, "chunk": chunk, "sched": sched, "bpc": bpc, "kinds": s.get_param("spmv_row_kinds"),
            "generate_factorize_s": tf, "solve_s": best, "iters": int(i["num_iterations"]), "res": i["true_residual"],
            "spmv_ms_avg": i["spmv_ms_avg"], "kernel": s.last_spmv_kernel(), "mdofs": n / best / 1e6}


if __name__ == "__main__":
    for args in [(256, "jacobi", 1, None, None, 1, None, -1), (256, "jacobi", 1, None, None, 1, None, -1, 4),
                 (256, "jacobi", 1, None, None, 1, None, -1, 5), (256, "jacobi", 1, None, None, 1, None, -1, 6),
                 (256, "jacobi", 1, None, None, 1, None, -1, 10), (256, "jacobi", 1, None, None, 1, None, -1, 12),
                 (256, "jacobi", 1, None, 1, 1, None, -1), (216, "amg", 1, None, None, 1, None, -1)]:
        try:
            print(json.dumps(run(*args)), flush=True)
        except Exception as e:
            print("ERR", args, e, flush=True)
