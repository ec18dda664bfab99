"""Round 6, item 6: the plain-CSR Jacobi-PCG iteration beyond the Infinity Cache (384^3, 512^3 on ONE device) by kernel, and
under the schedules / cache policies the library already exposes.  Iterations are capped (the rates do not depend on it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
IT = int(os.environ.get("IT", "64"))
SIZES = [int(v) for v in os.environ.get("SIZES", "256,384,512").split(",")]
VARIANTS = json.loads(os.environ.get("VARIANTS", "null")) or [
    {}, {"spmv_chunk_rows": 2048}, {"spmv_chunk_rows": 32768}, {"spmv_chunk_rows": 131072}, {"spmv_xcd_map": 0}, {"spmv_xcd_map": 1},
    {"vec_policy": 0}, {"vec_policy": 1}, {"vec_policy": 15}, {"spmv_nt": 0}, {"spmv_blocks_per_cu": 4}, {"spmv_blocks_per_cu": 8},
    {"blocks_per_cu": 4}, {"blocks_per_cu": 16}]
for N in SIZES:
    n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
    for v in VARIANTS:
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(dict(tolerance=1e-30, max_iter=IT, spmv_kernel=1, spmv_value_dict=False, profile_spmv=4, true_residual=False), **v)})
        s.generate_poisson7(N)
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e30
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
        i = s.info_struct()
        it = max(int(i.num_iterations), 1)
        k1, k2, k3 = i.spmv_ms_avg, s.get_param("stats.update_r_ms_avg"), s.get_param("stats.update_xp_ms_avg")
        by = 12 * nnz + 100 * n
        print(json.dumps({"N": N, "variant": v, "ms_per_it": best * 1e3 / it, "iter_frac": by / (best / it) / 8e12,
                          "spmv_ms": k1, "spmv_frac": (12 * nnz + 20 * n) / (k1 * 1e-3) / 8e12 if k1 > 0 else None,
                          "k2_ms": k2, "k2_frac": 32 * n / (k2 * 1e-3) / 8e12 if k2 > 0 else None,
                          "k3_ms": k3, "k3_frac": 48 * n / (k3 * 1e-3) / 8e12 if k3 > 0 else None, "kernel": s.last_spmv_kernel()}), flush=True)
        b.free(); x.free()
        del s
