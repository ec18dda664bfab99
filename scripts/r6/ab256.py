"""A/B of schedule knobs on the bench system itself (256^3, plain CSR, full solves), variants interleaved, R rounds."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "256")); R = int(os.environ.get("R", "3"))
VARIANTS = json.loads(os.environ.get("VARIANTS", "null")) or [
    {}, {"spmv_chunk_rows": 1024}, {"spmv_chunk_rows": 2048}, {"spmv_chunk_rows": 4096}, {"spmv_chunk_rows": 131072}, {"spmv_xcd_map": 1},
    {"blocks_per_cu": 16}, {"spmv_chunk_rows": 2048, "blocks_per_cu": 16}, {"spmv_xcd_map": 1, "blocks_per_cu": 16}, {"vec_policy": 15},
    {"spmv_chunk_rows": 2048, "vec_policy": 15}]
n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
hs = []
for v in VARIANTS:
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(dict(tolerance=1e-8, max_iter=20000, spmv_kernel=1, spmv_value_dict=False, profile_spmv=8), **v)})
    s.generate_poisson7(N)
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    s.axpby_device(n, 0.0, b, 0.0, x); s.solve_device(b, x)
    hs.append((v, s, b, x, []))
for r in range(R):
    for v, s, b, x, acc in hs:
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); dt = time.perf_counter() - t
        i = s.info_struct()
        acc.append((dt, i.spmv_ms_avg, s.get_param("stats.update_r_ms_avg"), s.get_param("stats.update_xp_ms_avg"), int(i.num_iterations)))
for v, s, b, x, acc in hs:
    dt = min(a[0] for a in acc); k1 = min(a[1] for a in acc)
    print(json.dumps({"variant": v, "solve_ms": [round(a[0] * 1e3, 1) for a in acc], "dof_per_s_best": n / dt, "spmv_ms": [round(a[1], 4) for a in acc],
                      "spmv_frac_best": (12 * nnz + 20 * n) / (k1 * 1e-3) / 8e12, "k2_ms": [round(a[2], 4) for a in acc], "k3_ms": [round(a[3], 4) for a in acc],
                      "iterations": acc[0][4]}), flush=True)
