"""SpMV beyond the Infinity Cache: N^3 Poisson on one device, the in-loop product's time and fraction
for the schedule variants (XCD map, chunk rows, nt policy), dictionary kernel and plain CSR."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
out = []
for N in [int(v) for v in os.environ.get("NS", "256,384,512").split(",")]:
    for kern in (3, 1):
        for tag, extra in (("default", {}), ("chunk64k", {"spmv_chunk_rows": 65536}), ("chunk1M", {"spmv_chunk_rows": 1 << 20}), ("xcd1", {"spmv_xcd_map": 1}), ("xcd0", {"spmv_xcd_map": 0}),
                           ("wg6", {"spmv_blocks_per_cu": 6}), ("wg10", {"spmv_blocks_per_cu": 10})):
            s = HIPSolver("")
            s.set_parameters({"HIP": dict({"spmv_kernel": kern, "tolerance": 1e-8}, **extra)})
            s.generate_poisson7(N, N, N); s.synchronize()
            n, nnz, _ = s.matrix_shape()
            x, y = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, x)
            ms = min(s.time_spmv(x, y, reps=10) for _ in range(3))
            by = (8 * nnz + 22 * n) if (kern == 3 and s.get_param("spmv_patterns") > 0) else (12 * nnz + 20 * n)
            rec = dict(N=N, kernel="pat" if kern == 3 else "csr", variant=tag, ms=round(ms, 4), gbs=round(by / ms / 1e6), frac=round(by / ms / 1e6 / 8000, 3))
            print(json.dumps(rec), flush=True); out.append(rec)
            x.free(); y.free(); del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_large_grid.json"), "w"), indent=1)
