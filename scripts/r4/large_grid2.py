"""chunk-size sweep of the XCD schedule on large grids (follow-up of large_grid.py)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
out = []
for N in [int(v) for v in os.environ.get("NS", "320,384,448,512").split(",")]:
    for kern in (3, 1):
        row = {}
        for chunk in (8192, 32768, 131072, 524288, 1 << 20, 1 << 22):
            s = HIPSolver("")
            s.set_parameters({"HIP": {"spmv_kernel": kern, "tolerance": 1e-8, "spmv_chunk_rows": chunk}})
            s.generate_poisson7(N, N, N); s.synchronize()
            n, nnz, _ = s.matrix_shape()
            x, y = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, x)
            ms = min(s.time_spmv(x, y, reps=10) for _ in range(3))
            by = (8 * nnz + 22 * n) if (kern == 3 and s.get_param("spmv_patterns") > 0) else (12 * nnz + 20 * n)
            row[chunk] = round(by / ms / 1e6 / 8000, 3)
            x.free(); y.free(); del s
        print(N, "pat" if kern == 3 else "csr", row, flush=True)
        out.append(dict(N=N, kernel=kern, frac_by_chunk=row))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_large_grid_chunks.json"), "w"), indent=1)
