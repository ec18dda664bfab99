"""Round 6: whose placement moves the contract kernel -- the matrix's blocks or the vectors'?  H handles (matrix placements) x
X input vectors x Y output vectors, the plain product timed back to back (psolve_hip_time_spmv, ms per launch)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polysolve_amd import HIPSolver
N = 256
n, nnz = N ** 3, 7 * N ** 3 - 6 * N ** 2
hs = []
for h in range(4):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, spmv_kernel=1, spmv_value_dict=False)})
    s.generate_poisson7(N)
    hs.append(s)
s0 = hs[0]
xs = [s0.device_array(n) for _ in range(3)]
ys = [s0.device_array(n) for _ in range(3)]
for v in xs: s0.generate_rhs(7, v)
for hi, s in enumerate(hs):
    row = []
    for xi, xv in enumerate(xs):
        for yi, yv in enumerate(ys):
            s.time_spmv(xv, yv, 5)
            row.append(round(s.time_spmv(xv, yv, 40), 4))
    print(json.dumps({"handle": hi, "ms_by_x_then_y": row, "min": min(row), "max": max(row)}), flush=True)
