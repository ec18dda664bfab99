"""does a product that sweeps the operator from the other end find the previous product's tail in the Infinity Cache?
time_spmv with "lab.alternate" (consecutive launches alternate the sweep direction) x "spmv_nt" (-1 auto, 0 plain, 1 non-temporal)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
out = []
cases = [("poisson", 256, 3), ("poisson", 256, 1), ("poisson", 216, 3), ("poisson", 160, 3), ("elast", 100, -1), ("elast", 64, -1)]
for kind, N, kern in cases:
    for nt in (-1, 0, 1):
        row = {}
        for alt in (0, 1, 0, 1):
            s = HIPSolver("")
            top = {"tolerance": 1e-8, "spmv_nt": nt, "lab.alternate": alt}
            if kern >= 0: top["spmv_kernel"] = kern
            if kind == "elast": top["block_size"] = 3
            s.set_parameters({"HIP": top})
            if kind == "poisson": s.generate_poisson7(N, N, N)
            else: s.generate_elasticity_q1(N)
            s.synchronize()
            n, nnz, _ = s.matrix_shape()
            x, y = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, x)
            ms = min(s.time_spmv(x, y, reps=10) for _ in range(3))
            row.setdefault(f"alt{alt}", []).append(round(ms, 4))
            x.free(); y.free(); del s
        print(kind, N, "kernel", kern, "nt", nt, row, flush=True)
        out.append(dict(kind=kind, N=N, kernel=kern, nt=nt, ms=row))
HIPSolver("").set_parameters({"HIP": {"lab.alternate": 0}})
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_alt_lab.json"), "w"), indent=1)
