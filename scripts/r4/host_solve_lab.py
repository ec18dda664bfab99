"""what the host contract's solve(b, x) adds to the device solve: pinned staging ("lab.stage_kb" >= the vector) against direct copies
from the caller's pageable arrays; Jacobi-PCG with max_iter 1 so that the transfers are what is timed."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import Solver
import oracle as O
for N in [int(v) for v in os.environ.get("NS", "64,100,128,160,200,256").split(",")]:
    A = O.poisson7(N).to_scipy().tocsc()
    n = A.shape[0]
    b = np.ones(n); x = np.zeros(n)
    row = {}
    for stage in (0, 1 << 30, 0, 1 << 30):
        s = Solver.create("HIP", "")
        s.set_parameters({"HIP": {"max_iter": 1, "tolerance": 1e-30, "lab.stage_kb": stage}})
        s.analyze_pattern(A, n); s.factorize(A)
        ts = []
        for _ in range(5):
            x[:] = 0.0
            t = time.perf_counter(); s.solve(b, x); ts.append(time.perf_counter() - t)
        row.setdefault("staged" if stage else "direct", []).append(round(min(ts) * 1e3, 3))
        del s
    print(N, f"{8*n/2**20:.0f} MiB per vector", row, flush=True)
Solver.create("HIP", "").set_parameters({"HIP": {"lab.stage_kb": 256}})
