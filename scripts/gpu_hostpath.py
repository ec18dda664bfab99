"""Host entry points (the actual drop-in boundary): factorize / solve wall times incl. PCIe, and
per-iteration cost of small systems (launch-bound regime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from polysolve_amd import Solver

for N in (32, 64, 128, 200):
    A = O.poisson7(N); M = A.to_scipy()
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    s = Solver.create("HIP", "")
    t = time.time(); s.analyze_pattern(M, A.n); ta = time.time() - t
    t = time.time(); s.factorize(M); tf = time.time() - t
    t = time.time(); s.factorize(M); tf2 = time.time() - t
    x = np.zeros(A.n); s.solve(b, x)
    x = np.zeros(A.n); t = time.time(); s.solve(b, x); ts = time.time() - t
    i = s.get_info()
    gb = (12 * A.nnz + 4 * A.n) / 1e9
    print(f"N={N} n={A.n}: analyze {ta*1e3:.1f} ms factorize {tf*1e3:.1f} / {tf2*1e3:.1f} ms ({gb/tf2:.1f} GB/s H2D) "
          f"solve(host b,x) {ts*1e3:.2f} ms device-part {i['time_solve_device']*1e3:.2f} ms iters={i['num_iterations']} "
          f"-> {ts*1e6/i['num_iterations']:.1f} us/iter  DOF/s={A.n/ts:.3e}", flush=True)
