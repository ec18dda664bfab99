"""The shards' loop (single-reduction recurrences, RCCL communicator of ONE rank) on a 256^3 slab: what an iteration
of a weak-scaling shard costs without the wire, next to the single-GPU loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "256"))
for dist in (0, 1, 2):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=20000, dist_single_reduction=(dist != 2))})
    if dist:
        s.comm_init(0, 1, HIPSolver.comm_unique_id())
    s.generate_poisson7(N)
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    print(f"N={N} dist={dist}: {best*1e3:.1f} ms, {i['num_iterations']} iterations, {best*1e3/i['num_iterations']:.4f} ms/it, true {i['true_residual']:.2e}", flush=True)
    del s
