"""What an iteration of a strong-scaling shard costs without the wire: a 256 x 256 x (256 / W) slab with a one-rank
RCCL communicator (no neighbours, all-reduce of one rank), both recurrences, next to the single-GPU loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
for W in (8, 4, 2):
    nz = 256 // W
    for mode in ("single-gpu loop", "shard loop, one reduction", "shard loop, two reductions"):
        s = HIPSolver("")
        s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=300, dist_single_reduction=(mode != "shard loop, two reductions"))})
        if mode != "single-gpu loop":
            s.comm_init(0, 1, HIPSolver.comm_unique_id())
        s.generate_poisson7(256, 256, nz)
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        best = 1e9
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
        i = s.get_info()
        print(f"256x256x{nz} ({n/1e6:.1f} M rows) {mode:28s}: {best*1e6/i['num_iterations']:7.1f} us per iteration ({i['num_iterations']} iterations)", flush=True)
        del s
