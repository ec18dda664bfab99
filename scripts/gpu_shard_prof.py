"""One 2.1 M-row shard (256 x 256 x 32), one-rank RCCL communicator, single-reduction loop: for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polysolve_amd import HIPSolver
s = HIPSolver("")
s.set_parameters({"HIP": dict(tolerance=1e-8, max_iter=300)})
s.comm_init(0, 1, HIPSolver.comm_unique_id())
s.generate_poisson7(256, 256, 32)
n = s.matrix_shape()[0]
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
for _ in range(3):
    s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
    t = time.perf_counter(); s.solve_device(b, x); dt = time.perf_counter() - t
print(f"{dt*1e6/s.get_info()['num_iterations']:.1f} us per iteration, {s.get_info()['num_iterations']} iterations")
