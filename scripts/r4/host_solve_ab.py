"""configs[2] through the host contract's solve(b, x): staging limit 256 KiB (round 4) against 32 MiB (until then), one handle each, interleaved"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
M = int(os.environ.get("M", "100"))
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED)}})
s.generate_elasticity_q1(M); s.synchronize()
n = s.matrix_shape()[0]
bd = s.device_array(n); s.generate_rhs(42, bd); b = bd.download(); x = np.zeros(n)
for kb in (32768, 256, 32768, 256, 32768, 256):
    s.set_parameters({"HIP": {"lab.stage_kb": kb}})
    ts = []
    for _ in range(3):
        x[:] = 0.0
        t = time.perf_counter(); s.solve(b, x); ts.append(time.perf_counter() - t)
    print(f"stage_kb {kb:6d}: host solve {min(ts)*1e3:.2f} ms, device part {s.get_info().get('time_solve_device', float('nan'))*1e3:.2f} ms, its {s.get_info()['num_iterations']}", flush=True)
s.set_parameters({"HIP": {"lab.stage_kb": 256}})
