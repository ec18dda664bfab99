import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
N = 256
s = HIPSolver("")
s.generate_poisson7(N)
n, nnz, _ = s.matrix_shape()
x, y, b = s.device_array(n), s.device_array(n), s.device_array(n)
s.generate_rhs(42, b, x)
byts = 12 * nnz + 20 * n
for m, ch, bpc in [(2, 8192, 4), (2, 8192, 5), (0, 8192, 5), (2, 8192, 6)]:
    s.set_parameters({"HIP": {"spmv_xcd_map": m, "spmv_chunk_rows": ch, "spmv_blocks_per_cu": bpc}})
    ms = min(s.time_spmv(x, y, 20) for _ in range(3))
    print(f"map={m} chunk_rows={ch} bpc={bpc}: {ms:.4f} ms  {byts/ms/1e6:.0f} GB/s", flush=True)
s.set_parameters({"HIP": {"spmv_xcd_map": 2, "spmv_chunk_rows": 8192, "spmv_blocks_per_cu": 5, "profile_spmv": 8}})
for _ in range(2):
    x.upload(np.zeros(n)); t = time.time(); s.solve_device(b, x); dt = time.time() - t
i = s.get_info()
print(f"solve: {dt*1e3:.1f} ms iters={i['num_iterations']} spmv_ms={i['spmv_ms_avg']:.4f} DOF/s={n/dt:.3e}")
