"""SpMV time vs row-block schedule (round-robin vs chunks dealt to XCDs) for several grid sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver
for N in [int(v) for v in os.environ.get("SIZES", "160,200,216,256").split(",")]:
    s = HIPSolver("")
    s.generate_poisson7(N)
    n, nnz, _ = s.matrix_shape()
    x, y, b = s.device_array(n), s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b, x)
    byts = 12 * nnz + 20 * n
    plane8 = max(256, (N * N // 8 + 255) // 256 * 256)
    for m, ch in [(0, 8192), (2, 2048), (2, 4096), (2, 8192), (2, 16384), (2, plane8), (2, 2 * plane8)]:
        s.set_parameters({"HIP": {"spmv_xcd_map": m, "spmv_chunk_rows": ch}})
        ms = min(s.time_spmv(x, y, 20) for _ in range(3))
        print(f"N={N} map={m} chunk_rows={ch}: {ms:.4f} ms  {byts/ms/1e6:.0f} GB/s", flush=True)
    del s
