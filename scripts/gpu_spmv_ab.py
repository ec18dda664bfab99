"""A/B of the SpMV kernels through the C ABI: standalone time_spmv on Poisson grids and the elasticity matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from polysolve_amd import HIPSolver

def run(label, gen, cfgs):
    for name, prm in cfgs:
        s = HIPSolver("")
        s.set_parameters({"HIP": prm})
        gen(s)
        n, nnz, _ = s.matrix_shape()
        x, y = s.device_array(n), s.device_array(n)
        s.generate_rhs(7, x)
        ms = min(s.time_spmv(x, y, 20) for _ in range(3))
        alg = 12 * nnz + 20 * n
        print(f"{label:16s} {name:34s} R={int(s.get_param('spmv_rows_per_block')):3d} grid={int(s.get_param('spmv_grid')):5d} "
              f"{ms:.4f} ms  {alg / ms / 1e6:7.0f} GB/s alg  {alg / ms / 1e6 / 80:.1f} %", flush=True)
        del s

cfgs = [("pattern dict nt=auto 6/cu", dict(spmv_kernel=3)),
        ("pattern dict nt=1 8/cu", dict(spmv_kernel=3, spmv_nt=1, spmv_blocks_per_cu=8)),
        ("pattern dict nt=1 9/cu", dict(spmv_kernel=3, spmv_nt=1, spmv_blocks_per_cu=9)),
        ("pattern dict nt=0 6/cu", dict(spmv_kernel=3, spmv_nt=0)),
        ("pipe (r1) 5/cu", dict(spmv_kernel=0, spmv_blocks_per_cu=5)),
        ("dma nt=auto 6/cu", dict(spmv_kernel=1)),
        ("dma nt=0 6/cu", dict(spmv_kernel=1, spmv_nt=0)),
        ("dma nt=1 6/cu", dict(spmv_kernel=1, spmv_nt=1)),
        ("dma nt=1 5/cu", dict(spmv_kernel=1, spmv_nt=1, spmv_blocks_per_cu=5))]
for N in (int(a) for a in (sys.argv[1:] or ["256"])):
    run(f"poisson {N}^3", lambda s: s.generate_poisson7(N), cfgs)
if os.environ.get("ELAST", "1") == "1":
    run("elasticity M=64", lambda s: s.generate_elasticity_q1(64), cfgs[4:8])
