"""Wide-row CSR SpMV: the level-1 operator of the 256^3 hierarchy (2.0 M rows, 31 nnz/row) and the Q1 elasticity
matrix as CSR (81 nnz/row), timed standalone under kernel / cache-policy / row-block variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from polysolve_amd import HIPSolver

N = int(os.environ.get("N", "256"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, renumber=int(os.environ.get("RENUMBER", "1"))))})
s.generate_poisson7(N)
(shape, ptr, col, val) = s.amg_level_matrix(1, 0)
A1 = sp.csr_matrix((val, col, ptr), shape=shape)
del s
deg = np.diff(ptr)
print(f"level 1 of {N}^3: rows={shape[0]} nnz={A1.nnz} avg={A1.nnz/shape[0]:.1f} max={deg.max()} p99={np.percentile(deg,99):.0f}", flush=True)

def run(label, M, cfgs):
    for name, prm in cfgs:
        t = HIPSolver("Eigen::IdentityPreconditioner")
        t.set_parameters({"HIP": prm})
        t.factorize(M)
        n, nnz, _ = t.matrix_shape()
        x, y = t.to_device(np.random.default_rng(0).uniform(-1, 1, n)), t.device_array(n)
        ms = min(t.time_spmv(x, y, 20) for _ in range(3))
        alg = 12 * nnz + 20 * n
        print(f"{label:10s} {name:38s} R={int(t.get_param('spmv_rows_per_block')):3d} {ms:.4f} ms {alg/ms/1e6:6.0f} GB/s alg {alg/ms/1e6/80:.1f} %", flush=True)
        del t

def check(M):
    ref = None
    for k in (1, 2):
        t = HIPSolver("Eigen::IdentityPreconditioner")
        t.set_parameters({"HIP": dict(spmv_kernel=k, spmv_nt=0)})
        t.factorize(M)
        xh = np.random.default_rng(1).uniform(-1, 1, M.shape[0])
        x, y = t.to_device(xh), t.device_array(M.shape[0])
        t.time_spmv(x, y, 1)
        yh = y.download()
        ex = M @ xh
        print(f"  kernel {k}: max |y - scipy| / |y|_inf = {np.abs(yh - ex).max() / np.abs(ex).max():.2e}", flush=True)
        del t

cfgs = [        ("sell nt=0", dict(spmv_kernel=2, spmv_nt=0)),
        ("sell nt=1", dict(spmv_kernel=2, spmv_nt=1)),
        ("pipe auto-R", dict(spmv_kernel=0, spmv_nt=0)),
        ("dma nt=0 auto-R", dict(spmv_kernel=1, spmv_nt=0)),
        ("dma nt=1 auto-R", dict(spmv_kernel=1, spmv_nt=1)),
        ("dma nt=1 R=64", dict(spmv_kernel=1, spmv_nt=1, spmv_rows_per_block=64)),
        ("dma nt=1 R=16", dict(spmv_kernel=1, spmv_nt=1, spmv_rows_per_block=16)),
        ("dma nt=0 R=64", dict(spmv_kernel=1, spmv_nt=0, spmv_rows_per_block=64)),
        ("pipe R=16", dict(spmv_kernel=0, spmv_nt=0, spmv_rows_per_block=16)),
        ("pipe R=64", dict(spmv_kernel=0, spmv_nt=0, spmv_rows_per_block=64))]
check(A1)
run("level1", A1, cfgs[:5] if os.environ.get("ALL", "0") == "0" else cfgs)
import oracle as O
E = O.elasticity_q1(64).to_scipy()
run("elast64", E, [("sell nt=0", dict(spmv_kernel=2, spmv_nt=0)), ("sell nt=1", dict(spmv_kernel=2, spmv_nt=1)), ("dma nt=1 R=16", dict(spmv_kernel=1, spmv_nt=1)), ("pipe", dict(spmv_kernel=0, spmv_nt=0))])
