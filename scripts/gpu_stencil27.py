"""27-point stencil (scalar Q1 on a structured hex grid) on N^3: the pattern dictionary with several lanes per row
against the SELL copy and the LDS-DMA CSR kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "96"))
T = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1], format="csr")
A = sp.kron(sp.kron(T, T), T, format="csr")
rng = np.random.default_rng(0)
A.data = -rng.uniform(0.5, 1.0, A.nnz)
A = (A + A.T) * 0.5
A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
A = A.tocsr(); A.sort_indices()
n = A.shape[0]
x = rng.uniform(-1, 1, n)
ref = A @ x
for name, prm in (("auto", {}), ("dictionary", dict(spmv_kernel=3)), ("sell", dict(spmv_kernel=2)), ("dma", dict(spmv_kernel=1)), ("pipe", dict(spmv_kernel=0))):
    s = HIPSolver("Eigen::IdentityPreconditioner")
    s.set_parameters({"HIP": prm})
    s.factorize(A)
    dx, dy = s.to_device(x), s.device_array(n)
    ms = min(s.time_spmv(dx, dy, 20) for _ in range(3))
    err = np.abs(dy.download() - ref).max() / np.abs(ref).max()
    print(f"27-point {N}^3 ({n} rows, {A.nnz/n:.1f} per row) {name:10s}: {ms:.4f} ms  R={int(s.get_param('spmv_rows_per_block'))} patterns={int(s.get_param('spmv_patterns'))} sell={int(s.get_param('sell_active'))} err={err:.1e}  {(12*A.nnz+20*n)/ms/1e6:.0f} GB/s on CSR bytes", flush=True)
    del s
