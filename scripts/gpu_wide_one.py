"""One standalone config of the level-1 SpMV for PMC passes: KERNEL/NT/R via env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from polysolve_amd import HIPSolver
N = int(os.environ.get("N", "192"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=0 + 5))})
s.generate_poisson7(N)
(shape, ptr, col, val) = s.amg_level_matrix(1, 0)
A1 = sp.csr_matrix((val, col, ptr), shape=shape)
del s
t = HIPSolver("Eigen::IdentityPreconditioner")
t.set_parameters({"HIP": dict(spmv_kernel=int(os.environ.get("KERNEL", "1")), spmv_nt=int(os.environ.get("NT", "0")))})
t.factorize(A1)
n, nnz, _ = t.matrix_shape()
x, y = t.to_device(np.random.default_rng(0).uniform(-1, 1, n)), t.device_array(n)
for _ in range(5):
    t.spmv_device(x, y)
t.synchronize()
print("rows", n, "nnz", nnz, "ms", t.time_spmv(x, y, 5))
