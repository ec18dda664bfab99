"""Debugging aid for amg_sweep.hip: one single-level gauss_seidel / ilu0 application on a small matrix, against the oracle."""
import os, sys, json
os.environ.setdefault("OMP_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
import oracle
from polysolve_amd import Solver
case, rt = sys.argv[1], sys.argv[2]
bs = 1
if case == "poisson": A = oracle.poisson7(14, 12, 13)
elif case == "coarse":
    ref = oracle.AMG(oracle.poisson7(14, 12, 13), coarse_enough=60)
    A = ref.level(1)
elif case == "elast": A, bs = oracle.elasticity_q1(6), 3
M = sp.csr_matrix(A.to_scipy()); M.sort_indices()
print(case, rt, "n", A.n, "nnz/row", M.nnz / A.n, flush=True)
s = Solver.create("HIP", "")
s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-9, max_iter=300, block_size=bs, reorder=0, amg={"relax_type": rt, "class": "relaxation"})})
s.analyze_pattern(M, M.shape[0]); s.factorize(M)
print("factorized", flush=True)
r = oracle.splitmix_vector(A.n, 11)
z = s.device_array(A.n)
s.precond_apply_device(s.to_device(r), z)
zo = oracle.AMG(oracle.CSR.from_scipy(M), relax_type=rt, precond_class="relaxation", block_size=bs).apply(r)
print("max diff", np.abs(z.download() - zo).max(), "equal", np.array_equal(z.download(), zo), flush=True)
