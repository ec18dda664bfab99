import sys; sys.path.insert(0, "/root/repo")
import numpy as np, oracle as O
from polysolve_amd import Solver
for name, A in (("poisson24", O.poisson7(24).to_scipy().tocsr()), ("elast8", O.elasticity_q1(8).to_scipy().tocsr())):
    s = Solver.create({"solver": "HIP", "HIP": {"precond": "amg", "amg": {"coarse_enough": 100, "aggregation_min_rows": 0}}})
    s.factorize(A.tocsc())
    out = []
    rng = np.random.default_rng(0)
    for f in (1.5, 3.0, 0.7):
        B = A.copy(); B.data *= f
        s.factorize(B.tocsc()); out.append(s.get_param("amg.last_setup_reused"))
    # random symmetric perturbation of the values (Newton-like)
    U = __import__("scipy.sparse", fromlist=["x"]).triu(A, k=1).tocoo()
    U.data = U.data * rng.uniform(0.5, 1.5, U.nnz)
    D = __import__("scipy.sparse", fromlist=["x"]).diags(A.diagonal() * 2)
    B = (U + U.T + D).tocsr(); B.sort_indices()
    s2 = Solver.create({"solver": "HIP", "HIP": {"precond": "amg", "amg": {"coarse_enough": 100, "aggregation_min_rows": 0}}})
    s2.factorize(B.tocsc())
    B2 = B.copy(); U2 = U.copy(); U2.data = U.data * rng.uniform(0.9, 1.1, U.nnz); B2 = (U2 + U2.T + D).tocsr(); B2.sort_indices()
    ok = B2.nnz == B.nnz
    s2.factorize(B2.tocsc()); out.append(("random", ok, s2.get_param("amg.last_setup_reused")))
    print(name, out)
