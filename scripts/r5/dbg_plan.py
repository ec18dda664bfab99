import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import oracle
from polysolve_amd import Solver, HostHierarchy
os.environ["PSOLVE_TIMING"] = "1"
M0 = sp.csr_matrix(oracle.poisson7(22, 19, 20).to_scipy()); M0.sort_indices()
n = M0.shape[0]
for plan in (0, 1):
    s = Solver.create("HIP", "")
    s.set_parameters({"HIP": {"lab.plan_verbose": 2}})
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-10, amg=dict(coarse_enough=40, max_levels=5, aggregation_min_rows=0, ncycle=1, cheb_degree=2, cheb_power_iters=5, product_plan=plan))})
    s.analyze_pattern(M0, n); s.factorize(M0)
    rng = np.random.default_rng(5)
    d = (1.0 + 0.3 * rng.uniform(0, 1, n))
    Mk = M0.copy(); rows = np.repeat(np.arange(n), np.diff(M0.indptr)); Mk.data = M0.data * d[rows] * d[M0.indices]
    s.factorize(Mk)
    print("plan", plan, "reused", s.get_param("amg.last_setup_reused"), "plan levels", s.get_param("amg.levels_with_product_plans"), flush=True)
    host = HostHierarchy(n, Mk.indptr, Mk.indices, Mk.data, max_levels=5, coarse_enough=40)
    for l in range(host.num_levels):
        for what, w in (("A", 0), ("P", 1), ("R", 2)):
            h = host.level(l, what)
            if h is None: continue
            shape, ptr, col, val = s.amg_level_matrix(l, w)
            print(l, what, shape, "ptr", np.array_equal(ptr, h[2]), "col", np.array_equal(col, h[3]), "val", np.array_equal(val, h[4]), np.abs(val - h[4]).max() if val.shape == h[4].shape else None, flush=True)
