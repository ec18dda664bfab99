import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import mesh_utils as mu
from polysolve_amd import Solver
P, T, bd = mu.tet_mesh(15, seed=7)
K, _ = mu.renumber_nodes(mu.p1_elasticity(P, T, bd), 3, seed=8)
n = K.shape[0]
for rev in (False, True):
    s = Solver.create({"solver": "HIP", "HIP": {"precond": "amg", "block_size": 3, "tolerance": 1e-9, "reorder_min_rows": 0, "reorder_reverse": rev,
                                                "amg": {"coarse_enough": 300, "cheb_degree": 3, "cheb_power_iters": 20}}})
    s._set("lab.verbose", 1)
    for k in range(4):
        H = (K + (0.05 * k) * sp.diags(K.diagonal())).tocsc()
        s.analyze_pattern(H, n); s.factorize(H)
        print(rev, k, "setups", s.get_param("stats.amg_setups"), "refreshes", s.get_param("stats.amg_refreshes"), "levels", s.get_info()["amg_levels"], [s.amg_level_info(l)[:2] for l in range(int(s.get_info()["amg_levels"]))], flush=True)
