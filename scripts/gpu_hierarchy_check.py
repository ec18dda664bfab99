"""Device-built hierarchy (aggregation rounds, row-set kernels, numeric kernels) vs the all-host construction at
sizes where the dependency rounds run into the hundreds: every A_l, P_l, R_l must be identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import oracle as O
from polysolve_amd import Solver, HostHierarchy

def check(name, M, bs=1, **amg):
    M = sp.csr_matrix(M); M.sort_indices()
    n = M.shape[0]
    t = time.time()
    host = HostHierarchy(n, M.indptr, M.indices, M.data, max_levels=amg.get("max_levels", 6),
                         coarse_enough=amg.get("coarse_enough", 3000), eps_strong=amg.get("eps_strong", 0.0), block_size=bs)
    th = time.time() - t
    s = Solver.create("HIP", "")
    s.set_parameters({"HIP": {"precond": "amg", "block_size": bs, "amg": dict(amg, aggregation_min_rows=0, cheb_power_iters=3)}})
    t = time.time(); s.factorize(M); td = time.time() - t
    assert s.get_info()["amg_levels"] == host.num_levels
    for l in range(host.num_levels):
        for what, w in (("A", 0), ("P", 1), ("R", 2)):
            h = host.level(l, what)
            if h is None: continue
            shape, ptr, col, val = s.amg_level_matrix(l, w)
            assert shape == (h[0], h[1]) and np.array_equal(ptr, h[2]) and np.array_equal(col, h[3]), (name, l, what)
            assert np.array_equal(val, h[4]), (name, l, what)
    print(f"{name}: {n} rows, {host.num_levels} levels identical; levels aggregated on the device: "
          f"{int(s.get_param('amg.levels_aggregated_on_device'))}; host build {th:.2f} s, device factorize (incl. upload) {td:.2f} s", flush=True)

check("poisson 128^3", O.poisson7(128).to_scipy())
check("poisson 100x80x90 eps 0.05", O.poisson7(100, 80, 90).to_scipy() + sp.diags(np.linspace(0, 2, 720000)), eps_strong=0.05)
check("elasticity M=40 scalar", O.elasticity_q1(40).to_scipy())
check("elasticity M=40 block 3", O.elasticity_q1(40).to_scipy(), bs=3)
