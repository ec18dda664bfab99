"""An unstructured tetrahedral mesh (tests/mesh_utils.py), P1 elasticity (block 3) and P1 Laplace, nodes in the generator's
and in a random order, AMG-PCG through the host contract: "reorder_reverse" 0 against 1 (forced renumbering, so the
generator's order is renumbered too), and the golden fixture's mesh (tests/golden/reorder_tets.npz)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
import mesh_utils as mu
from polysolve_amd import Solver
from bench import AMG_RECOMMENDED
m = int(sys.argv[1]) if len(sys.argv) > 1 else 48
P, T, bd = mu.tet_mesh(m, seed=1)
out = {"nodes": int(len(P)), "tets": int(len(T))}
systems = []
for kind, b3 in (("elasticity", 3), ("laplace", 1)):
    K0 = mu.p1_elasticity(P, T, bd) if b3 == 3 else mu.p1_laplace(P, T, bd)
    for numbering in ("generator", "random"):
        K = K0 if numbering == "generator" else mu.renumber_nodes(K0, b3, seed=2)[0]
        systems.append((f"{kind}/{numbering}", K.tocsc(), b3))
g = np.load(os.path.join(ROOT, "tests", "golden", "reorder_tets.npz"))
systems.append(("golden reorder_tets", sp.csr_matrix((g["val"], g["col"], g["rowptr"])).tocsc(), 1))
for name, K, b3 in systems:
    n = K.shape[0]
    xs = np.random.default_rng(0).uniform(-1, 1, n); b = K @ xs
    for reorder, rev in ((0, 0), (1, 0), (1, 1)):
        s = Solver.create({"solver": "HIP", "HIP": {"tolerance": 1e-8, "max_iter": 20000, "block_size": b3, "precond": "amg", "reorder": reorder,
                                                    "reorder_reverse": bool(rev), "reorder_min_rows": 0, "amg": dict(AMG_RECOMMENDED, coarse_enough=3000 if n > 20000 else 100)}})
        s.analyze_pattern(K, n); s.factorize(K)
        x = np.zeros(n); s.solve(b, x)
        x = np.zeros(n); t = time.time(); s.solve(b, x); ts = time.time() - t
        i = s.get_info()
        r = dict(n=n, iterations=int(i["num_iterations"]), solve_ms=ts * 1e3, levels=int(i["amg_levels"]), reordered=bool(s.get_param("reorder.active")))
        out[f"{name}/reorder{reorder}/reverse{rev}"] = r
        print(name, reorder, rev, json.dumps(r), flush=True)
        del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_tetmesh_reverse.json"), "w"), indent=1)
