"""Round 3: an UNSTRUCTURED mesh (Delaunay tetrahedra of a jittered point cloud, tests/mesh_utils.py), P1 elasticity
with 3 x 3 node blocks and P1 Laplace, through the host contract: the mesh generator's numbering (lattice order of the
points) and a random numbering of the nodes, "reorder" 0 against the default (auto), Jacobi and AMG."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mesh_utils as mu
from polysolve_amd import Solver
from bench import AMG_RECOMMENDED
m = int(sys.argv[1]) if len(sys.argv) > 1 else 56
t = time.time(); P, T, bd = mu.tet_mesh(m, seed=1); print(f"mesh: {len(P)} nodes, {len(T)} tetrahedra, {time.time() - t:.1f} s", flush=True)
out = {"nodes": int(len(P)), "tets": int(len(T))}
for kind, b3 in (("elasticity", 3), ("laplace", 1)):
    t = time.time(); K0 = mu.p1_elasticity(P, T, bd) if b3 == 3 else mu.p1_laplace(P, T, bd)
    print(f"{kind}: n = {K0.shape[0]}, nnz = {K0.nnz} ({K0.nnz / K0.shape[0]:.1f} per row), assembled in {time.time() - t:.1f} s", flush=True)
    for numbering in ("generator", "random"):
        K = K0 if numbering == "generator" else mu.renumber_nodes(K0, b3, seed=2)[0]
        K = K.tocsc()
        n = K.shape[0]
        xs = np.random.default_rng(0).uniform(-1, 1, n); b = K @ xs
        for precond in ("jacobi", "amg"):
            for reorder in (0, 2):
                s = Solver.create({"solver": "HIP", "HIP": {"tolerance": 1e-8, "max_iter": 20000, "block_size": b3, "precond": precond,
                                                            "reorder": reorder, "reorder_min_rows": 0, "profile_spmv": 4, "amg": dict(AMG_RECOMMENDED)}})
                s.analyze_pattern(K, n)
                t = time.time(); s.factorize(K); t1 = time.time() - t
                t = time.time(); s.factorize(K); t2 = time.time() - t
                x = np.zeros(n); s.solve(b, x)
                x = np.zeros(n); t = time.time(); s.solve(b, x); ts = time.time() - t
                i = s.get_info(); st = s.info_struct()
                r = {"factorize_first_s": t1, "factorize_again_s": t2, "solve_s": ts, "device_solve_s": i["time_solve_device"] if "time_solve_device" in i else None,
                     "iterations": int(i["num_iterations"]), "true_residual": i["true_residual"], "err": float(np.abs(x - xs).max()),
                     "spmv_ms": st.spmv_ms_avg, "reordered": bool(s.get_param("reorder.active")), "spread_before": s.get_param("reorder.spread_before"),
                     "spread_after": s.get_param("reorder.spread_after"), "bfs_levels": s.get_param("reorder.levels"), "amg_levels": int(i["amg_levels"])}
                out[f"{kind}/{numbering}/{precond}/reorder{reorder}"] = r
                print(kind, numbering, precond, reorder, json.dumps(r), flush=True)
                del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_tetmesh.json"), "w"), indent=1)
