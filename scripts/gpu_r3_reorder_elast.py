"""Round 3: configs[2] (Q1 elasticity, block-3 AMG-PCG) with the nodes renumbered pseudo-randomly: the backend's default
(renumbered at factorize on the node graph) against the caller's numbering, next to the grid numbering."""
import sys, json
sys.path.insert(0, ".")
from polysolve_amd import HIPSolver
import bench
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = {}
for name, mode, reorder in (("grid", 0, 2), ("random_nodes/default", 1, 2), ("random_nodes/caller_numbering", 1, 0), ("grid/forced", 0, 1)):
    r, n, nnz = bench.elasticity_leg(HIPSolver, M, mode, reorder)
    r.pop("amg"); r.pop("levels")
    out[name] = r
    print(name, json.dumps(r), flush=True)
json.dump(out, open("gpurun_out/r03_reorder_elast.json", "w"), indent=1)
