import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from polysolve_amd import HIPSolver
step = sys.argv[1]
if step == "main":
    out, n, nnz = bench.elasticity_leg(HIPSolver, 12, 0, 2); print("main ok", out["iterations"])
elif step == "direct":
    out, n, nnz = bench.elasticity_leg(HIPSolver, 12, 0, 2, {"direct_coarse": True}); print("direct ok", out["iterations"])
elif step == "perm":
    out, n, nnz = bench.elasticity_leg(HIPSolver, 12, 1, 2); print("perm ok", out["iterations"])
elif step == "perm0":
    out, n, nnz = bench.elasticity_leg(HIPSolver, 12, 1, 0); print("perm0 ok", out["iterations"])
