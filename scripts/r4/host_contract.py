"""bench.py's host_contract block on its own (the HOST contract at config size: 256^3 Jacobi-PCG, Q1 elasticity M = 100 AMG-PCG)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from polysolve_amd import HIPSolver
from bench import host_contract_block, box_static
out = host_contract_block(HIPSolver, np, int(os.environ.get("N", "256")), int(os.environ.get("M", "100")))
out["box"] = box_static()
for k in ("poisson", "elasticity"):
    print(k, json.dumps(out[k]))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_host_contract.json"), "w"), indent=1)
