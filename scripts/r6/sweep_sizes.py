"""Round 6: one gauss_seidel sweep of level 0 against the grid size (is a line a whole number of 64-row tickets?)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
for N in [int(v) for v in os.environ.get("SIZES", "64,96,100,128,130,160,192,200").split(",")]:
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 50, "precond": "amg", "amg": {"relax_type": "gauss_seidel", "class": "relaxation"}}})
    s.generate_poisson7(N); s.synchronize()
    ops = s.amg_time_level_ops(0, 2)
    print(json.dumps({"N": N, "rows": N ** 3, "sweep_us": round(ops["cheb_first_us"] , 1), "ns_per_row": round(ops["cheb_first_us"] * 1e3 / N ** 3, 2)}), flush=True)
    del s
