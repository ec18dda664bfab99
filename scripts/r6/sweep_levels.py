"""Round 6: time of one gauss_seidel / ilu0 application per level of a hierarchy (psolve_hip_amg_time_level_ops)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
N = int(os.environ.get("N", "216")); rt = os.environ.get("RELAX", "gauss_seidel")
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 50, "precond": "amg", "amg": dict(AMG_RECOMMENDED, relax_type=rt)}})
s.generate_poisson7(N); s.synchronize()
for l in range(int(s.get_info()["amg_levels"])):
    rows, nnz, _ = s.amg_level_info(l)
    ops = s.amg_time_level_ops(l, 2)
    print(json.dumps({"N": N, "relax": rt, "level": l, "rows": rows, "nnz_per_row": round(nnz / rows, 1), "two_applications_us_each": round(ops["cheb_step_us"], 1),
                      "first_application_us": round(ops["cheb_first_us"], 1), "residual_us": round(ops["residual_us"], 1)}), flush=True)
