"""256^3 (N) Poisson AMG-PCG under rocprofv3: one setup, three solves of AMG_RECOMMENDED, and the launch plan of one
iteration (scripts/evidence/amg_by_level.py) from the shapes of the hierarchy that was built.  Env: N (256), PLAN, AMG (json overrides)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "scripts", "evidence"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED, BoxSampler, box_static
import amg_by_level as ab
N = int(os.environ.get("N", "256"))
amg = dict(AMG_RECOMMENDED, **json.loads(os.environ.get("AMG", "{}")))
top = json.loads(os.environ.get("HIP", "{}"))
s = HIPSolver("")
s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=amg, **top)})
s.generate_poisson7(N, N, N); s.synchronize()
n = s.matrix_shape()[0]
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
with BoxSampler() as box:
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.time(); s.solve_device(b, x); s.synchronize(); dt = time.time() - t
info = s.get_info()
print(f"solve {dt*1e3:.1f} ms its={info['num_iterations']}")
nl = int(info["amg_levels"])
levels = []
has_pat = s.get_param("spmv_patterns") > 0
for l in range(nl):
    rows, nnz, _ = s.amg_level_info(l)
    csr = lambda shape: dict(fmt="csr", rows=int(shape[0]), cols=int(shape[1]), nnz=int(shape[2]))
    A = dict(fmt=("kinds" if s.get_param("spmv_row_kinds") > 0 else "pat") if (l == 0 and has_pat) else "csr", rows=rows, cols=rows, nnz=nnz)
    L = dict(n=rows, A=A, P=None, R=None, block=False, fused=True)
    if l + 1 < nl:
        L["P"] = csr(s.amg_level_matrix_shape(l, 1))
        L["R"] = csr(s.amg_level_matrix_shape(l, 2))
    levels.append(L)
plan = ab.iteration_plan(levels, dict(ncycle=amg["ncycle"], npre=1, npost=1, cheb_degree=amg["cheb_degree"]))
json.dump(dict(workload=f"Poisson {N}^3 AMG-PCG", amg=amg, levels=levels, iterations=int(info["num_iterations"]),
               solve_ms_under_rocprof=dt * 1e3, box=dict(box_static(), during_solves=box.summary(), probe=s.box_probe()), plan=plan),
          open(os.environ.get("PLAN", os.path.join(ROOT, "gpurun_out", "r04_poisson_plan.json")), "w"), indent=1)
