"""configs[2] under rocprofv3: one setup, three solves of AMG_RECOMMENDED -- and the launch plan of one iteration
(scripts/evidence/amg_by_level.py: iteration_plan) written next to the trace, from the shapes of the hierarchy that was built.
Env: M (100), BL ("amg.block_levels", 1), PLAN (output path of the plan)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "scripts", "evidence"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED, BoxSampler, box_static
import amg_by_level as ab
M = int(os.environ.get("M", "100")); BL = int(os.environ.get("BL", "1"))
s = HIPSolver("")
amg = dict(AMG_RECOMMENDED, block_levels=bool(BL))
s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=amg)})
s.generate_elasticity_q1(M); s.synchronize()
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
for l in range(nl):
    rows, nnz, _ = s.amg_level_info(l)
    blk = lambda shape: dict(fmt="bsr3", rows=int(shape[0]), cols=int(shape[1]), nnz=int(shape[2]) // 9)
    csr = lambda shape: dict(fmt="csr", rows=int(shape[0]), cols=int(shape[1]), nnz=int(shape[2]))
    if l == 0:
        A = dict(fmt="bkinds" if s.get_param("bsr3_row_kinds") > 0 else "bsr3", rows=rows, cols=rows, nnz=int(s.get_param("bsr3_nnzb")))
    else:
        A = (blk if BL else csr)((rows, rows, nnz))
    L = dict(n=rows, A=A, P=None, R=None, block=True, fused=bool(BL))
    if l + 1 < nl:
        L["P"] = (blk if BL else csr)(s.amg_level_matrix_shape(l, 1))
        L["R"] = (blk if BL else csr)(s.amg_level_matrix_shape(l, 2))
    levels.append(L)
plan = ab.iteration_plan(levels, dict(ncycle=amg["ncycle"], npre=1, npost=1, cheb_degree=amg["cheb_degree"]))
json.dump(dict(workload=f"Q1 elasticity M={M} block-3 AMG-PCG", amg=amg, levels=levels, iterations=int(info["num_iterations"]),
               solve_ms_under_rocprof=dt * 1e3, box=dict(box_static(), during_solves=box.summary()), plan=plan),
          open(os.environ.get("PLAN", os.path.join(ROOT, "gpurun_out", "r04_elast_plan.json")), "w"), indent=1)
