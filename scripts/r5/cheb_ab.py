"""A/B of the fused block Chebyshev step (spmv_bsr3_dma<SPMV_CHEB>) on configs[2], level 0: "bsr3_variant" -1 (round 5: the
node's residuals meet by shuffles inside a wave) against 5 (gathers before the barrier as before, lane count at run time: the
LDS exchange behind a third barrier), interleaved, HIP events around the hierarchy's own operators
(psolve_hip_amg_time_level_ops), plus whole solves.  env M (100), REPS (5)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED, amg_cycle_ops
M = int(os.environ.get("M", "100")); REPS = int(os.environ.get("REPS", "5"))
s = HIPSolver("")
s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 3, "amg": dict(AMG_RECOMMENDED)}})
s.generate_elasticity_q1(M); s.synchronize()
n, nnz, _ = s.matrix_shape()
nnzb = int(s.get_param("bsr3_nnzb"))
b, x = s.device_array(n), s.device_array(n)
s.generate_rhs(42, b)
res = {}
for rep in range(REPS):
    for v in (-1, 5):
        s.set_parameters({"HIP": {"bsr3_variant": v}})
        ops = amg_cycle_ops(s, int(s.get_info().get("amg_levels", 4) or 4), block=True, nnzb0=nnzb, max_level=0)[0]["ops"]
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); dt = time.perf_counter() - t
        r = res.setdefault(v, {"cheb_step_us": [], "cheb_frac": [], "residual_us": [], "solve_ms": [], "iterations": None})
        r["cheb_step_us"].append(round(ops["cheb_step"]["us"], 1)); r["cheb_frac"].append(round(ops["cheb_step"]["frac_of_peak"], 4))
        r["residual_us"].append(round(ops["residual"]["us"], 1)); r["solve_ms"].append(round(dt * 1e3, 2))
        r["iterations"] = s.get_info()["num_iterations"]
print(json.dumps({"M": M, "variants": {"-1 (shuffles)": res[-1], "5 (LDS exchange)": res[5]}}))
