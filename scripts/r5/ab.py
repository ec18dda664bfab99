"""Round-5 A/B on one box: first factorize (a second full setup on one handle), numeric refresh and solve of the recommended
configuration under a list of option sets.  env: KIND=poisson|elast, N (grid / M), SETS = ';'-separated JSON objects merged
into /HIP/amg (default: the options this round added), REPS."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
KIND = os.environ.get("KIND", "poisson"); N = int(os.environ.get("N", "216")); REPS = int(os.environ.get("REPS", "4"))
SETS = [json.loads(x) for x in os.environ.get("SETS", '{"product_plan":0,"overlap_smoothers":0};{"product_plan":1,"overlap_smoothers":0};{"aggregation":"parallel","product_plan":1}').split(";")]
out = []
for extra in SETS:
    s = HIPSolver("")
    amg = dict(AMG_RECOMMENDED); amg.update(extra)
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 1 if KIND == "poisson" else 3, "amg": amg}})
    gen = (lambda: s.generate_poisson7(N)) if KIND == "poisson" else (lambda: s.generate_elasticity_q1(N))
    gen(); s.synchronize()                      # warm-up: code objects, first-touch allocations
    s.set_parameters({"HIP": {"amg": {"reuse": False}}})
    setups = []
    for _ in range(REPS):
        t = time.perf_counter(); gen(); s.synchronize(); setups.append(time.perf_counter() - t)
    s.set_parameters({"HIP": {"amg": {"reuse": True}}})
    gen(); s.synchronize()                      # (a full setup once more: it is this one that keeps its patterns for reuse)
    refreshes = []
    for k in range(REPS + 2):
        if k == REPS + 1 and os.environ.get("LAPS"): os.environ["PSOLVE_TIMING"] = "1"
        t = time.perf_counter(); gen(); s.synchronize(); refreshes.append(time.perf_counter() - t)
        assert s.get_param("amg.last_setup_reused") == 1
    os.environ.pop("PSOLVE_TIMING", None)
    n, nnz, _ = s.matrix_shape()
    b = s.device_array(n); x = s.device_array(n)
    s.generate_rhs(42, b)
    solves = []
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); solves.append(time.perf_counter() - t)
    info = s.get_info()
    rec = dict(kind=KIND, N=N, options=extra, setup_s=[round(v, 4) for v in setups], refresh_s=[round(v, 4) for v in refreshes],
               solve_s=[round(v, 4) for v in solves], iterations=info.get("num_iterations"), levels=info.get("amg_levels"),
               level_rows=[s.amg_level_info(l)[0] for l in range(info.get("amg_levels", 0))],
               true_residual=info.get("true_residual"), plan_levels=s.get_param("amg.levels_with_product_plans"),
               plan_mbytes=round(s.get_param("amg.product_plan_mbytes"), 1), device_mbytes=round(s.get_param("stats.device_bytes") / 2**20))
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del s
