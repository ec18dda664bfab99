"""Round 6: hunt for an intermittent NaN -- fresh handles, AMG-PCG on small Poisson grids, recycled blocks poisoned (run with
PSOLVE_ALLOC_CACHE_POISON=1); every deviation from the first result of a size is printed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import math
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
REPS = int(os.environ.get("REPS", "150"))
ref = {}
bad = 0
keep = []
for i in range(REPS):
    N = (48, 40, 56, 33)[i % 4]
    s = HIPSolver("")
    amg = dict(AMG_RECOMMENDED) if i % 2 == 0 else {}
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 500, "precond": "amg", "amg": amg}})
    out = []
    for k in range(3):   # first setup, then two refreshes
        s.generate_poisson7(N)
        n = N ** 3
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.solve_device(b, x); s.synchronize()
        inf = s.get_info()
        out.append((int(inf["num_iterations"]), inf["true_residual"]))
        b.free(); x.free()
    key = (N, i % 2)
    if key not in ref: ref[key] = out
    ok = all(math.isfinite(r) and r < 2e-8 for _, r in out) and [a for a, _ in out] == [a for a, _ in ref[key]]
    if not ok:
        bad += 1
        print(json.dumps({"rep": i, "N": N, "recommended": i % 2 == 0, "got": out, "first": ref[key]}), flush=True)
    if i % 3 == 0: keep.append(s)   # some handles stay alive (their caches hold blocks)
    if len(keep) > 4: keep.pop(0)
print(json.dumps({"reps": REPS, "deviations": bad}))
