"""precond = "ic": natural ordering against Eigen's default AMD ordering -- dependency levels of the
triangular solves, factorization time (host), apply time, IC-PCG against Jacobi-PCG, N^3 Poisson through the host contract."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oracle as O
from polysolve_amd import Solver
out = []
for N in [int(v) for v in os.environ.get("NS", "64,128").split(",")]:
    A = O.poisson7(N); M = A.to_scipy().tocsc()
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    for precond, ordering in (("jacobi", 0), ("ic", 0), ("ic", 1)):
        s = Solver.create({"solver": "HIP", "HIP": {"precond": precond, "tolerance": 1e-8, "max_iter": 5000, "ic": {"ordering": ordering}}})
        s.analyze_pattern(M, A.n)
        t = time.time(); s.factorize(M); tf = time.time() - t
        x = np.zeros(A.n); s.solve(b, x)
        x = np.zeros(A.n); t = time.time(); s.solve(b, x); ts = time.time() - t
        i = s.get_info()
        rec = dict(N=N, precond=precond, ordering=("amd" if ordering else "natural") if precond == "ic" else None, factorize_s=round(tf, 3), solve_ms=round(ts * 1e3, 1),
                   iterations=int(i["solver_iter"]), true_residual=i["true_residual"])
        if precond == "ic":
            rec.update(levels_forward=int(s.get_param("ic.levels")), levels_backward=int(s.get_param("ic.levels_backward")))
            r, z = s.to_device(b), s.device_array(A.n)
            s.precond_apply_device(r, z); s.synchronize()
            t = time.time()
            for _ in range(5): s.precond_apply_device(r, z)
            s.synchronize(); rec["apply_ms"] = round((time.time() - t) / 5 * 1e3, 3)
        print(json.dumps(rec), flush=True); out.append(rec)
        del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_ic_amd.json"), "w"), indent=1)
