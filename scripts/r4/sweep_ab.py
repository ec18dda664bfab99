"""products of one AMG cycle sweep their operator from alternating ends (default) against all forward ("lab.alternate" 8):
AMG-PCG solve time, interleaved in one process; x must be bit-equal."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
cases = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ.get("CASES", "elast:100,elast:64,poisson:128,poisson:160,poisson:216,poisson:256").split(",")]
out = []
for kind, N in cases:
    row, xs = {}, []
    for flag in [int(v) for v in os.environ.get("FLAGS", "8,0,8,0").split(",")]:
        s = HIPSolver("")
        top = {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "amg": dict(AMG_RECOMMENDED), "lab.alternate": flag}
        if kind == "elast": top["block_size"] = 3
        s.set_parameters({"HIP": top})
        if kind == "poisson": s.generate_poisson7(N, N, N)
        else: s.generate_elasticity_q1(N)
        s.synchronize()
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        ts = []
        for _ in range(4):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.time(); s.solve_device(b, x); s.synchronize(); ts.append(time.time() - t)
        row.setdefault(f"flag{flag}", []).append(round(min(ts) * 1e3, 2))
        xs.append(x.download()); its = s.get_info()["num_iterations"]
        b.free(); x.free(); del s
    eq = all(np.array_equal(xs[0], v) for v in xs[1:])
    print(kind, N, "its", its, row, "bit-equal", eq, flush=True)
    out.append(dict(kind=kind, N=N, iterations=its, ms=row, bit_equal=bool(eq)))
HIPSolver("").set_parameters({"HIP": {"lab.alternate": 0}})
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_sweep_ab.json"), "w"), indent=1)
