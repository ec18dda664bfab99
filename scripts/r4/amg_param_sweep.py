"""AMG_RECOMMENDED revisited with the round-4 kernels: one parameter changed at a time, Poisson N^3 and Q1 elasticity M^3;
setup s, solve ms (min of 3), iterations."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
base = dict(AMG_RECOMMENDED)
variants = [("base", {}), ("deg3", dict(cheb_degree=3)), ("lower0.05", dict(cheb_lower=0.05)), ("lower0.15", dict(cheb_lower=0.15)),
            ("lower0.2", dict(cheb_lower=0.2)), ("relax1.0", dict(sa_relax=1.0)), ("relax1.5", dict(sa_relax=1.5)),
            ("higher1.05", dict(cheb_higher=1.05)), ("higher1.2", dict(cheb_higher=1.2)), ("npre0", dict(npre=0)), ("npost0", dict(npost=0)),
            ("deg3lower0.05", dict(cheb_degree=3, cheb_lower=0.05)), ("power10", dict(cheb_power_iters=10))]
cases = [(c.split(":")[0], int(c.split(":")[1])) for c in os.environ.get("CASES", "poisson:216,elast:100").split(",")]
out = []
for kind, N in cases:
    for name, ch in variants:
        amg = dict(base, **ch)
        s = HIPSolver("")
        top = {"tolerance": 1e-8, "max_iter": 2000, "precond": "amg", "amg": amg}
        if kind == "elast": top["block_size"] = 3
        try:
            s.set_parameters({"HIP": top})
        except Exception as e:
            print(kind, N, name, "rejected:", str(e)[:80], flush=True); continue
        t = time.time()
        if kind == "poisson": s.generate_poisson7(N, N, N)
        else: s.generate_elasticity_q1(N)
        s.synchronize(); ts = time.time() - t
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        tt = []
        for _ in range(3):
            s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
            t = time.time(); s.solve_device(b, x); s.synchronize(); tt.append(time.time() - t)
        i = s.get_info()
        print(kind, N, f"{name:14s} setup {ts:.3f} s  solve {min(tt)*1e3:7.2f} ms  its {i['num_iterations']:3d}  true {i['true_residual']:.1e}", flush=True)
        out.append(dict(kind=kind, N=N, variant=name, amg=amg, setup_s=ts, solve_ms=min(tt) * 1e3, iterations=i["num_iterations"]))
        b.free(); x.free(); del s
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_amg_param_sweep.json"), "w"), indent=1)
