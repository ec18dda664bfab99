"""BASELINE.json configs[4] at scale: Newton on f(x) = 1/2 x'Ax - b'x + c/4 sum x^4 (A = 7-point Poisson N^3),
every Hessian solve through the host entry points (analyze_pattern / factorize / solve, Newton.cpp:173-214).
Prints per Newton iteration: factorize s (upload + preconditioner), solve s (incl. b/x transfers), CG iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import oracle as O
from polysolve_amd import Solver

N = int(os.environ.get("NEWTON_N", "128"))
A = O.poisson7(N).to_scipy().tocsr()
n = A.shape[0]
rng = np.random.default_rng(0)
b = rng.uniform(-1, 1, n) * 50
c = 2.0
diag_pos = np.flatnonzero(A.indices == np.repeat(np.arange(n), np.diff(A.indptr)))
for name, hip in [("jacobi", dict(precond="jacobi")),
                  ("amg (refresh on the same pattern)", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20))),
                  ("amg (reuse off: full device setup each time)", dict(precond="amg", amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, reuse=0)))]:
    s = Solver.create({"solver": "HIP", "HIP": dict(hip, tolerance=1e-10, absolute_tolerance=1e-9, max_iter=5000)})
    x = np.zeros(n); d = np.zeros(n)
    H = A.copy()
    print(f"== N={N} ({n} DOF) {name}")
    tot_f = tot_s = 0.0
    for it in range(30):
        g = A @ x - b + c * x ** 3
        gn = np.linalg.norm(g)
        if gn < 1e-7: break
        H.data[:] = A.data
        H.data[diag_pos] += 3 * c * x ** 2          # Hessian = A + 3c diag(x^2): same pattern, new values
        t = time.time(); s.analyze_pattern(H, n); s.factorize(H); tf = time.time() - t
        t = time.time(); s.solve(-g, d); ts = time.time() - t
        i = s.get_info()
        res = np.linalg.norm(H @ d + g)
        print(f"  newton {it}: |g|={gn:.2e} factorize {tf*1e3:7.1f} ms  solve {ts*1e3:7.1f} ms  (lib {i['time_solve']*1e3:6.1f}, device {i['time_solve_device']*1e3:6.1f})  cg its {i['num_iterations']:4d}  |Hd+g|={res:.1e}", flush=True)
        tot_f += tf; tot_s += ts
        f0 = 0.5 * x @ (A @ x) - b @ x + 0.25 * c * np.sum(x ** 4)
        rate = 1.0
        while True:
            xn = x + rate * d
            if 0.5 * xn @ (A @ xn) - b @ xn + 0.25 * c * np.sum(xn ** 4) <= f0 + 1e-4 * rate * (g @ d) or rate < 1e-8: break
            rate *= 0.5
        x = xn
    print(f"  total: {it} Newton iterations, factorize {tot_f:.3f} s, solve {tot_s:.3f} s")
