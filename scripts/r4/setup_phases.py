"""first-setup phases of the N^3 Poisson hierarchy (PSOLVE_TIMING laps of the device setup; every lap synchronises, so the
total is a little above the untimed setup) and of configs[2]; env N, ELAST_M."""
import os, sys, time
os.environ["PSOLVE_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
N = int(os.environ.get("N", "256"))
for rep in range(2):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(AMG_RECOMMENDED))})
    print(f"== poisson {N}^3 rep {rep}", file=sys.stderr, flush=True)
    t = time.time(); s.generate_poisson7(N, N, N); s.synchronize(); print(f"setup {time.time()-t:.4f} s", file=sys.stderr, flush=True)
    del s
M = int(os.environ.get("ELAST_M", "100"))
if M:
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, block_size=3, amg=dict(AMG_RECOMMENDED))})
    print(f"== elasticity M={M}", file=sys.stderr, flush=True)
    t = time.time(); s.generate_elasticity_q1(M); s.synchronize(); print(f"setup {time.time()-t:.4f} s", file=sys.stderr, flush=True)
