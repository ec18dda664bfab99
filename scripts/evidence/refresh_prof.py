"""One first factorize + K numeric refreshes (env K, default 6) of configs[2] (KIND=elast, M) or of the 256^3 Poisson
hierarchy (KIND=poisson, N) under AMG_RECOMMENDED, for rocprofv3; the refresh is timed without the generator (the matrix is
generated once, then factorize_device is called on the same device arrays with the values scaled in place)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from polysolve_amd import HIPSolver
from bench_legs import AMG_RECOMMENDED
KIND = os.environ.get("KIND", "elast"); K = int(os.environ.get("K", "6"))
s = HIPSolver("")
if KIND == "elast":
    M = int(os.environ.get("M", "100"))
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(AMG_RECOMMENDED, **json.loads(os.environ.get("AMG", "{}"))))})
    MODE = int(os.environ.get("MODE", "0"))  # 1: the nodes renumbered pseudo-randomly (no block-row kinds: what a caller's mesh gets)
    gen = (lambda: s.generate_elasticity_q1(M)) if MODE == 0 else (lambda: s.generate_elasticity_q1_permuted(M, mode=MODE, seed=7))
else:
    N = int(os.environ.get("N", "256"))
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(AMG_RECOMMENDED, **json.loads(os.environ.get("AMG", "{}"))))})
    gen = lambda: s.generate_poisson7(N, N, N)
gen(); s.synchronize()
ts = []
for k in range(K):
    t = time.time(); gen(); s.synchronize(); ts.append(time.time() - t)
    assert s.get_param("amg.last_setup_reused") == 1
print(json.dumps(dict(kind=KIND, refreshes=K, generate_plus_refresh_s=ts)))
