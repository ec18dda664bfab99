"""Algorithmic bytes of the numeric refresh's kernels from the shapes of the hierarchy: prints a JSON
{kernel: {level: bytes}} for configs[2] (KIND=elast) or the 256^3 Poisson hierarchy (KIND=poisson)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
KIND = os.environ.get("KIND", "elast")
s = HIPSolver("")
if KIND == "elast":
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, amg=dict(AMG_RECOMMENDED))})
    s.generate_elasticity_q1(int(os.environ.get("M", "100")))
else:
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, amg=dict(AMG_RECOMMENDED))})
    N = int(os.environ.get("N", "256")); s.generate_poisson7(N, N, N)
s.synchronize()
nl = int(s.get_info()["amg_levels"])
shapes = []
for l in range(nl):
    d = {"A": s.amg_level_matrix_shape(l, 0)}
    if l + 1 < nl:
        for k, w in (("P", 1), ("R", 2), ("AP", 3)):
            d[k] = s.amg_level_matrix_shape(l, w)
    shapes.append({k: [int(x) for x in v] for k, v in d.items()})
out = {"kind": KIND, "levels": shapes, "bsr3_nnzb": int(s.get_param("bsr3_nnzb"))}
print(json.dumps(out))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r04_refresh_shapes_{KIND}.json"), "w"), indent=1)
