#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
K="block or bsr3 or elasticity or config2" bash scripts/r4/tests.sh
python - <<'P'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from polysolve_amd import HIPSolver
from bench import AMG_RECOMMENDED
for rep in range(2):
  for var in (2, -1):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", block_size=3, tolerance=1e-8, bsr3_variant=var, amg=dict(AMG_RECOMMENDED))})
    s.generate_elasticity_q1(100); s.synchronize()
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(4):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); s.synchronize(); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    print(json.dumps(dict(variant=var, solve_ms=round(best * 1e3, 2), its=i["num_iterations"], res=i["true_residual"])), flush=True)
    b.free(); x.free(); del s
P
