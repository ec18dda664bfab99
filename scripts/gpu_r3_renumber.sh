#!/bin/bash
# round 3: renumbered coarse levels + four gathers in flight on the wide-row path
mkdir -p gpurun_out
python -m pytest tests/test_gpu_amg.py -x -q -m gpu -k "renumbered" 2>&1 | tail -15
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ragged or poisson_bit" 2>&1 | tail -4
for rn in 0 1; do
RN=$rn python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from polysolve_amd import HIPSolver
rn = int(os.environ["RN"])
for N in (216, 256):
    s = HIPSolver("")
    s.set_parameters({"HIP": dict(precond="amg", tolerance=1e-8, max_iter=2000, amg=dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_power_iters=20, renumber=rn))})
    s.generate_poisson7(N); s.synchronize()
    s.set_parameters({"HIP": {"amg": {"reuse": False}}})
    t = time.perf_counter(); s.generate_poisson7(N); s.synchronize(); ts = time.perf_counter() - t
    n = s.matrix_shape()[0]
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best = 1e9
    for _ in range(3):
        s.axpby_device(n, 0.0, b, 0.0, x); s.synchronize()
        t = time.perf_counter(); s.solve_device(b, x); best = min(best, time.perf_counter() - t)
    i = s.get_info()
    print(f"renumber={rn} N={N} setup {ts:.3f} s solve {best*1e3:.1f} ms its={i['num_iterations']} res={i['true_residual']:.2e} levels={[s.amg_level_info(l)[:2] for l in range(i['amg_levels'])]}", flush=True)
    del s
PY
done
echo "--- level 1 standalone, setup numbering"; RENUMBER=0 N=256 python scripts/gpu_wide_rows.py 2>&1 | head -12
echo "--- level 1 standalone, renumbered"; ALL=1 RENUMBER=1 N=256 python scripts/gpu_wide_rows.py 2>&1 | head -20
