#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_schwarz.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from polysolve_amd import HIPSolver
for N in (256, 384):
    s = HIPSolver("")
    s.generate_poisson7(N)
    for _ in range(2):
        print(N, "K2 %.4f ms  K3 %.4f ms" % s.time_vecops(50), flush=True)
    n = s.matrix_shape()[0]
    print("   K2 %.0f GB/s  K3 %.0f GB/s" % tuple(b * n / (t * 1e6) for b, t in zip((32, 48), s.time_vecops(50))))
    del s
PY
timeout 300 python bench.py --no-cpu-baseline --no-north-star --no-extra --steps 5 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 'ms/it', j['ms_per_iteration'], 'spmv', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])"
