#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
cp polysolve_amd/lib/libpsolve_hip.so /tmp/lib_cur.so
V="r3 v0 v3" bash scripts/r4/ab_libs.sh "WGS=6 python scripts/r4/bsr_lab.py" 3 | cut -c1-100
cp /tmp/lib_cur.so polysolve_amd/lib/libpsolve_hip.so
python -m pytest tests/test_gpu_amg.py tests/test_gpu_kernels.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3
python scripts/r4/elast_ab.py | cut -c1-200
