#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
F='==|L0 restrict|L0 prolong|L1 cheb_step|L1 residual|L1 restrict|per live'
HIPJ='{"spmv_col16":1}' bash scripts/r4/prof_poisson.sh col16 '{}' 2>&1 | grep -E "$F" | cut -c1-170
bash scripts/r4/prof_poisson.sh ref '{}' 2>&1 | grep -E "$F" | cut -c1-170
timeout 600 python -m pytest tests/test_gpu_ic.py -x -q -m gpu 2>&1 | tail -2
