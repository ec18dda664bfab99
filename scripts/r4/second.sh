#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python -m pytest tests/test_gpu_amg.py tests/test_gpu_kernels.py -x -q -m gpu -k "block or bsr3 or refresh or fp32 or unsorted or elasticity" 2>&1 | tail -5
bash scripts/r4/prof_elast.sh
