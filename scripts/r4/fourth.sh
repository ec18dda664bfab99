#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
python -m pytest tests/test_gpu_amg.py tests/test_gpu_kernels.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
bash scripts/r4/prof_elast.sh 2>&1 | grep -v simple_timer | cut -c1-170
