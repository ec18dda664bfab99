#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python -m pytest tests/test_gpu_amg.py -x -q -m gpu 2>&1 | tail -3
KINDS=poisson bash scripts/r4/prof_refresh.sh 2>&1 | grep -E "^\{|one refresh|spgemm" | cut -c1-150
