#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 600 python -m pytest tests/test_gpu_amg.py -q -m gpu -x 2>&1 | tail -3
ELAST_M=100 timeout 500 python scripts/r4/setup_phases.py 2>&1 | grep -E '^==|R \(A P\)|A P  |setup' | awk '/rep 1/,0'
