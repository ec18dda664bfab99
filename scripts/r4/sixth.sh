#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
python -m pytest tests/test_gpu_solver.py tests/test_gpu_fem.py tests/test_gpu_newton.py tests/test_gpu_multi.py tests/test_gpu_reorder.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "config4" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python scripts/r4/host_contract.py 2>&1 | cut -c1-1600
