#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
python -m pytest tests/test_gpu_amg.py tests/test_gpu_kernels.py -x -q -m gpu -k "block or bsr3" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python scripts/r4/bsr_var_lab.py
python scripts/r4/host_contract.py 2>&1 | cut -c1-1500
