#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python -m pytest tests/test_gpu_ic.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python scripts/r4/ic_amd.py 2>&1 | tail -8
