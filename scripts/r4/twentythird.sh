#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "packed_to_the_tile or 16_bit or wide_row" 2>&1 | tail -3
timeout 300 python scripts/r4/vrb_ab.py
N=216 timeout 300 python scripts/r4/vrb_ab.py
timeout 600 python scripts/r4/elast_prof.py 2>&1 | tail -3
