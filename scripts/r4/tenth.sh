#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 600 python scripts/r4/reorder_agg.py 2>&1 | cut -c1-330
F='==|L2 restrict|L1 restrict|per live'
bash scripts/r4/prof_poisson.sh new2 '{}' 2>&1 | grep -E "$F"
