#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/r4/tests.sh
F='==|pcg_update|pcg_dot|per live'
bash scripts/r4/prof_poisson.sh vec '{}' 2>&1 | grep -E "$F" | cut -c1-170
python scripts/r4/elast_ab.py 2>&1 | cut -c1-170 | tail -2
