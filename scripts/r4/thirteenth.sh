#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
KIND=elast timeout 300 python scripts/r4/refresh_bytes.py | cut -c1-600
KIND=poisson timeout 300 python scripts/r4/refresh_bytes.py | cut -c1-600
KINDS="elast poisson" bash scripts/r4/prof_refresh.sh 2>&1 | cut -c1-150
python scripts/r4/elast_ab.py 2>&1 | cut -c1-200 | tail -2
