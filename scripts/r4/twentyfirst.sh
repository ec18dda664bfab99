#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/r4/tests.sh
BLS=1 bash scripts/r4/prof_elast.sh 2>&1 | grep -E "cheb_first|per live|L0 cheb_step" | cut -c1-170
