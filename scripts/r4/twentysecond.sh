#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/r4/tests.sh
F='==|L0 restrict|L1 cheb_step|L1 residual|L1 restrict|L2 cheb_step|per live'
bash scripts/r4/prof_poisson.sh vrb '{}' 2>&1 | grep -E "$F" | cut -c1-170
N=216 bash scripts/r4/prof_poisson.sh vrb216 '{}' 2>&1 | grep -E "$F" | cut -c1-170
