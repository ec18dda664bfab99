#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/r4/tests.sh
KINDS=poisson bash scripts/r4/prof_refresh.sh 2>&1 | grep -E "^\{|one refresh|gershgorin|prolongation|spgemm" | cut -c1-150
