#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
K="block or bsr3 or elasticity or refresh or config2 or newton or fem" bash scripts/r4/tests.sh
KINDS=elast bash scripts/r4/prof_refresh.sh
