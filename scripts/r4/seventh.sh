#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
bash scripts/r4/prof_poisson.sh r64 '{"level_rows_per_block":64}' r16 '{"level_rows_per_block":16}' r128 '{"level_rows_per_block":128}' base3 '{}' 2>&1 | grep -E "==|L1 cheb_step|L1 residual|per live|json"
bash scripts/r4/pmc_level1.sh
