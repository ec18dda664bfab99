#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in '{"aggregation":"amgcl"}' '{"aggregation":"compact"}' '{"aggregation":"compact","direct_coarse":true}' '{"aggregation":"parallel"}'; do
  echo "== elast $v"
  PSOLVE_TIMING=1 CASES=elast VARIANTS="[$v]" python scripts/r6/agg_eval.py 2>&1 | grep -E "psolve timing|case" | tail -45
done
echo "== poisson216 compact"
PSOLVE_TIMING=1 CASES=poisson216 VARIANTS='[{"aggregation":"compact"}]' python scripts/r6/agg_eval.py 2>&1 | grep -E "psolve timing|case" | tail -45
