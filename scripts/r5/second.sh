#!/bin/bash
# round 5, second lease: the new tests (product plans, parallel aggregation, advice regressions), then A/B timings
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "product_plans or parallel_aggregation or packed_row_blocks_stay or kept_symbolic_work or refresh_on_same_pattern or device_setup_equals_host" > gpurun_out/r05_second_tests.log 2>&1
tail -15 gpurun_out/r05_second_tests.log
for spec in "poisson 216" "poisson 256" "elast 100"; do
  set -- $spec
  KIND=$1 N=$2 LAPS=1 timeout 600 python scripts/r5/ab.py > gpurun_out/r05_ab_$1_$2.jsonl 2> gpurun_out/r05_ab_$1_$2.err
  cat gpurun_out/r05_ab_$1_$2.jsonl | cut -c1-900
  grep "product plans" gpurun_out/r05_ab_$1_$2.err | head -8
done
