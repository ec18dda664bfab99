#!/bin/bash
# round 5, third lease: tests of the round, then the overlap A/B and the parallel aggregation after its fix
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "product_plans or parallel_aggregation or packed_row_blocks_stay or kept_symbolic_work or refresh_on_same_pattern or device_setup_equals_host or amg" > gpurun_out/r05_third_tests.log 2>&1
tail -5 gpurun_out/r05_third_tests.log
export SETS='{"overlap_smoothers":0};{"overlap_smoothers":1};{"aggregation":"parallel"}'
for spec in "poisson 216" "poisson 256" "elast 100"; do
  set -- $spec
  KIND=$1 N=$2 timeout 600 python scripts/r5/ab.py > gpurun_out/r05_ab3_$1_$2.jsonl 2> gpurun_out/r05_ab3_$1_$2.err
  cat gpurun_out/r05_ab3_$1_$2.jsonl | cut -c1-700
done
AMG='{"aggregation":"parallel"}' python scripts/r5/setup_laps.py 2>&1 | grep "aggregation\|setup " | head -20
AMG='{}' python scripts/r5/setup_laps.py 2>&1 | grep -v "^\[psolve timing\] factorize" | tail -45
