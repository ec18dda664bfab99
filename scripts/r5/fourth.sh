#!/bin/bash
# round 5, fourth lease: the whole GPU suite on the current tree, the runtime-class tests, then setup laps / A/B
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_fourth_tests.log 2>&1
tail -12 gpurun_out/r05_fourth_tests.log
export SETS='{};{"aggregation":"parallel"};{"direct_coarse":true};{"aggregation":"parallel","direct_coarse":true}'
for spec in "poisson 216" "elast 100"; do
  set -- $spec
  KIND=$1 N=$2 timeout 600 python scripts/r5/ab.py > gpurun_out/r05_ab4_$1_$2.jsonl 2> gpurun_out/r05_ab4_$1_$2.err
  cat gpurun_out/r05_ab4_$1_$2.jsonl | cut -c1-600
  tail -3 gpurun_out/r05_ab4_$1_$2.err
done
AMG='{"aggregation":"parallel"}' python scripts/r5/setup_laps.py 2>&1 | grep "aggregation\|setup " | head -20
