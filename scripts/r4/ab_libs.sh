#!/bin/bash
# A/B of two builds of the library on ONE box, interleaved: scripts/lab/ab/lib_base.so against lib_new.so
# usage: ab_libs.sh "<command printing a JSON line>" [rounds]
R=${GRAFT_REPO_ROOT:-.}
cd $R
CMD="$1"; N=${2:-2}
for i in $(seq 1 $N); do
  for v in ${V:-base new}; do
    cp scripts/lab/ab/lib_$v.so polysolve_amd/lib/libpsolve_hip.so
    echo "== $v"
    eval "$CMD"
  done
done
