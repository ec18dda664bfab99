#!/bin/bash
# build library variants for A/B runs: scripts/r4/build_variants.sh name "EXTRA_DEFINES" ...
set -e
cd /root/repo
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  touch polysolve_amd/csrc/kernels.hip
  make -s -C polysolve_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result $defs"
  cp polysolve_amd/lib/libpsolve_hip.so scripts/lab/ab/lib_$name.so
  echo built $name
done
touch polysolve_amd/csrc/kernels.hip
make -s -C polysolve_amd/csrc -j8
