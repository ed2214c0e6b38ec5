#!/bin/bash
# tools/build_exp.sh NAME [extra hipcc flags...] -> tools/libexp_NAME.so : an experiment build of the product library
# (same sources, extra -D switches) that tools/microbench times next to the shipped one ("EXP tools/libexp_NAME.so").
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden -pthread -DENV2048_BUILDING -I$R/include -I$R/simulate_2048_amd/csrc \
  -mllvm -amdgpu-kernarg-preload-count=16 "$@" $R/simulate_2048_amd/csrc/env2048_kernels.hip -o $R/tools/libexp_$NAME.so
echo built tools/libexp_$NAME.so
