#!/bin/bash
# tools/build_exp.sh NAME [extra hipcc flags...] -> tools/libexp_NAME.so : an experiment build of the product library
# (same sources, extra -D switches) that tools/microbench times next to the shipped one ("EXP tools/libexp_NAME.so").
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
python3 $R/simulate_2048_amd/build.py --out $R/tools/libexp_$NAME.so "$@" > /dev/null
echo built tools/libexp_$NAME.so
