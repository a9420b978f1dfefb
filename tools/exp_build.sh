#!/bin/bash
# Experiment builds of the library: bash tools/exp_build.sh <name> <-D flags...>  ->  tools/_probe/<name>/liblili_hip.so  (lili_s2m.hip recompiled with the flags, every other object as built)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/lili_om_amd/csrc; NAME=$1; shift
OUT=$ROOT/tools/_probe/$NAME; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -c $CS/lili_s2m.hip -o $OUT/lili_s2m.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblili_hip.so $OUT/lili_s2m.o $(ls $CS/*.o | grep -v "/lili_s2m.o")
echo built $OUT/liblili_hip.so
