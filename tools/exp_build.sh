#!/bin/bash
# Experiment builds of the library: bash tools/exp_build.sh <name> <-D flags...>  ->  tools/_probe/<name>/liblili_hip.so  (lili_s2m.hip recompiled with the flags, every other object as built)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/lili_om_amd/csrc; NAME=$1; shift
OUT=$ROOT/tools/_probe/$NAME; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -c $CS/lili_s2m.hip -o $OUT/lili_s2m.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblili_hip.so $OUT/lili_s2m.o $CS/lili_api.o $CS/lili_map.o $CS/lili_match.o $CS/lili_s2m_coop.o $CS/lili_s2m_lm.o $CS/lili_extract_rot.o $CS/lili_extract_livox.o $CS/lili_voxel.o $CS/lili_formats.o $CS/lili_p2p.o $CS/lili_pipeline.o
echo built $OUT/liblili_hip.so
