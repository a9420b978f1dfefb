#!/bin/bash
# Where does a wave of k_associate_surf spend its lifetime?  Builds a copy of the library with -DLILI_PHASE_PROBE (s_memrealtime stamps at
# the phase boundaries of the association, lili_s2m.hip) into tools/_probe/ and runs tools/assoc_blocks.py on it.
#   bash tools/assoc_phases.sh build          here (hipcc cross-compiles), the .so travels with gpurun
#   bash tools/assoc_phases.sh run [iters]    on the GPU box
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/lili_om_amd/csrc
OUT=$ROOT/tools/_probe
if [ "$1" = "build" ]; then
    mkdir -p $OUT
    make -C $CS -s
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DLILI_PHASE_PROBE -c $CS/lili_s2m.hip -o $OUT/lili_s2m_probe.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblili_hip.so $OUT/lili_s2m_probe.o $CS/lili_api.o $CS/lili_s2m_coop.o $CS/lili_s2m_lm.o $CS/lili_extract_rot.o $CS/lili_extract_livox.o $CS/lili_voxel.o $CS/lili_formats.o $CS/lili_p2p.o $CS/lili_pipeline.o
    echo built $OUT/liblili_hip.so
else
    shift || true
    LILI_HIP_LIBRARY=$OUT/liblili_hip.so LILI_PHASE_PROBE=1 python $ROOT/tools/assoc_blocks.py "${1:-10}"
fi
