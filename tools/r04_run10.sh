#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for nc in 1 0; do
  echo "=== nn_cache=$nc"; LILI_NN_CACHE=$nc LILI_HIP_LIBRARY=$PWD/tools/_probe/liblili_hip.so LILI_PHASE_PROBE=1 timeout 300 python tools/assoc_blocks.py 10 131072 2>&1 | grep -v "amdgpu.ids" | head -40
done > gpurun_out/r04/assoc_phases_keep.txt
cat gpurun_out/r04/assoc_phases_keep.txt
