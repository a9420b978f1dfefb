#!/bin/bash
export TMPDIR=/tmp
for nc in 0 1; do
  echo "=== nn_cache=$nc"; LILI_NN_CACHE=$nc LILI_HIP_LIBRARY=$PWD/tools/_probe/liblili_hip.so LILI_PHASE_PROBE=1 timeout 300 python tools/assoc_blocks.py 10 131072 2>&1 | grep -v "amdgpu.ids" | grep -E "kernel span|wave lifetime|end time|moved into|ranges|walked|winners|fitted|stored|slow block" | head -16
done
