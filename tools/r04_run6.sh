#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
bash tools/assoc_split_probe.sh r04 > gpurun_out/r04/assoc_split.log 2>&1
cat gpurun_out/r04/assoc_split.txt
