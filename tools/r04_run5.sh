#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
bash tools/assoc_split_probe.sh r04 > gpurun_out/r04/assoc_split.log 2>&1
timeout 600 python tools/graph_probe.py > gpurun_out/r04/graph_probe.jsonl 2> gpurun_out/r04/graph_probe.err
cat gpurun_out/r04/assoc_split.txt; cat gpurun_out/r04/graph_probe.jsonl; tail -3 gpurun_out/r04/graph_probe.err
