#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 600 python tools/keep_probe.py > gpurun_out/r04/keep_probe.jsonl 2> gpurun_out/r04/keep_probe.err
cat gpurun_out/r04/keep_probe.jsonl; tail -3 gpurun_out/r04/keep_probe.err
