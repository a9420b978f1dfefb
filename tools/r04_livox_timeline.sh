#!/bin/bash
# round 4: the Livox extraction chain under a kernel + memory-copy trace, after its tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
timeout 600 python -m pytest tests/test_extract_livox_gpu.py tests/test_reference_gpu.py tests/test_reference_cfg2_gpu.py tests/test_sequence_gpu.py tests/test_replay_gpu.py tests/test_formats_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/livox_timeline.py
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r04l -o lv -- python tools/livox_timeline.py > gpurun_out/r04l/lv.log 2>&1
