#!/bin/bash
# round 4: the ROT extraction from page-locked host buffers — tests, timings, and a kernel + memory-copy trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
timeout 900 python -m pytest tests/test_extract_rot_gpu.py tests/test_reference_gpu.py tests/test_reference_cfg2_gpu.py tests/test_config0_gpu.py tests/test_replay_gpu.py tests/test_voxel_gpu.py tests/test_s2m_gpu.py -m gpu -q -x > gpurun_out/r04r/pytest.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r04r/pytest.log | tail -5
python tools/rot_host_time.py 2>&1 | tail -3; python tools/livox_timeline.py 2>&1 | tail -1
