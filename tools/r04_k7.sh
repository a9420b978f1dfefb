#!/bin/bash
# round 4, K7: index builds that read the caller's cloud in place and start from the previous build's box — tests, then the build's timings + kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04k; mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_map_build_gpu.py tests/test_fullsize_gpu.py tests/test_dense_map_gpu.py tests/test_knn_stress_gpu.py tests/test_voxel_gpu.py tests/test_s2m_gpu.py -m gpu -q -x ) > $OUT/pytest_k7.log 2>&1
grep -E "passed|failed|Error|error" $OUT/pytest_k7.log | tail -8
timeout 300 python tools/k7_time.py > $OUT/k7_time.json 2> $OUT/k7_time.err; cat $OUT/k7_time.json; tail -3 $OUT/k7_time.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k7 -- python tools/k7_time.py > /dev/null 2> $OUT/k7_prof.err
python tools/kstats.py $OUT/k7_kernel_stats.csv | head -16
