#!/bin/bash
# round 3, GPU run 4: LM (512-thread workgroups), sharded window / stalled peer / 4 ranks, full suite, LM timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_lm_gpu.py -m gpu -q ) > $OUT/pytest_lm.log 2>&1
tail -25 $OUT/pytest_lm.log
( time timeout 1200 python -m pytest tests/test_multi_rank_gpu.py -m gpu -q -x ) > $OUT/pytest_mr.log 2>&1
tail -30 $OUT/pytest_mr.log
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_lm_gpu.py --deselect tests/test_multi_rank_gpu.py ) > $OUT/pytest_all.log 2>&1
tail -8 $OUT/pytest_all.log
timeout 600 python tools/lm_time.py $OUT/lm_time.json > $OUT/lm_time.log 2> $OUT/lm_time.err
cat $OUT/lm_time.log; tail -3 $OUT/lm_time.err
