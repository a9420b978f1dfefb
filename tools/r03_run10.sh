#!/bin/bash
# round 3, GPU run 10: persistent outer iterations (k_iterate_coop) + wave-parallel LM step: targeted tests first, timing, then the full suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03q; mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_coop_gpu.py tests/test_lm_gpu.py -q -x ) > $OUT/pytest_coop_lm.log 2>&1
tail -25 $OUT/pytest_coop_lm.log
timeout 300 python tools/iter_time.py $OUT/iter_time.json > $OUT/iter_time.log 2> $OUT/iter_time.err; cat $OUT/iter_time.log; tail -3 $OUT/iter_time.err
timeout 300 python tools/lm_time.py $OUT/lm_time.json > $OUT/lm_time.log 2> $OUT/lm_time.err; cat $OUT/lm_time.log; tail -3 $OUT/lm_time.err
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -15 $OUT/pytest_all.log
