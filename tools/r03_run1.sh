#!/bin/bash
# round 3, GPU run 1 (through gpurun): cooperative-association tests, full GPU suite, size x lanes sweep, default bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a; mkdir -p $OUT
( time timeout 600 python -m pytest tests/test_coop_gpu.py -m gpu -x -q ) > $OUT/pytest_coop.log 2>&1
tail -5 $OUT/pytest_coop.log
( time timeout 900 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -8 $OUT/pytest_all.log
timeout 900 python tools/coop_sweep.py $OUT/sweep.json > $OUT/sweep.log 2> $OUT/sweep.err
tail -3 $OUT/sweep.err
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
