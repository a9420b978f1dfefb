#!/bin/bash
# full GPU suite (through gpurun): bash tools/r02_tests.sh <tag> [pytest selection]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-t}; SEL=${2:-tests}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 1500 python -m pytest $SEL -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | head -20
