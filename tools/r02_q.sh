#!/bin/bash
# quick GPU check (through gpurun): bash tools/r02_q.sh <tag> [pytest selection] [extra bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-q}; SEL=${2:-tests/test_s2m_gpu.py}; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$SEL" != "none" ]; then
  ( time timeout 900 python -m pytest $SEL -m gpu -x -q ) > $OUT/pytest.log 2>&1
  tail -4 $OUT/pytest.log
fi
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras $@"
timeout 300 $B > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], "it/s", d["ms_per_step"], "ms  assoc", d["roofline"]["us_per_launch"], "us  inner", d.get("inner_iteration", {}).get("us_per_iteration"), "us  pose", d["final_pose"]["t"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench.err").read()[-1500:])
PY
LILI_PHASES=1 timeout 300 $B > /dev/null 2> $OUT/phases.err; tail -11 $OUT/phases.err
