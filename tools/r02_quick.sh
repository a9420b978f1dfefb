#!/bin/bash
# quick GPU A/B (through gpurun): bash tools/r02_quick.sh <tag> [pytest selection]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-q}; SEL=${2:-tests/test_s2m_gpu.py}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 900 python -m pytest $SEL -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
for opt in "fuse_tail=0" "fuse_tail=1"; do
  timeout 300 $B --opt $opt > $OUT/bench_$opt.json 2> $OUT/bench_$opt.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$opt.json").read().strip().splitlines()[-1])
    print("AB $opt", d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d.get("inner_iteration", {}).get("us_per_iteration"), d["final_pose"]["t"])
except Exception as e:
    print("AB $opt failed", e); print(open("$OUT/bench_$opt.err").read()[-1500:])
PY
done
LILI_PHASES=1 timeout 300 $B > /dev/null 2> $OUT/phases.err; tail -14 $OUT/phases.err
