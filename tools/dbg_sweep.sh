#!/bin/bash
# association sensitivity: bench line under LILI_DEBUG ablation bits (results are wrong under most of them; timing only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-dbg}; mkdir -p $OUT; shift
for bits in "$@"; do
  LILI_DEBUG=$bits timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/b$bits.json 2> $OUT/b$bits.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/b$bits.json").read().strip().splitlines()[-1])
    print("LILI_DEBUG=$bits", d["value"], "it/s", d["ms_per_step"], "ms  assoc", d["roofline"]["us_per_launch"], "us  pose", d["final_pose"]["t"][0])
except Exception as e:
    print("bench failed", e); print(open("$OUT/b$bits.err").read()[-800:])
PY
done
