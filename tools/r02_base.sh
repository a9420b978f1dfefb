#!/bin/bash
# Round-2 baseline (through gpurun): default bench line, kernel-trace stats of the same command, phase timings.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r02base}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
python tools/profile_summary.py $OUT 2>&1 | head -20
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "inner_iteration", "pose_delta_vs_cpu", "gpu_over_cpu") if k in d}))
print(json.dumps(d.get("extras")))
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
