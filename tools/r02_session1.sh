#!/bin/bash
# Round-2 GPU session 1 (run through gpurun): full GPU suite, the default bench line, A/B of the launch-structure options,
# kernel-trace stats.  Everything lands under gpurun_out/r02a/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02a
mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
for opt in "fuse_tail=0" "fuse_tail=1"; do
  timeout 300 $B --opt $opt > $OUT/bench_$opt.json 2> $OUT/bench_$opt.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_$opt.json").read().strip().splitlines()[-1])
    print("AB $opt", d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d.get("inner_iteration", {}).get("us_per_iteration"), d["final_pose"]["t"])
except Exception as e:
    print("AB $opt failed", e)
PY
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
python tools/profile_summary.py $OUT 2>&1 | head -20
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "inner_iteration", "pose_delta_vs_cpu", "gpu_over_cpu") if k in d}))
print(json.dumps(d.get("extras")))
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
