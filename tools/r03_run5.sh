#!/bin/bash
# round 3, GPU run 5: the default bench with the new extras (configs 0/1/4, seam, small launches), each --config alone, LM timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e; mkdir -p $OUT
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    e = d.get("extras", {})
    print("HEAD", d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d["roofline"]["frac"], d.get("inner_iteration", {}).get("us_per_iteration"))
    print("REG", e.get("headline_regions"))
    print("SEAM", e.get("blocking_seam"), e.get("blocking_seam_error"))
    for r in e.get("small_launches", []): print("SMALL", r)
    print(e.get("small_launches_error"))
    for k, v in e.get("configs", {}).items(): print("CFG", k, json.dumps(v)[:900])
    for k in ("extract_rot_device_resident", "extract_livox", "map_index_build", "map_index_build_base_only", "localmap_commit", "keyframe_pipeline", "frontend_flavour"):
        print(k, json.dumps(e.get(k))[:300])
except Exception as ex:
    print("failed", ex); print(open("$OUT/bench.err").read()[-3000:])
PY
for c in 0 1 4; do timeout 300 python bench.py --config $c --no-cpu-baseline > $OUT/cfg$c.json 2> $OUT/cfg$c.err; tail -c 600 $OUT/cfg$c.json; echo; done
timeout 600 python tools/lm_time.py $OUT/lm_time.json > $OUT/lm_time.log 2> $OUT/lm_time.err; cat $OUT/lm_time.log
