#!/bin/bash
# round 3, GPU run 11: restructured ROT extractor (5 launches): full suite + extractor kernel stats + bench extras
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03s; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -5 $OUT/pytest_all.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o extract -- python tools/rot_phases.py > $OUT/rot_phases.log 2>&1
python tools/kstats.py $OUT/extract_kernel_stats.csv | head -9
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03s/bench.json').read().strip().splitlines()[0])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])
for k,v in d['extras'].items():
    if isinstance(v,dict) and 'ms' in v: print(k, v['ms'])
PY
