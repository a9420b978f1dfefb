#!/bin/bash
# round 4, first GPU call: new parity tests, launch-floor probe, bench self-check
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 600 python -m pytest tests/test_reference_cfg2_gpu.py -x -q -m gpu > gpurun_out/r04/cfg2_tests.log 2>&1; echo "cfg2 tests rc $?" >> gpurun_out/r04/summary.txt
timeout 120 tools/_probe/launch_floor 2000 > gpurun_out/r04/launch_floor.txt 2>&1; echo "launch_floor rc $?" >> gpurun_out/r04/summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err; echo "bench rc $?" >> gpurun_out/r04/summary.txt
tail -5 gpurun_out/r04/cfg2_tests.log; cat gpurun_out/r04/launch_floor.txt; cat gpurun_out/r04/summary.txt; tail -3 gpurun_out/r04/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'pose', d.get('pose_delta_vs_cpu'), 'fail', d.get('parity_failures'))
for k,v in d['extras']['configs'].items(): print(k, v.get('value'), v.get('parity'), v.get('error'))
PY
