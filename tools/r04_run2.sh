#!/bin/bash
# round 4, GPU call 2: the whole GPU suite on the new window seam / LM / ADVICE fixes, bench, trivial kernels under rocprofv3
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04/gpu_tests.log 2>&1; echo "gpu tests rc $?" > gpurun_out/r04/summary2.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err; echo "bench rc $?" >> gpurun_out/r04/summary2.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lf_prof -o lf -- $GRAFT_REPO_ROOT/tools/_probe/launch_floor 1000 > /tmp/lf_prof.log 2>&1 )
find /tmp/lf_prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv
tail -4 gpurun_out/r04/gpu_tests.log; cat gpurun_out/r04/summary2.txt; tail -2 gpurun_out/r04/bench2.err
cat gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv | head -8
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench2.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'fail', d.get('parity_failures'))
c4=d['extras']['configs']['4']
print({k:c4.get(k) for k in ('value','us_per_window_evaluation','us_per_single_keyframe_linearize_blocking','us_per_window_association_blocking','cpp_seam','device_lm','parity')})
PY
