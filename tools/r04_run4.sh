#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r04/gpu_tests4.log 2>&1; echo "gpu tests rc $?" > gpurun_out/r04/summary4.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r04/bench4_c4.json 2> gpurun_out/r04/bench4_c4.err; echo "bench c4 rc $?" >> gpurun_out/r04/summary4.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lf_prof -o lf -- $GRAFT_REPO_ROOT/tools/_probe/launch_floor 1000 > /tmp/lf_prof.log 2>&1; find /tmp/lf_prof -type f >> /tmp/lf_prof.log )
f=$(find /tmp/lf_prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv
grep -E "^\(|uncached" /tmp/lf_prof.log > gpurun_out/r04/launch_floor_under_rocprof.txt
tail -14 gpurun_out/r04/gpu_tests4.log; cat gpurun_out/r04/summary4.txt
head -8 gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench4_c4.json').read().strip().splitlines()[-1])
c4=d['details']
print(d['value'], {k:c4.get(k) for k in ('us_per_window_evaluation','us_per_single_keyframe_linearize_blocking','us_per_window_association_blocking','cpp_seam')})
PY
