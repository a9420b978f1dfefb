#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_s2m_gpu.py tests/test_coop_gpu.py tests/test_window_gpu.py tests/test_lm_gpu.py tests/test_reference_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > gpurun_out/r04/tests16.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/r04/tests16.log
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/r04/bench16.json 2> gpurun_out/r04/bench16.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench16.json').read().strip().splitlines()[-1])
print(d['value'], 'it/s', d['ms_per_step'], 'assoc us', d['roofline']['us_per_launch'], 'inner', d['inner_iteration']['us_per_iteration'], d['final_pose']['t'])
PY
done
( cd /tmp && rm -rf /tmp/kp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /tmp/kp.log 2>&1 )
f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-200
