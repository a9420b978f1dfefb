#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn_keep_gpu.py -q -m gpu -x > gpurun_out/r04/keep_tests.log 2>&1; echo "keep tests rc $?" > gpurun_out/r04/summary7.txt
tail -25 gpurun_out/r04/keep_tests.log
for nc in 1 0; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --opt nn_cache=$nc > gpurun_out/r04/bench7_nc$nc.json 2> gpurun_out/r04/bench7_nc$nc.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench7_nc$nc.json').read().strip().splitlines()[-1])
print('nn_cache=$nc', d['value'], 'it/s', d['ms_per_step'], 'assoc us', d['roofline']['us_per_launch'], 'pose', d['final_pose'])
PY
done
cat gpurun_out/r04/summary7.txt
