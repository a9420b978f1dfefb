#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_s2m_gpu.py tests/test_coop_gpu.py tests/test_window_gpu.py tests/test_lm_gpu.py tests/test_c_host_gpu.py tests/test_reference_gpu.py tests/test_multi_rank_gpu.py -q -m gpu -x > gpurun_out/r04/tests15.log 2>&1; echo "tests rc $?" > gpurun_out/r04/summary15.txt
tail -4 gpurun_out/r04/tests15.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench15.json 2> gpurun_out/r04/bench15.err; echo "bench rc $?" >> gpurun_out/r04/summary15.txt
cat gpurun_out/r04/summary15.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench15.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'fail', d.get('parity_failures'), 'regions', d['extras']['headline_regions']['median'])
for r in d['extras']['small_launches']: print(r)
print(d['extras']['blocking_seam'])
c=d['extras']['configs']
print('c0', c['0']['value'], 'c1', c['1']['value'], 'c4', c['4']['value'], c['4']['us_per_single_keyframe_linearize_blocking'], c['4']['cpp_seam'])
PY
