#!/bin/bash
# round 4: blocking calls whose reduction kernels write into page-locked memory — tests, then the seam figures of configs[4]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
timeout 900 python -m pytest tests/test_s2m_gpu.py tests/test_window_gpu.py tests/test_c_host_gpu.py tests/test_coop_gpu.py tests/test_reference_gpu.py tests/test_multi_rank_gpu.py -m gpu -q -x > gpurun_out/r04s/pytest.log 2>&1; grep -E "passed|failed|^E " gpurun_out/r04s/pytest.log | tail -5
timeout 300 python bench.py --config 4 > gpurun_out/r04s/c4.json 2> gpurun_out/r04s/c4.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04s/c4.json') if l.startswith('{')][-1])
print(d['value'], {k: d['config'].get(k) for k in d['config'] if 'us_' in k or k=='cpp_seam'} if 'config' in d else '')
print(json.dumps(d)[:1500])
PY
