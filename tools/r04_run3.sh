#!/bin/bash
# round 4, GPU call 3: window association in one launch, extractor pinned lookup, association ablations, trivial kernels under rocprofv3
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_s2m_gpu.py tests/test_window_gpu.py tests/test_c_host_gpu.py tests/test_extract_rot_gpu.py tests/test_multi_rank_gpu.py tests/test_coop_gpu.py -q -m gpu -x > gpurun_out/r04/gpu_tests3.log 2>&1; echo "gpu tests rc $?" > gpurun_out/r04/summary3.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r04/bench3_c4.json 2> gpurun_out/r04/bench3_c4.err; echo "bench c4 rc $?" >> gpurun_out/r04/summary3.txt
bash tools/dbg_sweep.sh r04/dbg 0 2048 16384 1 > gpurun_out/r04/dbg_sweep.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/lf_prof -o lf -- $GRAFT_REPO_ROOT/tools/_probe/launch_floor 1000 > /tmp/lf_prof.log 2>&1; echo "rocprof rc $?" >> /tmp/lf_prof.log; find /tmp/lf_prof -type f >> /tmp/lf_prof.log )
cp /tmp/lf_prof.log gpurun_out/r04/lf_prof.log
f=$(find /tmp/lf_prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv
tail -3 gpurun_out/r04/gpu_tests3.log; cat gpurun_out/r04/summary3.txt; cat gpurun_out/r04/dbg_sweep.txt; tail -12 gpurun_out/r04/lf_prof.log
head -6 gpurun_out/r04/launch_floor_under_rocprof_kernel_stats.csv 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench3_c4.json').read().strip().splitlines()[-1])
c4=d['details']
print(d['value'], {k:c4.get(k) for k in ('us_per_window_evaluation','us_per_single_keyframe_linearize_blocking','us_per_window_association_blocking','cpp_seam')})
PY
python - <<'PY'
import time, numpy as np, lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=500_000, n_az=3125, half_extent=(150.0,150.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0],1),10.0,np.float32)],1)
ctx = L.Context(0); ex = L.RotExtractor(ctx, n_scans=64, ds_rate=4)
p = L.api.PinnedArray(raw.shape, np.float32); p.array[...] = raw
for name, fn in (("pinned", lambda: ex.extract(p.array, reuse=True)), ("pageable", lambda: ex.extract(raw))):
    for _ in range(3): fn()
    t=time.perf_counter()
    for _ in range(30): fn()
    print("extract_rot", name, (time.perf_counter()-t)/30*1e3, "ms")
PY
