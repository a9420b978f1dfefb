#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nn_keep_gpu.py tests/test_s2m_gpu.py tests/test_coop_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > gpurun_out/r04/keep_tests2.log 2>&1; echo "tests rc $?" > gpurun_out/r04/summary11.txt
tail -15 gpurun_out/r04/keep_tests2.log
for nc in 1 0; do
( cd /tmp && rm -rf /tmp/kp$nc && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kp$nc -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-extras --opt nn_cache=$nc > /tmp/kp$nc.log 2>&1 )
f=$(find /tmp/kp$nc -name "*kernel_trace.csv" | head -1)
python - "$f" $nc <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000) for r in rows if "k_associate" in r["Kernel_Name"]]
print("nn_cache=%s: association launches in order (us):" % sys.argv[2])
print(" ".join(f"{d:.1f}" for n,d in seq[:40]))
PY
done
for nc in 1 0; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --opt nn_cache=$nc > gpurun_out/r04/bench11_nc$nc.json 2> gpurun_out/r04/bench11_nc$nc.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench11_nc$nc.json').read().strip().splitlines()[-1])
print('nn_cache=$nc', d['value'], 'it/s', d['ms_per_step'], 'assoc us', d['roofline']['us_per_launch'], 'pose', d['final_pose']['t'], d['final_pose']['q'])
PY
done
cat gpurun_out/r04/summary11.txt
