#!/bin/bash
# ROT / Livox extractor check through gpurun: parity tests, then per-kernel times (rocprofv3 kernel trace of tools/time_extract.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-rot}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_extract_rot_gpu.py tests/test_extract_livox_gpu.py tests/test_reference_gpu.py tests/test_config0_gpu.py tests/test_fullsize_gpu.py tests/test_voxel_gpu.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ex -- python tools/time_extract.py > $OUT/time.log 2>&1
grep -v "^\[synth\]\|amdgpu.ids" $OUT/time.log | tail -8
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/ex_kernel_stats.csv")))
for r in rows[:22]:
    n = r["Name"].split("(")[0].replace("void ", "").replace("lili::", "")
    print(f"  {n[:44]:46s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us  {float(r['Percentage']):6.2f} %")
PY
