#!/bin/bash
# K7 (map index build) check through gpurun: exactness tests, then per-kernel times of one bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-k7}; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_knn_stress_gpu.py tests/test_s2m_gpu.py tests/test_voxel_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
for lb in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k7_$lb -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt scan_lookback=$lb > $OUT/bench_$lb.json 2> $OUT/bench_$lb.err
  echo "== scan_lookback=$lb"
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/k7_${lb}_kernel_stats.csv")))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("lili::", "")
    if any(k in n for k in ("k_cloud_to_f4", "k_bbox", "k_cell_count", "k_scan", "k_scatter", "fillBuffer", "k_vox")):
        print(f"  {n[:40]:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e3:9.1f} us")
d = json.loads(open("$OUT/bench_$lb.json").read().strip().splitlines()[-1])
print("  value", d["value"], "map_index_build", d["extras"]["map_index_build"]["ms"], "ms", "map_index_build_s", d["config"]["map_index_build_s"])
PY
done
