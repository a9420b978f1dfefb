#!/bin/bash
# super-row layout check (through gpurun): exactness tests, bench line, phase probe, K7 kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-sr}; OUT=gpurun_out/$TAG; mkdir -p $OUT
SEL=${2:-"tests/test_s2m_gpu.py tests/test_knn_stress_gpu.py tests/test_dense_map_gpu.py"}
if [ "$SEL" != "none" ]; then
  ( timeout 900 python -m pytest $SEL -m gpu -x -q ) > $OUT/pytest.log 2>&1
  grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
fi
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
timeout 300 $B > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("BENCH", d["value"], "it/s", d["ms_per_step"], "ms  assoc", d["roofline"]["us_per_launch"], "us  inner", d.get("inner_iteration", {}).get("us_per_iteration"), "us  pose", d["final_pose"]["t"], "build_s", d["config"]["map_index_build_s"])
except Exception as e:
    print("bench failed", e); print(open("$OUT/bench.err").read()[-1500:])
PY
if [ "$3" = "probe" ]; then
  timeout 300 bash tools/assoc_phases.sh run 10 > $OUT/phases.txt 2>&1; tail -30 $OUT/phases.txt
fi
if [ "$4" = "k7" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k7 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k7.json 2> $OUT/bench_k7.err
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/k7_kernel_stats.csv")))
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("lili::", "")
    if any(k in n for k in ("k_cloud_to_f4", "k_bbox", "k_cell_count", "k_scan", "k_scatter", "k_start9", "fillBuffer", "k_assoc")):
        print(f"  {n[:40]:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e3:9.1f} us")
d = json.loads(open("$OUT/bench_k7.json").read().strip().splitlines()[-1])
print("  value", d["value"], "map_index_build", d["extras"]["map_index_build"]["ms"], "ms", "map_index_build_s", d["config"]["map_index_build_s"])
PY
fi
