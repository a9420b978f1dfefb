#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-s9}; mkdir -p $OUT
for mode in 0; do
  LILI_S9_MODE=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m$mode -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/b$mode.json 2> $OUT/b$mode.err
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/m${mode}_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("void ", "").replace("lili::", "")
    if any(k in n for k in ("k_scatter9", "k_start9")):
        print("mode $mode", f"{n[:40]:32s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.2f} us min {float(r['MinNs'])/1e3:9.2f} max {float(r['MaxNs'])/1e3:9.2f}")
PY
done
