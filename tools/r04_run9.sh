#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for nc in 1 0; do
  echo "nn_cache=$nc assoc-only at the converged pose:"; timeout 300 python bench.py --assoc-only 300 --assoc-after 10 --opt nn_cache=$nc --no-cpu-baseline --no-extras 2>/dev/null | tail -1
done
for nc in 1 0; do
( cd /tmp && rm -rf /tmp/kp$nc && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kp$nc -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-extras --opt nn_cache=$nc > /tmp/kp$nc.log 2>&1 )
f=$(find /tmp/kp$nc -name "*kernel_trace.csv" | head -1)
python - "$f" $nc <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000) for r in rows if "k_associate" in r["Kernel_Name"]]
print("nn_cache=%s: association launches in order (us):" % sys.argv[2])
print(" ".join(f"{n.split('<')[0].replace('void lili::','')[:18]}:{d:.1f}" for n,d in seq[:45]))
PY
done
