#!/bin/bash
# two ranks of bench.py on ONE GPU (through gpurun): the N > 1 code path of the bench and of lili_s2m_iterate_sharded end to end
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-share2}; mkdir -p $OUT
export LILI_BENCH_SHARE_GPU=1
for coll in auto torch; do
  for sc in strong weak; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 10 \
        --collective $coll --scaling $sc --no-extras > $OUT/b_${coll}_${sc}.json 2> $OUT/b_${coll}_${sc}.err
    echo "== $coll $sc rc=$?"; grep "collectives\|host enqueue" $OUT/b_${coll}_${sc}.err | tail -2
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/b_${coll}_${sc}.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["scaling"], d["config"]["collectives"], d["final_pose"], {k: v for k, v in d.items() if k.startswith("weak")})
except Exception as e:
    print("failed", e); print(open("$OUT/b_${coll}_${sc}.err").read()[-2000:])
PY
  done
done
