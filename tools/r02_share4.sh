#!/bin/bash
# four ranks of bench.py on ONE GPU (through gpurun): the world > 2 logic of the P2P exchange (slots, flags, rank-order sums)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-share4}; mkdir -p $OUT
export LILI_BENCH_SHARE_GPU=1
for n in 4 8; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus $n --steps 60 --warmup 10 \
      --collective auto --no-extras > $OUT/b$n.json 2> $OUT/b$n.err
  echo "== world $n rc=$?"; grep "collectives\|host enqueue" $OUT/b$n.err | tail -2
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/b$n.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["scaling"], d["config"]["collectives"], d["final_pose"]["t"], d["final_pose"]["gn_status"])
except Exception as e:
    print("failed", e); print(open("$OUT/b$n.err").read()[-2500:])
PY
done
