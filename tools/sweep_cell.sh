#!/bin/bash
# cell-size sweep of the map index (through gpurun): reach / cell_pct vs iteration rate and association time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "$@"; do set -- $cfg
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --reach $1 --cell-pct $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reach $1 pct $2:', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['final_pose']['t'][0], d['config']['map_index_build_s'])"
done
