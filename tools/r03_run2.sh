#!/bin/bash
# round 3, GPU run 2: kernel-trace stats of the small launches (true device durations, not host-paced loops)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b; mkdir -p $OUT
for cfg in "2000 0" "2000 1" "25000 0" "25000 1" "25000 8" "200000 1"; do
  set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s_$1_$2 -- python tools/coop_profile.py $1 $2 > /dev/null 2> $OUT/s_$1_$2.err
  echo "== n=$1 lanes=$2"; python tools/kstats.py $OUT/s_$1_$2_kernel_stats.csv 2>/dev/null | head -8 || head -8 $OUT/s_$1_$2_kernel_stats.csv
done
