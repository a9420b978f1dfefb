#!/bin/bash
# VERDICT r3 #3, the measurement behind the decision: what would "a search that fills the chip + a fit that runs once" cost?
#   one lane per query (k_associate_surf) and two lanes per query (k_associate_coop<2>), each with the fit and with the fit switched off
#   (LILI_DEBUG bit 65536: the gate on the fifth neighbour only; switched on AFTER the ten iterations that bring the pose to convergence, so that every
#   variant is timed at the same pose) — 200 k queries vs the 5 M-point map, converged pose, launches back to back.
# search_L1 = the kernel without the fit, search_L2 = the same with two lanes per query; fit ~ full_L1 - search_L1.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r04}/assoc_split.txt; mkdir -p $(dirname $OUT); : > $OUT
for lpq in 1 2 4; do for bits in 0 65536; do
  r=$(LILI_DEBUG_AFTER=$bits timeout 300 python bench.py --assoc-only 300 --assoc-after 10 --opt assoc_lpq=$lpq --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "lanes_per_query=$lpq LILI_DEBUG(timed launches)=$bits $r" | tee -a $OUT
done; done
