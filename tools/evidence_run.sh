#!/bin/bash
# round 4, evidence run: full GPU suite, the committed profile set (kernel trace + PMC passes of the default bench command), small launches, device LM,
# local map, ROT extractor, the C++ window seam, and the bench line itself.  Outputs under gpurun_out/r04x; tools/r04_collect.sh copies the judged ones to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04x; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -4 $OUT/pytest_all.log
bash tools/make_profiles.sh r04x > $OUT/profile_summary.txt 2>&1
tail -24 $OUT/profile_summary.txt
for cfg in "2000 0" "25000 0"; do
  set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o small_$1 -- python tools/coop_profile.py $1 $2 > /dev/null 2> $OUT/small_$1.err
  echo "== small launches n=$1"; python tools/kstats.py $OUT/small_$1_kernel_stats.csv | head -6
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o localmap -- python tools/localmap_loop.py 1 > /dev/null 2> $OUT/localmap.err
echo "== local map step"; python tools/kstats.py $OUT/localmap_kernel_stats.csv | head -10
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lm -- python tools/lm_time.py > $OUT/lm_time.log 2> $OUT/lm.err
echo "== device LM"; python tools/kstats.py $OUT/lm_kernel_stats.csv | head -6; cat $OUT/lm_time.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
tail -c 300 $OUT/bench.json; echo
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench (driver command) rc $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o extract -- python tools/rot_phases.py > $OUT/rot_phases.log 2>&1
echo "== ROT extractor"; python tools/kstats.py $OUT/extract_kernel_stats.csv | head -8; grep "blocking call" $OUT/rot_phases.log
timeout 300 python tools/iter_time.py $OUT/iter_time.json > $OUT/iter_time.log 2>&1; tail -6 $OUT/iter_time.log
timeout 120 tools/_probe/launch_floor 2000 > $OUT/launch_floor.txt 2>&1
timeout 300 python tools/k7_time.py > $OUT/k7_time.jsonl 2> $OUT/k7_time.err; echo "== map index build"; cat $OUT/k7_time.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k7 -- python tools/k7_time.py > /dev/null 2> $OUT/k7_prof.err
python tools/kstats.py $OUT/k7_kernel_stats.csv | head -12
