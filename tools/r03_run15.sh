#!/bin/bash
# round 3, GPU run 15: full suite, the committed profile set (kernel trace + PMC passes of the default bench command), kernel stats of the small launches /
# the device LM / the local-map step, and the bench line itself
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03x; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
tail -6 $OUT/pytest_all.log
bash tools/make_profiles.sh r03x > $OUT/profile_summary.txt 2>&1
tail -30 $OUT/profile_summary.txt
for cfg in "2000 0" "25000 0"; do
  set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o small_$1 -- python tools/coop_profile.py $1 $2 > /dev/null 2> $OUT/small_$1.err
  echo "== small launches n=$1"; python tools/kstats.py $OUT/small_$1_kernel_stats.csv | head -6
done
for inc in 1 0; do python tools/localmap_loop.py $inc; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o localmap -- python tools/localmap_loop.py 1 > /dev/null 2> $OUT/localmap.err
echo "== local map step"; python tools/kstats.py $OUT/localmap_kernel_stats.csv | head -14
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lm -- python tools/lm_time.py > $OUT/lm_time.log 2> $OUT/lm.err
echo "== device LM"; python tools/kstats.py $OUT/lm_kernel_stats.csv | head -8; cat $OUT/lm_time.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o extract -- python tools/rot_phases.py > $OUT/rot_phases.log 2>&1
echo "== ROT extractor"; python tools/kstats.py $OUT/extract_kernel_stats.csv | head -8; grep "blocking call" $OUT/rot_phases.log
timeout 300 python tools/iter_time.py $OUT/iter_time.json > $OUT/iter_time.log 2>&1; tail -12 $OUT/iter_time.log
