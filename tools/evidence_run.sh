#!/bin/bash
# Evidence run of a round (through gpurun): bash tools/evidence_run.sh r05x
# Full GPU suite, the committed profile set (kernel trace + PMC passes of the default bench command: tools/make_profiles.sh), the bench line of the plain run and of the
# driver's command, kernel traces of the front-end frame pipeline (configs[1]), small launches, device LM, local map, ROT extractor, index build, the any-order launch probe.
# Outputs under gpurun_out/<tag>; tools/collect.sh <tag> <round> copies the judged ones to profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06x}
OUT=gpurun_out/$TAG; mkdir -p $OUT tools/_probe
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_all.log 2>&1
grep -E "passed|failed" $OUT/pytest_all.log | tail -2
bash tools/make_profiles.sh $TAG > $OUT/profile_summary.txt 2>&1
tail -24 $OUT/profile_summary.txt
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench (driver command) rc $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o frame -- python bench.py --config 1 --no-cpu-baseline > $OUT/frame_bench.json 2> $OUT/frame.err
echo "== front-end frame pipeline (configs[1])"; python tools/kstats.py $OUT/frame_kernel_stats.csv | head -16
for cfg in "2000 0" "25000 0"; do
  set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o small_$1 -- python tools/coop_profile.py $1 $2 > /dev/null 2> $OUT/small_$1.err
  echo "== small launches n=$1"; python tools/kstats.py $OUT/small_$1_kernel_stats.csv | head -6
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o localmap -- python tools/localmap_loop.py 1 > /dev/null 2> $OUT/localmap.err
echo "== local map step"; python tools/kstats.py $OUT/localmap_kernel_stats.csv | head -10
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lm -- python tools/lm_time.py > $OUT/lm_time.log 2> $OUT/lm.err
echo "== device LM"; python tools/kstats.py $OUT/lm_kernel_stats.csv | head -6; cat $OUT/lm_time.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o extract -- python tools/rot_phases.py > $OUT/rot_phases.log 2>&1
echo "== ROT extractor"; python tools/kstats.py $OUT/extract_kernel_stats.csv | head -8; grep "blocking call" $OUT/rot_phases.log
timeout 300 python tools/k7_time.py > $OUT/k7_time.jsonl 2> $OUT/k7_time.err; echo "== map index build"; cat $OUT/k7_time.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k7 -- python tools/k7_time.py > /dev/null 2> $OUT/k7_prof.err
python tools/kstats.py $OUT/k7_kernel_stats.csv | head -12
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/_probe/anyorder_probe > /dev/null 2>&1 && timeout 120 tools/_probe/anyorder_probe > $OUT/anyorder_probe.txt 2>&1
LILI_PHASES=1 timeout 200 python bench.py --no-extras --no-cpu-baseline 2> $OUT/iteration_phases.txt > /dev/null; grep -v "synth\|bench\]\|amdgpu" $OUT/iteration_phases.txt | tail -14
# ---- round 6: configs[2] variant B (dense map) launch by launch, its PMC passes, the memory-pattern probe behind its roofline, the index build's traffic kernel by kernel
echo "== dense map (configs[2] variant B)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dense2b -- python tools/dense_2b_probe.py > $OUT/dense2b_probe.jsonl 2> $OUT/dense2b.err
tail -2 $OUT/dense2b_probe.jsonl; python tools/kstats.py $OUT/dense2b_kernel_stats.csv | head -6
bash tools/pmc_2b.sh $TAG/pmc2b > $OUT/dense2b_pmc.json 2> $OUT/dense2b_pmc.err; tail -40 $OUT/dense2b_pmc.json | head -60
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/randrun_probe.hip -o tools/_probe/randrun_probe > /dev/null 2>&1 && timeout 120 tools/_probe/randrun_probe > $OUT/randrun_probe.txt 2>&1; cat $OUT/randrun_probe.txt
echo "== index build traffic"
bash tools/k7_pmc.sh $TAG/k7pmc > $OUT/k7_pmc.log 2>&1; cp gpurun_out/$TAG/k7pmc/k7_traffic.json $OUT/k7_traffic.json 2>/dev/null; tail -12 $OUT/k7_pmc.log
echo "== frame pipeline probe"; python tools/frame_probe.py 2>/dev/null | tail -1
echo "== back-end keyframe"; ./examples/backend_demo 60 2500 250 40 3
# ---- round 6, second half: BASELINE configs[0] as one call (guessed feature counts): kernel timeline of the steady state
echo "== configs[0] timeline"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o c0 -- python bench.py --config 0 --no-cpu-baseline > $OUT/c0_bench.json 2> $OUT/c0.err
python tools/trace_gaps.py $OUT/c0_kernel_trace.csv > $OUT/config0_timeline.txt; cat $OUT/config0_timeline.txt

