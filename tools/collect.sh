#!/bin/bash
# copies the judged files of an evidence run (tools/evidence_run.sh <tag> -> gpurun_out/<tag>) into profiles/ under <round>_* names:  bash tools/collect.sh r05x r05
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/${1:-r06x}; R=${2:-r06}; D=profiles
cp $S/stats_kernel_stats.csv        $D/${R}_kernel_stats.csv
cp $S/pmc_traffic.json              $D/${R}_pmc_traffic.json
cp $S/pmc_traffic.json              $D/pmc_traffic.json
cp $S/profile_summary.txt           $D/${R}_profile_summary.txt
cp $S/bench.json                    $D/${R}_bench.json
cp $S/bench_driver_cmd.json         $D/${R}_bench_driver_command.json
cp $S/frame_kernel_stats.csv        $D/${R}_frame_pipeline_kernel_stats.csv
cp $S/frame_bench.json              $D/${R}_frame_pipeline_bench.json
cp $S/small_2000_kernel_stats.csv   $D/${R}_small_launches_2000_kernel_stats.csv
cp $S/small_25000_kernel_stats.csv  $D/${R}_small_launches_25000_kernel_stats.csv
cp $S/localmap_kernel_stats.csv     $D/${R}_localmap_kernel_stats.csv
cp $S/lm_kernel_stats.csv           $D/${R}_lm_kernel_stats.csv
cp $S/lm_time.log                   $D/${R}_lm_time.jsonl
cp $S/extract_kernel_stats.csv      $D/${R}_extract_kernel_stats.csv
cp $S/rot_phases.log                $D/${R}_extract_phases.txt
cp $S/k7_kernel_stats.csv           $D/${R}_k7_kernel_stats.csv
cp $S/k7_time.jsonl                 $D/${R}_k7_time.jsonl
[ -f $S/anyorder_probe.txt ] && cp $S/anyorder_probe.txt $D/${R}_anyorder_probe.txt
grep -v "synth\|bench\]\|amdgpu" $S/iteration_phases.txt > $D/${R}_iteration_phases.txt || true
grep -E "passed|failed" $S/pytest_all.log | tail -1 > $D/${R}_gpu_tests.txt
# round 6: configs[2] variant B and the index build's traffic
[ -f $S/dense2b_kernel_stats.csv ] && cp $S/dense2b_kernel_stats.csv $D/${R}_2B_kernel_stats.csv
[ -f $S/dense2b_probe.jsonl ] && cp $S/dense2b_probe.jsonl $D/${R}_2B_probe.jsonl
[ -f $S/dense2b_pmc.json ] && cp $S/dense2b_pmc.json $D/${R}_2B_pmc.json
[ -f $S/randrun_probe.txt ] && cp $S/randrun_probe.txt $D/${R}_randrun_probe.txt
[ -f $S/k7_traffic.json ] && cp $S/k7_traffic.json $D/${R}_k7_traffic.json
[ -f $S/config0_timeline.txt ] && cp $S/config0_timeline.txt $D/${R}_config0_timeline.txt
ls -la $D/${R}_*
