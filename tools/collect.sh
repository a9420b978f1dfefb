#!/bin/bash
# copies the judged files of the round-4 evidence run (tools/r04_final.sh -> gpurun_out/r04x) into profiles/ under r04_* names
set -e
cd "$(dirname "$0")/.."
S=${1:-gpurun_out/r04x}; D=profiles
cp $S/stats_kernel_stats.csv        $D/r04_kernel_stats.csv
cp $S/pmc_traffic.json              $D/r04_pmc_traffic.json
cp $S/pmc_traffic.json              $D/pmc_traffic.json
cp $S/profile_summary.txt           $D/r04_profile_summary.txt
cp $S/bench.json                    $D/r04_bench.json
cp $S/bench_driver_cmd.json         $D/r04_bench_driver_command.json
cp $S/small_2000_kernel_stats.csv   $D/r04_small_launches_2000_kernel_stats.csv
cp $S/small_25000_kernel_stats.csv  $D/r04_small_launches_25000_kernel_stats.csv
cp $S/localmap_kernel_stats.csv     $D/r04_localmap_kernel_stats.csv
cp $S/lm_kernel_stats.csv           $D/r04_lm_kernel_stats.csv
cp $S/lm_time.log                   $D/r04_lm_time.jsonl
cp $S/extract_kernel_stats.csv      $D/r04_extract_kernel_stats.csv
cp $S/rot_phases.log                $D/r04_extract_phases.txt
cp $S/iter_time.json                $D/r04_iter_time.json
[ -f $S/window_seam.json ] && cp $S/window_seam.json $D/r04_window_seam_cpp.json
[ -f $S/k7_kernel_stats.csv ] && cp $S/k7_kernel_stats.csv $D/r04_k7_kernel_stats.csv
[ -f $S/k7_time.jsonl ] && cp $S/k7_time.jsonl $D/r04_k7_time.jsonl
grep -E "passed|failed" $S/pytest_all.log | tail -1 > $D/r04_gpu_tests.txt
ls -la $D/r04_*
