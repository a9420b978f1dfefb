#!/bin/bash
# Collects the per-round profile set on the GPU box (run through gpurun):  bash tools/make_profiles.sh r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>/stats_kernel_stats.csv
# 2. PMC passes (separate runs, --kernel-trace only; gpurun forbids pmc + sys-trace) -> p*_counter_collection.csv
# 3. summary JSON incl. HBM traffic per launch of the dominant kernel -> gpurun_out/<tag>/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
# --no-extras: the secondary measurements (3-slot window, ROT extractor) launch the same kernels concurrently from three streams and
# would skew the per-launch averages this trace is compared with
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python bench.py --steps 200 --warmup 20 --no-extras > $OUT/bench.json 2> $OUT/bench.err
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o p_fetch -- $B > /dev/null 2> $OUT/p_fetch.err
timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT -o p_write -- $B > /dev/null 2> $OUT/p_write.err
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o p_sq -- $B > /dev/null 2> $OUT/p_sq.err
timeout 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $OUT -o p_sq2 -- $B > /dev/null 2> $OUT/p_sq2.err
python tools/profile_summary.py $OUT
