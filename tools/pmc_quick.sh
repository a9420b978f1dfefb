# three PMC passes over `python tools/dense_2b_probe.py quick=10` (the last 10 launches of k_associate_fine are at the perturbed start); usage: bash tools/pmc_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-pq}; mkdir -p gpurun_out/$T
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/$T -o q1 -- python tools/dense_2b_probe.py quick=10 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/$T -o q2 -- python tools/dense_2b_probe.py quick=10 > /dev/null 2>&1
rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d gpurun_out/$T -o q3 -- python tools/dense_2b_probe.py quick=10 > /dev/null 2>&1
echo "perturbed start:"; python tools/pmc_summary.py --last 10 gpurun_out/$T/q*_counter_collection.csv | grep fine
