# PMC passes (separate runs, --kernel-trace only) for bench.py; usage: bash tools/pmc_passes.sh <outdir-tag> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-pmc}; shift
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline $*"
mkdir -p gpurun_out/$TAG
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/$TAG -o p1 -- $B > /dev/null 2> gpurun_out/$TAG/p1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG -o p2 -- $B > /dev/null 2> gpurun_out/$TAG/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$TAG -o p3 -- $B > /dev/null 2> gpurun_out/$TAG/p3.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d gpurun_out/$TAG -o p4 -- $B > /dev/null 2> gpurun_out/$TAG/p4.err
python tools/pmc_summary.py gpurun_out/$TAG/p*_counter_collection.csv | grep -E "associate|linearize|reduce"
