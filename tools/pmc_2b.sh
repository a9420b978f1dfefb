# PMC passes (separate runs, kernel trace only) over tools/dense_2b_probe.py: bash tools/pmc_2b.sh <tag> [probe args]; rows are summarised per phase by tools/pmc_2b_summary.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-pmc2b}; shift
B="python tools/dense_2b_probe.py $*"
mkdir -p gpurun_out/$TAG
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/$TAG -o p1 -- $B > /dev/null 2> gpurun_out/$TAG/p1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG -o p2 -- $B > /dev/null 2> gpurun_out/$TAG/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$TAG -o p3 -- $B > /dev/null 2> gpurun_out/$TAG/p3.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/$TAG -o p4 -- $B > /dev/null 2> gpurun_out/$TAG/p4.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d gpurun_out/$TAG -o p5 -- $B > /dev/null 2> gpurun_out/$TAG/p5.err
python tools/pmc_2b_summary.py gpurun_out/$TAG
