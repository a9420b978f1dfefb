# PMC passes (separate runs, kernel trace only) over the ROT extractor (tools/rot_phases.py): instruction mix, waits and HBM traffic per kernel launch;  bash tools/pmc_rot.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-pmcrot}; mkdir -p gpurun_out/$T
B="python tools/rot_phases.py"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/$T -o r1 -- $B > /dev/null 2> gpurun_out/$T/r1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$T -o r2 -- $B > /dev/null 2> gpurun_out/$T/r2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$T -o r3 -- $B > /dev/null 2> gpurun_out/$T/r3.err
python tools/pmc_summary.py gpurun_out/$T/r*_counter_collection.csv | grep k_rot
