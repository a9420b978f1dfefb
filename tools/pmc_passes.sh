cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc1 -o p1 -- $B > /dev/null 2> gpurun_out/pmc1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o p2 -- $B > /dev/null 2> gpurun_out/pmc2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p3 -- $B > /dev/null 2> gpurun_out/pmc3.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_ANY SQ_INSTS_VALU_ADD_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4 -o p4 -- $B > /dev/null 2> gpurun_out/pmc4.err
ls gpurun_out/pmc*/
