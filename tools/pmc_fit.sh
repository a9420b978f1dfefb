cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fitpmc
for d in 0 2048 16384 1; do
  LILI_DEBUG_AFTER=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace --output-format csv -d gpurun_out/fitpmc -o d$d -- python bench.py --assoc-only 10 --assoc-after 10 > /dev/null 2> gpurun_out/fitpmc/d$d.err
  python tools/pmc_summary.py gpurun_out/fitpmc/d${d}_counter_collection.csv --last 10 | grep associate | sed "s/^/d=$d /"
done
