# VALU/VMEM instruction counts of the association kernel per debug-bit variant at a fixed pose
# (bit 1: skip fit, bit 2: skip search).  usage: bash tools/assoc_pmc.sh <tag> <after_iters> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-apmc}; AFTER=${2:-0}; shift; shift
mkdir -p gpurun_out/$TAG
for d in 0 1 2 3; do
  LILI_DEBUG_AFTER=$d python bench.py --assoc-only 50 --assoc-after $AFTER "$@" 2>/dev/null | tail -1 | sed "s/^/debug $d /"
  LILI_DEBUG_AFTER=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/$TAG -o d$d -- python bench.py --assoc-only 10 --assoc-after $AFTER "$@" > /dev/null 2> gpurun_out/$TAG/d$d.err
  python tools/pmc_summary.py gpurun_out/$TAG/d${d}_counter_collection.csv --last 10 | grep associate
done
