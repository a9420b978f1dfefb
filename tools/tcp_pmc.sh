# L1 (TA/TCP) pressure of the association kernel at the converged pose (2 counters per pass, each pass under its own timeout)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-tcp}; shift
mkdir -p gpurun_out/$TAG
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/$TAG -o t$i -- python bench.py --assoc-only 10 --assoc-after 5 "$@" > /dev/null 2> gpurun_out/$TAG/t$i.err || echo "pass $i failed"
  python tools/pmc_summary.py gpurun_out/$TAG/t${i}_counter_collection.csv --last 10 2>/dev/null | grep associate
done
