# GPU check of the dense-map association: parity test, kernel trace of the 2B probe, wall time of back-to-back launches.  usage: bash tools/dense_check.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-dense}; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_dense_map_gpu.py -x -q -s 2>&1 | tail -5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T -o p2b -- python tools/dense_2b_probe.py > gpurun_out/$T/probe.jsonl 2> gpurun_out/$T/prof.err
head -1 gpurun_out/$T/probe.jsonl | cut -c1-200; tail -2 gpurun_out/$T/probe.jsonl
python tools/kstats.py gpurun_out/$T/p2b_kernel_stats.csv | head -4
python tools/dense_2b_probe.py quick=40 2>/dev/null | tail -2
python - <<PY
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/$T/p2b_kernel_trace.csv'))]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'k_associate_fine' in r['Kernel_Name']]
print('k_associate_fine per launch (us), first two iterations of the probe:', [round(x,1) for x in d[:14]])
print('resources', [(r['VGPR_Count'],r['SGPR_Count'],r['LDS_Block_Size'],r['Scratch_Size']) for r in rows if 'k_associate_fine' in r['Kernel_Name']][0])
PY
