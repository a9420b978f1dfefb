# HBM traffic of the 5 M-point index build (K7) kernel by kernel: two PMC passes (FETCH_SIZE; WRITE_SIZE) over tools/k7_time.py's voxel-order / guessed-box builds, base
# index and focused super-row copy.  usage: bash tools/k7_pmc.sh <tag>      -> gpurun_out/<tag>/k7_traffic.json  (MB per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-k7}; mkdir -p gpurun_out/$T
K7_ONLY=voxel python tools/k7_time.py > gpurun_out/$T/k7_time.jsonl 2> gpurun_out/$T/k7_time.err; cat gpurun_out/$T/k7_time.jsonl
K7_ONLY=voxel rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$T -o kf -- python tools/k7_time.py > /dev/null 2>&1
K7_ONLY=voxel rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$T -o kw -- python tools/k7_time.py > /dev/null 2>&1
K7_ONLY=voxel rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T -o ks -- python tools/k7_time.py > /dev/null 2>&1
python - <<PY
import csv, json, collections
T="gpurun_out/$T"
def per(path, name):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"]==name: acc[r["Kernel_Name"].split("(")[0].replace("void ","")].append(float(r["Counter_Value"]))
    return acc
f, w = per(T+"/kf_counter_collection.csv","FETCH_SIZE"), per(T+"/kw_counter_collection.csv","WRITE_SIZE")
out={}
for k in sorted(set(f)|set(w)):
    # the LAST launches are the guessed-box builds with the focused copy; report the mean over all launches of the run and the last 20
    fv, wv = f.get(k,[0]), w.get(k,[0])
    out[k]={"launches": len(fv), "mb_per_launch_mean": round((2*sum(fv)/len(fv)+sum(wv)/len(wv))*1024/1e6,1), "mb_per_launch_last20": round((2*sum(fv[-20:])/len(fv[-20:])+sum(wv[-20:])/len(wv[-20:]))*1024/1e6,1)}
json.dump(out, open(T+"/k7_traffic.json","w"), indent=1); print(json.dumps(out, indent=1))
PY
python tools/kstats.py gpurun_out/$T/ks_kernel_stats.csv | head -12
