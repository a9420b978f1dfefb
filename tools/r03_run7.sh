#!/bin/bash
# round 3, GPU run 7: count barrier (ROT small launches), keyframe pool + sync-free keyframe sort, full suite, local-map breakdown, small-launch sweep
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03g; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_coop_gpu.py tests/test_voxel_gpu.py -m gpu -q ) > $OUT/pytest_new.log 2>&1
tail -30 $OUT/pytest_new.log
( time timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_coop_gpu.py --deselect tests/test_voxel_gpu.py ) > $OUT/pytest_all.log 2>&1
tail -8 $OUT/pytest_all.log
cat > /tmp/lm_loop.py <<PY
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import lili_om_amd as L
from lili_om_amd import synth
inc = int(sys.argv[1])
w = synth.make_workload(n_map=200_000, half_extent=(150.0, 150.0), verbose=False)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
ctx.set_option("localmap_incremental", inc)
feats = np.ascontiguousarray(np.concatenate([w["scan_xyz"][::10], np.zeros((w["scan_xyz"][::10].shape[0], 1), np.float32)], 1))
lm = L.api.LocalMap(ctx, L.KIND_SURF, width=50, leaf=0.4, max_sq_radius=1.0)
for k in range(50): lm.push(feats, [0.8 * k, 0.1 * k, 0.0], [1.0, 0.0, 0.0, 0.0])
n_raw, n_map = lm.commit()
kf = [0]; tp = [0.0]; tc = [0.0]
def one():
    kf[0] += 1
    a = time.perf_counter()
    lm.push(feats, [0.8 * (50 + kf[0]), 0.1 * (50 + kf[0]), 0.0], [1.0, 0.0, 0.0, 0.0])
    b = time.perf_counter()
    lm.commit()
    c = time.perf_counter()
    tp[0] += b - a; tc[0] += c - b
for _ in range(3): one()
tp[0] = tc[0] = 0.0
torch.cuda.synchronize(); tic = time.perf_counter()
for _ in range(20): one()
torch.cuda.synchronize()
print("LOCALMAP incremental=%d: %.4f ms per keyframe (push %.4f + commit %.4f host-side) (%d -> %d points) stats %s" % (inc, (time.perf_counter() - tic) / 20 * 1e3, tp[0] / 20 * 1e3, tc[0] / 20 * 1e3, n_raw, n_map, lm.stats()))
ctx.close()
PY
for inc in 1 0; do python /tmp/lm_loop.py $inc; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lmloop -- python /tmp/lm_loop.py 1 > /dev/null 2> $OUT/lmloop.err
python tools/kstats.py $OUT/lmloop_kernel_stats.csv | head -30
python - <<PY
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=5_000_000, half_extent=(460.0, 380.0), verbose=False)
order = np.argsort(w["scan_ring"], kind="stable"); scan = np.ascontiguousarray(w["scan_xyz"][order])
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
P = L.make_params("rot"); m = L.ScanToMapMatcher(ctx, P)
m.map_focus(w["lidar_t"], float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0)
m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
m.pose_set(1, t0, q0)
for n in (2000, 5000, 8000, 10000, 20000):
    q = np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n]); m.set_queries(0, L.KIND_SURF, q)
    row = {"n": n}
    for cb in (1, 0):
        ctx.set_option("count_barrier", cb)
        m.iterate_restart(0, 20, 10, 1, L.MASK_SURF); torch.cuda.synchronize(); tic = time.perf_counter()
        m.iterate_restart(0, 400, 10, 1, L.MASK_SURF); torch.cuda.synchronize()
        row["cb%d_us" % cb] = round((time.perf_counter() - tic) / 400 * 1e6, 2); row["st%d" % cb] = int(m.pose_get(0)[2]); row["dt%d" % cb] = float(np.linalg.norm(m.pose_get(0)[0] - tb))
    print("ROTSMALL", json.dumps(row))
ctx.close()
PY
