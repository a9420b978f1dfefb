"""Keyframe pipeline probe (through gpurun, optionally under rocprofv3 --kernel-trace): map rebuild per keyframe, blocking vs lili_map_set_begin/_end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth
import bench
w = synth.make_workload(n_map=bench.N_MAP, n_az=bench.N_AZ, half_extent=(460.0, 380.0))
P = L.make_params("rot")
ctx = L.Context(0)
m = L.ScanToMapMatcher(ctx, P)
d_map = torch.from_numpy(np.ascontiguousarray(w["map_xyz"])).cuda()
cloud = L.api.cloud_from_device(d_map.data_ptr(), w["map_xyz"].shape[0], 12, -1)
m.map_focus(w["lidar_t"], 143.0)
m.set_input_cloud(L.KIND_SURF, cloud)
m.set_queries(0, L.KIND_SURF, np.ascontiguousarray(w["scan_xyz"]))
tb, qb = bench.body_pose_for_lidar(L, P, w["lidar_t"])
tp, qp = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
m.pose_set(1, tp, qp)
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
n_kf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def loop():
    for k in range(n_kf):
        m.iterate_restart(0, 30, 10, 1, L.MASK_SURF)
        if mode == "pipe":
            m.set_input_cloud_begin(L.KIND_SURF, cloud); m.set_input_cloud_end(L.KIND_SURF)
        else:
            m.set_input_cloud(L.KIND_SURF, cloud)
    ctx.sync()
loop()
tic = time.perf_counter(); loop(); el = time.perf_counter() - tic
print(mode, "ms per keyframe", el / n_kf * 1e3)
