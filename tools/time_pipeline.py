"""End-to-end, device-resident: 200k-point scan (already in HBM) -> ROT extraction -> VoxelGrid(0.4) of the surf
features -> 10 outer scan-to-map iterations against the 5M-point map.  Prints scans/s."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import lili_om_amd as L
from lili_om_amd import synth
n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
w = synth.make_workload(n_map=n_map, half_extent=(460.0, 380.0) if n_map >= 4_000_000 else (150.0, 150.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 9.0, np.float32)], 1).astype(np.float32)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
P = L.make_params("rot")
tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
m = L.ScanToMapMatcher(ctx, P); m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
for ds in (4, 1):
    ex = L.RotExtractor(ctx, ds_rate=ds)
    d_scan = torch.from_numpy(raw).cuda()
    cloud = L.api.cloud_from_device(d_scan.data_ptr(), raw.shape[0], 16, 12)
    d_q = torch.empty((raw.shape[0], 4), dtype=torch.float32, device="cuda")
    qi = np.array([1.0, 0, 0, 0]); ql = np.array(list(P.q_lb))
    outs = [L.api.FeatureOut(None, 0, 16, 0, 0) for _ in range(3)]
    def one():
        ctx._chk(ctx.lib.lili_extract_rot(ctx.h, C.byref(cloud), qi.ctypes.data_as(C.c_void_p), ql.ctypes.data_as(C.c_void_p), C.byref(ex.params), C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        _, _, d_surf = L.api.extract_rot_device(ctx)
        qd = L.api.voxel_filter_device(ctx, d_surf, 0.4, d_q.data_ptr(), raw.shape[0])
        m.set_queries(0, L.KIND_SURF, qd)
        m.pose_set(0, t0, q0); m.iterate(0, 10, L.MASK_SURF)
        return m.pose_get(0), qd.n
    one()
    t = time.perf_counter()
    for _ in range(20): (tt, qq, st), nq = one()
    dt = (time.perf_counter() - t) / 20
    print(f"ds_rate {ds}: {dt*1e3:.3f} ms/scan = {1/dt:.0f} scans/s  ({nq} surf queries after VoxelGrid, pose err {np.linalg.norm(tt-tb):.4f} m, status {st})")
