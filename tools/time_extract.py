"""Timing driver of the two extractors and the voxel filter (run it under rocprofv3 --kernel-trace --stats)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=300_000, n_az=3125, half_extent=(150.0, 150.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10, np.float32)], 1)
ctx = L.Context(0)
for ds in (4, 1):
    ex = L.RotExtractor(ctx, ds_rate=ds)
    ex.extract(raw)
    t = time.perf_counter()
    for _ in range(20): r = ex.extract(raw)
    dt = (time.perf_counter() - t) / 20
    print(f"ROT ds_rate {ds}: {dt*1e3:.3f} ms/scan incl. H2D+D2H ({len(r['edge'])} edge, {len(r['surf'])} surf, {len(r['full'])} full)")
    if hasattr(ex, "extract_device"):
        d_raw = torch.from_numpy(raw).cuda()
        ex.extract_device(d_raw.data_ptr(), raw.shape[0]); ctx.sync()
        t = time.perf_counter()
        for _ in range(50): ex.extract_device(d_raw.data_ptr(), raw.shape[0])
        ctx.sync()
        print(f"ROT ds_rate {ds}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms/scan device-resident in and out")
ls = synth.make_livox_scan(3, inject_bad=False)
lx = L.LivoxExtractor(ctx)
lx.extract(ls)
t = time.perf_counter()
for _ in range(20): rl = lx.extract(ls)
print(f"Livox: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms/scan incl. H2D+D2H ({len(rl['edge'])} edge, {len(rl['surf'])} surf)")
kf = np.concatenate([w["scan_xyz"], np.zeros((w["scan_xyz"].shape[0], 1), np.float32)], 1)
L.api.voxel_filter(ctx, kf, 0.4)
t = time.perf_counter()
for _ in range(10): L.api.voxel_filter(ctx, kf, 0.4)
print(f"VoxelGrid(0.4) of {kf.shape[0]} points: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms host in / host out")
