"""A/B of the radix passes without the scan launch (option sort_fused_max_tiles) on the device-resident voxel filter.   python tools/voxel_ab.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import lili_om_amd as L
ctx = L.Context(0)
rng = np.random.default_rng(1)
for n in (200_000, 1_000_000, 2_000_000):
    pts = np.concatenate([rng.uniform(-60, 60, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    cloud = L.api.cloud_from_device(d.data_ptr(), n, 16, 12)
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    row = []
    for lim in (64, 256, 512):
        ctx.set_option("sort_fused_max_tiles", lim)
        for _ in range(3): L.api.voxel_filter_device(ctx, cloud, 0.4, out.data_ptr(), n)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): L.api.voxel_filter_device(ctx, cloud, 0.4, out.data_ptr(), n)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t) / 20 * 1e6)
    print(n, "limit 64: %.1f us   limit 256: %.1f us   limit 512: %.1f us" % tuple(row), flush=True)
ctx.close()
