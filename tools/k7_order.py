"""lili_map_set on the bench's 5 M-point map in random order (the synthetic map) and in VoxelGrid order (what the pipeline hands over).
    python tools/k7_order.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

w = synth.make_workload(n_map=5_000_000, half_extent=(460.0, 380.0))
ctx = L.Context(0)
m = L.ScanToMapMatcher(ctx, L.make_params("rot"))
mp = w["map_xyz"]
vi = np.floor(mp / np.float32(0.4)).astype(np.int64)
order = np.lexsort((vi[:, 0], vi[:, 1], vi[:, 2]))
focus_r = float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0
for name, pts in (("random order", mp), ("voxel order", np.ascontiguousarray(mp[order]))):
    d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    cloud = L.api.cloud_from_device(d.data_ptr(), pts.shape[0], 12, -1)
    for sr, foc in ((0, False), (1, True)):
        ctx.set_option("super_rows", sr)
        m.map_focus(w["lidar_t"], focus_r) if foc else m.map_focus(None)
        for _ in range(2):
            m.set_input_cloud(L.KIND_SURF, cloud)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            m.set_input_cloud(L.KIND_SURF, cloud)
        torch.cuda.synchronize()
        print(f"{name}, {'focused super rows' if sr else 'base only'}: {(time.perf_counter() - t) / 10 * 1e3:.4f} ms per build", flush=True)
ctx.set_option("super_rows", 1)
ctx.close()
