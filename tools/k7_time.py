"""lili_map_set on the bench's 5 M-point map: measured box first vs the previous build's box (option map_guess_box), base index alone and with the
focused super-row copy; the map in the pipeline's order (ascending voxel id, what the synthetic map already is) and randomly permuted.
    python tools/k7_time.py            (one JSON line per case)"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

w = synth.make_workload(n_map=5_000_000, half_extent=(460.0, 380.0))
ctx = L.Context(0)
m = L.ScanToMapMatcher(ctx, L.make_params("rot"))
mp = w["map_xyz"]
focus_r = float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0
perm = np.random.default_rng(3).permutation(mp.shape[0])
import os
cases = (("voxel order", mp), ("random order", np.ascontiguousarray(mp[perm])))
if os.environ.get("K7_ONLY") == "voxel":
    cases = cases[:1]
for name, pts in cases:
    d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    cloud = L.api.cloud_from_device(d.data_ptr(), pts.shape[0], 12, -1)
    for guess in (0, 1):
        ctx.set_option("map_guess_box", guess)
        for sr, foc in ((0, False), (1, True)):
            ctx.set_option("super_rows", sr)
            m.map_focus(w["lidar_t"], focus_r) if foc else m.map_focus(None)
            for _ in range(3):
                m.set_input_cloud(L.KIND_SURF, cloud)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20):
                m.set_input_cloud(L.KIND_SURF, cloud)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 20 * 1e3
            print(json.dumps({"order": name, "map_guess_box": guess, "layout": "focused super rows" if sr else "base only", "ms_per_build": round(ms, 4),
                              "n_cells": m.map_info(L.KIND_SURF)[1], "build_stats": m.map_build_stats()}), flush=True)
ctx.set_option("super_rows", 1)
ctx.set_option("map_guess_box", 1)
ctx.close()
