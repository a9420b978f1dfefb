"""Per-keyframe local map (f-1) timing probe: 50-keyframe ring of 20 k-point keyframes, push + commit (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=300_000, n_az=3125, half_extent=(150.0, 150.0))
feats = np.ascontiguousarray(np.concatenate([w["scan_xyz"][::10], np.zeros((w["scan_xyz"][::10].shape[0], 1), np.float32)], 1))
ctx = L.Context(0)
lm = L.api.LocalMap(ctx, L.KIND_SURF, width=50, leaf=0.4, max_sq_radius=1.0)
for k in range(50):
    lm.push(feats, [0.8 * k, 0.1 * k, 0.0], [1.0, 0.0, 0.0, 0.0])
print(lm.commit())
ctx.sync()
tic = time.perf_counter()
for k in range(20):
    lm.push(feats, [0.8 * (50 + k), 0.1 * (50 + k), 0.0], [1.0, 0.0, 0.0, 0.0]); lm.commit()
ctx.sync()
print("ms per keyframe", (time.perf_counter() - tic) / 20 * 1e3)
