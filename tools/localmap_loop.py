"""Time of one keyframe of the ring local map (push + lili_localmap_commit): 50-keyframe ring of 20 k-point keyframes -> VoxelGrid(0.4) -> 152 k-point map + index.
    python tools/localmap_loop.py <incremental 0|1> [sort_fused_scan 0|1]"""
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
if len(sys.argv) > 2:
    ctx.set_option("sort_fused_scan", int(sys.argv[2]))
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
print("LOCALMAP incremental=%d%s: %.4f ms per keyframe (push %.4f + commit %.4f host-side) (%d -> %d points) stats %s" % (inc, (" fused_scan=" + sys.argv[2]) if len(sys.argv) > 2 else "", (time.perf_counter() - tic) / 20 * 1e3, tp[0] / 20 * 1e3, tc[0] / 20 * 1e3, n_raw, n_map, lm.stats()))
ctx.close()
