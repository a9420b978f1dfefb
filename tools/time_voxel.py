import sys, time, numpy as np
sys.path.insert(0, '.')
import lili_om_amd as L
from lili_om_amd import synth
rng = np.random.default_rng(0)
sc = synth.OutdoorScene()
raw = sc.sample_surfaces(460.0, 380.0, 0.4, rng).astype(np.float32)
raw = np.concatenate([raw, np.zeros((raw.shape[0], 1), np.float32)], 1)
ctx = L.Context(0)
import ctypes as C
lm = L.LocalMap(ctx, L.KIND_SURF, width=1, leaf=0.4)
lm.push(raw, [0, 0, 0], [1, 0, 0, 0])
lm.commit(); ctx.sync()
t = time.perf_counter()
for _ in range(5):
    n_raw, n_map = lm.commit()
ctx.sync()
dt = (time.perf_counter() - t) / 5
print(f"local-map commit: {n_raw} raw -> {n_map} map points: {dt*1e3:.2f} ms (concat + VoxelGrid + grid index, device-resident input)")
