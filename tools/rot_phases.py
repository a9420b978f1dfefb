import sys, numpy as np
sys.path.insert(0, '.')
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=300_000, n_az=3125, half_extent=(150.0, 150.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10, np.float32)], 1)
ctx = L.Context(0)
ex = L.RotExtractor(ctx, ds_rate=1)
for _ in range(3): ex.extract(raw, debug=True)
