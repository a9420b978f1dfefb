import sys, time, numpy as np
sys.path.insert(0, '.')
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
    print(f"ds_rate {ds}: {dt*1e3:.3f} ms/scan incl. H2D+D2H ({len(r['edge'])} edge, {len(r['surf'])} surf, {len(r['full'])} full)")
