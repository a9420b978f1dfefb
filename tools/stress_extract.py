"""Stability check of the extractors and the local map (through gpurun): many scans of changing size / options through lili_extract_rot, lili_extract_livox and
the ring local map; results of a repeated scan must stay bit-identical, the free device memory must not drift."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth

ctx = L.Context(0)
w = synth.make_workload(n_map=100_000, half_extent=(60.0, 60.0), verbose=False)
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10.0, np.float32)], 1)
ls = synth.make_livox_scan(3, inject_bad=True)
rng = np.random.default_rng(1)
ref = {}
free0 = None
lm = L.api.LocalMap(ctx, L.KIND_SURF, width=10, leaf=0.4, max_sq_radius=1.0)
for it in range(200):
    ds = (4, 2, 1)[it % 3]
    full = it % 4 == 0
    n = raw.shape[0] if full else int(rng.integers(2000, raw.shape[0]))
    lo = 0 if full else int(rng.integers(0, raw.shape[0] - n + 1))
    ex = L.RotExtractor(ctx, n_scans=64, ds_rate=ds, ds_v=(0.6, 0.05)[it % 7 == 6])
    g = ex.extract(raw[lo:lo + n], debug=True)
    if full and it % 7 != 6:
        key = ("rot", ds)
        sig = (g["edge_idx"].tobytes(), g["surf"].tobytes(), g["label"].tobytes())
        assert ref.setdefault(key, sig) == sig, ("ROT result changed", it)
    lx = L.LivoxExtractor(ctx)
    r = lx.extract(ls if it % 2 == 0 else ls[: int(rng.integers(100, ls.shape[0]))], debug=True)
    if it % 2 == 0:
        sig = (r["edge"].tobytes(), r["surf"].tobytes(), r["cutted"].tobytes())
        assert ref.setdefault("livox", sig) == sig, ("Livox result changed", it)
    lm.push(np.ascontiguousarray(g["surf"]), [0.5 * it, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0])
    lm.commit()
    if it == 20:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free memory after 20 / 200 rounds: %d / %d MiB; local map commits (incremental, full) %s" % (free0 >> 20, free1 >> 20, lm.stats()))
assert abs(free0 - free1) < (64 << 20), "device memory drifted"
print("STRESS OK")
ctx.close()
