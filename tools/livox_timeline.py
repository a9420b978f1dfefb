"""One Livox extraction (24 k points, page-locked host buffers as in bench.py) in a loop — run under rocprofv3 --kernel-trace --memory-copy-trace to see the chain."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import lili_om_amd as L
from lili_om_amd import synth
ctx = L.Context(0)
ls = synth.make_livox_scan(3, inject_bad=False)
lx = L.LivoxExtractor(ctx)
lx.extract(ls)
pin = L.api.PinnedArray(ls.shape, np.float32)
pin.array[...] = ls
for _ in range(5): lx.extract(pin.array, reuse=True)
t = time.perf_counter()
for _ in range(40): r = lx.extract(pin.array, reuse=True)
print(f"Livox: {(time.perf_counter() - t) / 40 * 1e3:.4f} ms/scan ({len(r['edge'])} edge, {len(r['surf'])} surf)")
ctx.close()
