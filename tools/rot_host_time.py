"""ROT extractor from host buffers: page-locked (reuse=True) vs pageable, in batches of 50 calls, on the library's own stream and on a torch stream
(debug aid for bench extras.extract_rot, which is bimodal: 0.22 / 0.45 ms)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=300_000, n_az=3125, half_extent=(150.0, 150.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10.0, np.float32)], 1)
def rate(fn, reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for mode in ("own stream", "torch stream", "own stream"):
    if mode == "torch stream":
        ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
        ctx = L.Context(0, stream=ts.cuda_stream)
    else:
        ctx = L.Context(0)
    ex = L.RotExtractor(ctx, n_scans=64, ds_rate=4)
    ex.extract(raw)
    praw = L.api.PinnedArray(raw.shape, np.float32)
    praw.array[...] = raw
    ex.extract(praw.array, reuse=True)
    print(mode, "page-locked, batches of 50:", " ".join(f"{rate(lambda: ex.extract(praw.array, reuse=True), 50):.3f}" for _ in range(6)), " pageable:", f"{rate(lambda: ex.extract(raw), 20):.3f}",
          " page-locked again:", " ".join(f"{rate(lambda: ex.extract(praw.array, reuse=True), 50):.3f}" for _ in range(3)), flush=True)
    praw.close(); ctx.close()
