"""Per-phase clock ticks of ring 0 of the ROT extractor (LILI_ROT_PHASES) and the blocking device-resident extraction time.   python tools/rot_phases.py"""
import os, sys, time
os.environ["LILI_ROT_PHASES"] = "1"
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402
import ctypes as C

w = synth.make_workload(n_map=100_000, half_extent=(60.0, 60.0))
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10.0, np.float32)], 1)
ctx = L.Context(0)
d_raw = torch.from_numpy(raw).cuda()
for ds in (4, 2, 1):
    ex = L.RotExtractor(ctx, n_scans=64, ds_rate=ds)
    for _ in range(3):
        out = ex.extract_device(d_raw.data_ptr(), raw.shape[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ex.extract_device(d_raw.data_ptr(), raw.shape[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"ds_rate {ds}: {out} {dt * 1e6:.1f} us per blocking call", flush=True)
    counts = np.zeros(8, np.int32)
    ctx._chk(ctx.lib.lili_extract_rot_debug(ctx.h, counts.ctypes.data_as(C.c_void_p), None, None, None, None, None, None, None, None, None, None))
ctx.close()
