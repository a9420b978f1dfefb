"""Stress of the ROT extractor's inter-workgroup waits (k_rot_ring's look-back over lower rings, k_rot_segments' wait for the predecessor's spill): K contexts extract the same
scan concurrently; every result must equal the single-context one and no call may take as long as a spin that gives up (~0.3-1 s).   python tools/rot_concurrent_stress.py [K] [scans]"""
import sys, time, threading
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w = synth.make_workload(n_map=100_000, half_extent=(60.0, 60.0), verbose=False)
raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10.0, np.float32)], 1)
d_raw = torch.from_numpy(raw).cuda()
ref = None
worst = [0.0] * K
bad = [0] * K


def work(k):
    ctx = L.Context(0)
    ex = L.RotExtractor(ctx, n_scans=64, ds_rate=1)
    for _ in range(N):
        t0 = time.perf_counter()
        out = ex.extract_device(d_raw.data_ptr(), raw.shape[0])
        worst[k] = max(worst[k], time.perf_counter() - t0)
        if out != ref:
            bad[k] += 1
    ctx.close()


c0 = L.Context(0)
ref = L.RotExtractor(c0, n_scans=64, ds_rate=1).extract_device(d_raw.data_ptr(), raw.shape[0])
c0.close()
th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print({"contexts": K, "scans_each": N, "counts": ref, "mismatches": sum(bad), "worst_call_ms": round(max(worst) * 1e3, 3), "scans_per_s": round(K * N / dt, 1)})
assert sum(bad) == 0 and max(worst) < 0.2
