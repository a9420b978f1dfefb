#!/usr/bin/env python
"""Profiling aid: per-workgroup begin / end timestamps of one k_associate_surf launch of the bench workload
(LILI_DEBUG bit 4096 writes them into the neighbour debug rows).  Prints the distribution of wave lifetimes, the start
skew and how well the SIMDs are packed.  usage: python tools/assoc_blocks.py [after_iters] [extra LILI_DEBUG bits]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (single HIP runtime)
import lili_om_amd as L  # noqa: E402
from lili_om_amd import synth  # noqa: E402
import bench  # noqa: E402

after = int(sys.argv[1]) if len(sys.argv) > 1 else 10
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_az = int(os.environ.get("LILI_N_AZ", bench.N_AZ))   # 3125 azimuth steps x 64 rings = the 200 k-point scan; other values scale the query count
w = synth.make_workload(n_map=bench.N_MAP, n_az=n_az, half_extent=(460.0, 380.0))
P = L.make_params("rot")
t_body, q_body = bench.body_pose_for_lidar(L, P, w["lidar_t"])
t0, q0 = synth.perturbed_pose(t_body, q_body, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
scan = bench.ring_major(w["scan_xyz"], w["scan_ring"])
ctx = L.Context(0)
ctx.set_debug(True)
m = L.ScanToMapMatcher(ctx, P)
m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
m.set_queries(0, L.KIND_SURF, scan)
m.pose_set(0, t0, q0)
m.iterate(0, after, L.MASK_SURF)
ctx.sync()
os.environ["LILI_DEBUG"] = str(4096 | extra)
for rep in range(5):
    m.associate_dev(0, L.MASK_SURF)
    ctx.sync()
idx, d2dbg = m.neighbors(0, L.KIND_SURF, scan.shape[0])
nb = (scan.shape[0] + 63) // 64
rows = idx[::64][:nb].copy()                      # (nb, 5) int32: t_begin lo/hi, t_end lo/hi, hw
tb = rows[:, 0:2].copy().view(np.int64)[:, 0]
te = rows[:, 2:4].copy().view(np.int64)[:, 0]
hw = rows[:, 4].astype(np.uint32)
t00 = tb.min()
dur = (te - tb) * 0.01                            # us (100 MHz ticks)
start = (tb - t00) * 0.01
end = (te - t00) * 0.01
print(f"blocks {nb}  kernel span {end.max():.2f} us  (first begin -> last end)")
print("wave lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (dur.mean(), *np.percentile(dur, [10, 50, 90, 99]), dur.max()))
print("start offset us : mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (start.mean(), *np.percentile(start, [50, 90, 99]), start.max()))
print("end time us     : p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f" % tuple(np.percentile(end, [10, 50, 90, 99])))
xcc = hw >> 16
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (cu.astype(np.int64) << 4) | simd
uk, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,cu,simd) = {len(uk)}  waves per SIMD: min {cnt.min()} max {cnt.max()}  mean {cnt.mean():.2f}")
print("waves per XCC:", np.bincount(xcc.astype(np.int64)))
tot = {k: 0.0 for k in uk}
for k, d in zip(key, dur):
    tot[k] += d
tt = np.array(list(tot.values()))
print("sum of wave lifetimes per SIMD us: mean %.1f max %.1f" % (tt.mean(), tt.max()))
order = np.argsort(-dur)[:10]
print("slowest blocks:", [(int(b), round(float(dur[b]), 1), round(float(start[b]), 1)) for b in order])
# lifetime vs block index (coarse): where in the scan are the slow waves
q = np.array_split(dur, 16)
print("mean lifetime by block-index sixteenth:", [round(float(x.mean()), 1) for x in q])
tcq = idx[:, 0].copy().astype(np.int64)
tcq[::64] = 0
tcb = np.array([tcq[b * 64:(b + 1) * 64].max() for b in range(nb)])
tcm = np.array([tcq[b * 64:(b + 1) * 64].sum() / 63.0 for b in range(nb)])
print("chunks per query: mean %.2f p50 %d p90 %d p99 %d max %d" % (tcq[tcq > 0].mean(), *np.percentile(tcq[tcq > 0], [50, 90, 99]), tcq.max()))
print("per wave: max-lane chunks mean %.2f p50 %d p90 %d p99 %d max %d ; mean-lane chunks mean %.2f" % (tcb.mean(), *np.percentile(tcb, [50, 90, 99]), tcb.max(), tcm.mean()))
print("corr(wave lifetime, max-lane chunks) = %.3f" % np.corrcoef(dur, tcb)[0, 1])
print("slowest blocks max-lane chunks:", [(int(b), int(tcb[b]), round(float(tcm[b]), 1)) for b in order])
if os.environ.get("LILI_PHASE_PROBE"):
    # library built with -DLILI_PHASE_PROBE (tools/assoc_phases.sh): ticks since the block began at the phase boundaries, in the d2 debug
    # floats 1..6 of the block's first query (spilling into the second query's row)
    flat = d2dbg.reshape(-1)
    ph = np.stack([flat[b * 64 * 5 + 1: b * 64 * 5 + 7] for b in range(nb)]).astype(np.float64) * 0.01      # us since block begin
    names = ["query moved into map frame", "row ranges loaded, table built", "inner 3x3x3 block walked", "shell decided / walked",
             "five winners resolved", "plane fitted"]
    okb = (ph > 0).all(1)
    print(f"phase probe: {okb.sum()} of {nb} blocks with all stamps")
    prev = np.zeros(nb)
    for k, nm in enumerate(names):
        seg = ph[:, k] - prev
        print("  %-34s  at %6.2f us (p50)   segment mean %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" %
              (nm, np.median(ph[okb, k]), seg[okb].mean(), *np.percentile(seg[okb], [10, 50, 90]), seg[okb].max()))
        prev = ph[:, k]
    seg = dur - prev
    print("  %-34s  at %6.2f us (p50)   segment mean %5.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" %
          ("records stored, block count", np.median(dur[okb]), seg[okb].mean(), *np.percentile(seg[okb], [10, 50, 90]), seg[okb].max()))
    slow = np.argsort(-dur)[:5]
    for b in slow:
        print("  slow block %5d: lifetime %.1f us, stamps" % (b, dur[b]), np.round(ph[b], 2))
np.save("gpurun_out/assoc_blocks.npy", np.stack([start, end, key.astype(np.float64)], 1)) if os.path.isdir("gpurun_out") else None
ctx.close()
