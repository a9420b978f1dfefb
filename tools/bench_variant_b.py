"""SURVEY §8d Config 2, variant B at full size (through gpurun): an 80 x 60 x 12 m room sampled at ~0.05 m (5 M map points, ~400 points per m^2)
and 200 k queries near its surfaces; association time with the density-adaptive fine index on / off, records compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth

import bench_configs as BC
mp, q_local, t_true, q_true = BC.make_variant_b()
print(f"variant B map: {mp.shape[0]} points", file=sys.stderr)
P = L.make_params("rot")
tb, qb = L.api.body_pose_from_lidar(t_true, q_true, P)
Q2, T2 = L.api.assoc_transform(tb, qb, P)
ctx = L.Context(0)
res = {}
for fine in (1, 0):
    ctx.set_option("fine_grid", fine)
    m = L.ScanToMapMatcher(ctx, P)
    t0 = time.perf_counter(); m.set_input_cloud(L.KIND_SURF, mp); ctx.sync(); t_build = time.perf_counter() - t0
    m.set_queries(0, L.KIND_SURF, q_local)
    n = m.find_corresponding_surf_features(0, Q2, T2)
    for _ in range(3): m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.sync(); torch.cuda.synchronize()
    tic = time.perf_counter()
    reps = 20 if fine else 5
    for _ in range(reps): m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
    ctx.sync()
    us = (time.perf_counter() - tic) / reps * 1e6
    G, cost, counts = m.linearize(0, tb, qb, L.MASK_SURF)
    res[fine] = (n, G, cost)
    print(f"fine_grid={fine}: {n} correspondences, association {us:.0f} us per launch (host clock over {reps} launches), map_set {t_build*1e3:.1f} ms incl. upload, density {m.map_density(L.KIND_SURF)}")
assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
print("records identical (Gram and cost bit for bit)")
