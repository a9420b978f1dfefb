"""SURVEY §8d Config 2, variant B at full size (through gpurun): an 80 x 60 x 12 m room sampled at ~0.05 m (5 M map points, ~400 points per m^2)
and 200 k queries near its surfaces; association time with the density-adaptive fine index on / off, records compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth

rng = np.random.default_rng(0x11110)
leaf = 0.05
def plane(u0, u1, v0, v1, fn):
    nu, nv = int((u1 - u0) / leaf), int((v1 - v0) / leaf)
    U, V = np.meshgrid(u0 + (np.arange(nu) + 0.5) * leaf, v0 + (np.arange(nv) + 0.5) * leaf, indexing="ij")
    U = U.ravel() + rng.uniform(-0.3, 0.3, U.size) * leaf; V = V.ravel() + rng.uniform(-0.3, 0.3, V.size) * leaf
    return fn(U, V) + rng.normal(0, 0.004, (U.size, 3))
X, Y, Z = 80.0, 60.0, 12.0
parts = [plane(-X/2, X/2, -Y/2, Y/2, lambda u, v: np.stack([u, v, np.zeros_like(u)], 1)),
         plane(-X/2, X/2, -Y/2, Y/2, lambda u, v: np.stack([u, v, np.full_like(u, Z)], 1)),
         plane(-X/2, X/2, 0, Z, lambda u, v: np.stack([u, np.full_like(u, -Y/2), v], 1)),
         plane(-X/2, X/2, 0, Z, lambda u, v: np.stack([u, np.full_like(u, Y/2), v], 1)),
         plane(-Y/2, Y/2, 0, Z, lambda u, v: np.stack([np.full_like(u, -X/2), u, v], 1)),
         plane(-Y/2, Y/2, 0, Z, lambda u, v: np.stack([np.full_like(u, X/2), u, v], 1))]
mp = np.concatenate(parts).astype(np.float32)
mp = mp[rng.permutation(mp.shape[0])[:5_000_000]] if mp.shape[0] > 5_000_000 else mp
print(f"variant B map: {mp.shape[0]} points", file=sys.stderr)
nq = 200_000
qw = mp[rng.choice(mp.shape[0], nq)].astype(np.float64) + rng.normal(0, 0.02, (nq, 3))
t_true = np.array([1.0, -2.0, 1.8]); ang = np.radians(20.0)
q_true = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)])
q_local = synth.quat_rot(q_true * np.array([1, -1, -1, -1]), qw - t_true).astype(np.float32)
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
