"""Stability check (through gpurun): many lili_map_set calls with changing sizes / focus / options, free-memory watermark and result stability."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lili_om_amd as L
from lili_om_amd import synth

room = synth.make_room(seed=5, n_query=6000, n_edge_query=300)
P = L.make_params("rot")
ctx = L.Context(0)
m = L.ScanToMapMatcher(ctx, P)
tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(3), 0.1, 0.5)
rng = np.random.default_rng(0)
ref = None
free0 = None
for it in range(300):
    n = int(rng.integers(room["map_xyz"].shape[0] // 2, room["map_xyz"].shape[0]))
    sub = room["map_xyz"] if it % 3 == 0 else room["map_xyz"][np.sort(rng.choice(room["map_xyz"].shape[0], n, replace=False))]
    ctx.set_option("super_rows", int(it % 5 != 4))
    if it % 2: m.map_focus(room["t_true"] + rng.normal(0, 1, 3), float(rng.uniform(0.5, 6)))
    else: m.map_focus(None)
    m.set_input_cloud(L.KIND_SURF, sub)
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_SURF, room["q_xyz"]); m.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    m.pose_set(0, t0, q0)
    m.iterate(0, 4, L.MASK_SURF | L.MASK_EDGE)
    t, q, st = m.pose_get(0)
    assert st == 0
    if it % 3 == 0:
        if ref is None: ref = (t.copy(), q.copy())
        assert np.array_equal(t, ref[0]) and np.array_equal(q, ref[1]), (it, t, ref[0])
    if it == 20: free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("ok: 300 map rebuilds, full-map results bit-stable, free memory change since rebuild 20:", (free1 - free0) / 1e6, "MB;",
      "index builds from a guessed box / repeated with the measured box / repeated with the three-kernel scan:", m.map_build_stats())
