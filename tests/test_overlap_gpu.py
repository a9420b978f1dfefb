"""Option "overlap_gn" (round 5): in lili_s2m_iterate* the association that follows a reduction + Gauss-Newton kernel is launched without a barrier against it and
takes the pose from the keyed granules that kernel publishes.  It moves no arithmetic: poses, records and counts must be bit-identical with the option on and off,
for every launch structure the loop chooses (one lane per query, cooperative lanes, both kinds in one launch, the dense-map index), across restarts."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


def _run(ctx, room, flavour, n_q, mask, overlap, n_iters=14, restart_every=5):
    ctx.set_option("overlap_gn", overlap)
    ctx.set_option("persistent_iterate", 0)
    P = L.make_params(flavour)
    m = L.ScanToMapMatcher(ctx, P)
    m.set_input_cloud(L.KIND_SURF, room["map_xyz"])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_SURF, np.ascontiguousarray(room["q_xyz"][:n_q]))
    m.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    t, q = room["t_true"], room["q_true"]
    if flavour == "rot":
        t, q = L.api.body_pose_from_lidar(t, q, P)
    t0, q0 = synth.perturbed_pose(t, q, np.random.default_rng(5), 0.2, 1.5)
    m.pose_set(1, t0, q0)
    m.iterate_restart(0, n_iters, restart_every, 1, mask)
    tp, qp, st = m.pose_get(0)
    n = m.find_corresponding_surf_features(0, *(L.api.assoc_transform(tp, qp, P) if flavour == "rot" else (qp, tp)))
    rec = m.surf_records(0, n)
    return np.r_[tp, qp], int(st), n, rec


@pytest.mark.parametrize("flavour,n_q,mask", [("rot", 150000, L.MASK_SURF), ("rot", 150000, L.MASK_SURF | L.MASK_EDGE), ("rot", 3000, L.MASK_SURF | L.MASK_EDGE),
                                              ("rot", 30000, L.MASK_SURF), ("frontend", 150000, L.MASK_SURF)])
def test_overlapped_association_changes_no_bit(flavour, n_q, mask):
    room = synth.make_room(seed=12, n_query=150000, n_edge_query=800)
    ctx = L.Context(0)
    try:
        a = _run(ctx, room, flavour, n_q, mask, 0)
        b = _run(ctx, room, flavour, n_q, mask, 1)
        c = _run(ctx, room, flavour, n_q, mask, 1)
        assert a[1] == b[1] == c[1] == 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(b[0], c[0]), (a[0], b[0])
        assert a[2] == b[2] and all(np.array_equal(a[3][k], b[3][k]) for k in a[3])
        t_ref = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], L.make_params(flavour))[0] if flavour == "rot" else room["t_true"]
        assert np.abs(a[0][:3] - t_ref).max() < 0.3          # (the last registration of the schedule has had 4 iterations: a sanity bound, not an accuracy claim)
    finally:
        ctx.set_option("overlap_gn", 0)
        ctx.close()


def test_overlapped_association_on_a_dense_map_index():
    """k_associate_fine (the density-adaptive index) takes the published pose the same way."""
    rng = np.random.default_rng(3)
    n = 400000
    xy = rng.uniform(-6, 6, (n, 2))
    mp = np.c_[xy, 0.002 * rng.standard_normal(n)].astype(np.float32)
    wall = np.c_[rng.uniform(-6, 6, n // 2), np.full(n // 2, 6.0) + 0.002 * rng.standard_normal(n // 2), rng.uniform(0, 3, n // 2)].astype(np.float32)
    mp = np.concatenate([mp, wall])
    qw = mp[rng.choice(mp.shape[0], 20000)].astype(np.float64) + rng.normal(0, 0.01, (20000, 3))
    t_true, q_true = np.array([0.3, -0.2, 1.0]), np.array([np.cos(0.1), 0, 0, np.sin(0.1)])
    ql = synth.quat_rot(q_true * np.array([1, -1, -1, -1]), qw - t_true).astype(np.float32)
    P = L.make_params("frontend")
    t0, q0 = synth.perturbed_pose(t_true, q_true, np.random.default_rng(9), 0.05, 0.5)
    out = []
    ctx = L.Context(0)
    try:
        for ov in (0, 1):
            ctx.set_option("overlap_gn", ov)
            ctx.set_option("fuse_lin", 0)          # the three-launch path (the fused association + linearisation launch has no separate GN hand-off)
            m = L.ScanToMapMatcher(ctx, P)
            m.set_input_cloud(L.KIND_SURF, mp)
            assert m.map_density(L.KIND_SURF)[1] > 0      # the fine index exists
            m.set_queries(0, L.KIND_SURF, ql)
            m.pose_set(0, t0, q0)
            m.iterate(0, 8, L.MASK_SURF)
            tp, qp, st = m.pose_get(0)
            assert st == 0
            out.append(np.r_[tp, qp])
        assert np.array_equal(out[0], out[1])
        assert np.abs(out[0][:3] - t_true).max() < 0.05
    finally:
        ctx.close()
