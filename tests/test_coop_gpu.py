"""Cooperative association (lili_s2m_coop.hip: 2 / 4 / 8 / 16 lanes of a wave per query) against the one-lane kernels.

The lanes of a group only SHARE the candidate walk of a query: the candidate set, the (distance, original index) order of the five
winners, the fit and the gates are the same functions — so neighbours, distances, records, counts and the Gram of a separate
linearisation must be bit-identical, for every flavour, from a pose far enough off that a good part of the queries walks the shell,
with the super-row layout on, off and restricted to a focus box (queries outside it take the nine-row walk inside the same launch).
"""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu

LANES = (2, 4, 8, 16)


def _clouds(room, flavour, seed=7):
    rng = np.random.default_rng(seed)
    refl = lambda n: rng.uniform(0.0, 0.05, (n, 1)).astype(np.float32)
    smap, emap, sq, eq = room["map_xyz"], room["edge_map_xyz"], room["q_xyz"], room["eq_xyz"]
    if flavour == "livox":
        smap, emap, sq, eq = [np.concatenate([a.astype(np.float32), refl(a.shape[0])], 1) for a in (smap, emap, sq, eq)]
    return smap, emap, sq, eq


def _run(ctx, P, clouds, t0, q0, lanes, mode, centre, n_iter=0):
    smap, emap, sq, eq = clouds
    ctx.set_option("assoc_lpq", lanes)
    ctx.set_option("super_rows", 0 if mode == "off" else 1)
    m = L.ScanToMapMatcher(ctx, P)
    m.map_focus(centre + [1.0, -0.5, 0.0], 2.5) if mode == "focus" else m.map_focus(None)
    ctx.set_debug(True)
    m.set_input_cloud(L.KIND_SURF, smap)
    m.set_input_cloud(L.KIND_EDGE, emap)
    m.set_queries(0, L.KIND_SURF, sq)
    m.set_queries(0, L.KIND_EDGE, eq)
    m.pose_set(0, t0, q0)
    m.associate_dev(0, L.MASK_SURF | L.MASK_EDGE)
    out = {}
    out["sidx"], out["sd2"] = m.neighbors(0, L.KIND_SURF, sq.shape[0])
    out["eidx"], out["ed2"] = m.neighbors(0, L.KIND_EDGE, eq.shape[0])
    out["srec"] = m.surf_records(0, sq.shape[0])
    out["erec"] = m.edge_records(0, eq.shape[0])
    out["lin"] = m.linearize(0, t0, q0, L.MASK_SURF | L.MASK_EDGE)
    if n_iter:
        m.iterate(0, n_iter, L.MASK_SURF | L.MASK_EDGE)
        out["pose"] = m.pose_get(0)
    return out


def _same_records(a, b):
    assert sorted(a.keys()) == sorted(b.keys()) and a["count"] == b["count"]
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize("flavour", ["rot", "livox", "frontend"])
@pytest.mark.parametrize("mode", ["all", "focus", "off"])
def test_cooperative_lanes_change_no_result(gpu_ctx, flavour, mode):
    room = synth.make_room(seed=43, n_query=6000, n_edge_query=400)
    P = L.make_params(flavour)
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(23), 0.3, 2.0)
    clouds = _clouds(room, flavour)
    centre = np.asarray(room["t_true"], np.float64)
    try:
        ref = _run(gpu_ctx, P, clouds, t0, q0, 1, mode, centre)
        gate = 1.0
        assert ref["lin"][2][0] > 500
        assert ((ref["sd2"][:, 4] >= 0.4) & (ref["sd2"][:, 4] < gate)).sum() > 100      # the shell matters for a good part of the queries
        for lanes in LANES:
            got = _run(gpu_ctx, P, clouds, t0, q0, lanes, mode, centre)
            for k in ("s", "e"):
                inside = ref[k + "d2"][:, 4] < gate          # queries the reference keeps: the neighbours are the global 5-NN on both sides
                assert np.array_equal(ref[k + "idx"][inside], got[k + "idx"][inside]), (lanes, k)
                assert np.array_equal(ref[k + "d2"][inside], got[k + "d2"][inside]), (lanes, k)
                outside = ~inside                             # the others end at the gate on both sides (which points they saw is immaterial)
                assert (got[k + "d2"][outside][:, 4] >= gate).all()
            _same_records(ref["srec"], got["srec"])
            _same_records(ref["erec"], got["erec"])
            G0, c0, n0 = ref["lin"]
            G1, c1, n1 = got["lin"]
            assert np.array_equal(G0, G1) and c0 == c1 and np.array_equal(n0, n1), lanes
    finally:
        gpu_ctx.set_option("assoc_lpq", 0)
        gpu_ctx.set_option("super_rows", 1)
        gpu_ctx.set_debug(False)
        L.ScanToMapMatcher(gpu_ctx, P).map_focus(None)


@pytest.mark.parametrize("flavour", ["rot", "rot_three_launches", "frontend"])
def test_cooperative_iterations_follow_the_one_lane_iterations(gpu_ctx, flavour):
    """Whole registrations.  Where the linearisation is its own launch on identical records (ROT with option count_barrier = 0) the poses are
    bit-identical; where it happens inside the association launch (the flavours without count scaling — and the count-scaled ROT flavour
    through the in-launch count barrier of small launches) only the PARTITION of the Gram sum changes with the lanes per query (one
    partial per 256 / L queries): poses agree to ~1e-11."""
    three = flavour == "rot_three_launches"
    flavour = "rot" if three else flavour
    gpu_ctx.set_option("count_barrier", 0 if three else 1)      # (off by default: measured no gain; the path stays tested)
    gpu_ctx.set_option("persistent_iterate", 0)                 # launch by launch here; the persistent launch has its own test below
    room = synth.make_room(seed=47, n_query=3000, n_edge_query=250)
    P = L.make_params(flavour)
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(29), 0.2, 1.5)
    clouds = _clouds(room, flavour)
    centre = np.asarray(room["t_true"], np.float64)
    try:
        ref = _run(gpu_ctx, P, clouds, t0, q0, 1, "all", centre, n_iter=8)
        for lanes in LANES + (0,):
            got = _run(gpu_ctx, P, clouds, t0, q0, lanes, "all", centre, n_iter=8)
            (t_a, q_a, st_a), (t_b, q_b, st_b) = ref["pose"], got["pose"]
            assert st_a == 0 and st_b == 0
            if three:
                assert np.array_equal(t_a, t_b) and np.array_equal(q_a, q_b), lanes
            else:
                assert np.abs(t_a - t_b).max() < 1e-10 and np.abs(q_a - q_b).max() < 1e-10, lanes
    finally:
        gpu_ctx.set_option("assoc_lpq", 0)
        gpu_ctx.set_option("count_barrier", 0)
        gpu_ctx.set_debug(False)


def test_cooperative_ragged_and_tiny_inputs(gpu_ctx):
    """Query counts that do not fill a group, a block or a wave, and a map with fewer than five points in reach of most queries."""
    room = synth.make_room(seed=53, n_query=1100, n_edge_query=70)
    P = L.make_params("rot")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    centre = np.asarray(room["t_true"], np.float64)
    try:
        for n_s, n_e in ((1, 1), (5, 3), (63, 17), (257, 65), (1100, 70)):
            clouds = (room["map_xyz"], room["edge_map_xyz"], room["q_xyz"][:n_s], room["eq_xyz"][:n_e])
            ref = _run(gpu_ctx, P, clouds, tb, qb, 1, "all", centre)
            for lanes in LANES:
                got = _run(gpu_ctx, P, clouds, tb, qb, lanes, "all", centre)
                _same_records(ref["srec"], got["srec"])
                _same_records(ref["erec"], got["erec"])
                assert np.array_equal(ref["lin"][0], got["lin"][0]) and np.array_equal(ref["lin"][2], got["lin"][2])
        sparse = (room["map_xyz"][::400], room["edge_map_xyz"][::40], room["q_xyz"][:300], room["eq_xyz"][:40])
        ref = _run(gpu_ctx, P, sparse, tb, qb, 1, "all", centre)
        for lanes in LANES:
            got = _run(gpu_ctx, P, sparse, tb, qb, lanes, "all", centre)
            _same_records(ref["srec"], got["srec"])
            assert np.array_equal(ref["lin"][2], got["lin"][2])
    finally:
        gpu_ctx.set_option("assoc_lpq", 0)
        gpu_ctx.set_debug(False)


@pytest.mark.parametrize("flavour", ["rot", "livox", "frontend"])
@pytest.mark.parametrize("n_surf,n_edge", [(2500, 200), (700, 0), (12000, 600)])
def test_persistent_iterations_follow_the_launch_by_launch_loop(gpu_ctx, flavour, n_surf, n_edge):
    """k_iterate_coop (option persistent_iterate; off by default — measured no faster than the launches): the outer iterations of a SMALL scan — associate, [exchange the counts],
    linearise, sum, Gauss-Newton step — inside one launch whose workgroups all stay resident; the pose travels through LDS and the
    granule exchange instead of through launches.  Same records, same rows; only the partition of the Gram sum differs from the
    launch-per-stage loop (one partial per 256 / L queries), so poses agree to 1e-10, iteration counters and statuses are equal, the
    records left behind are those of the LAST association, and a repeat gives the same bits.  Also with restarts (iterate_sharded's
    restart_every, here through lili_s2m_iterate with a restart slot) and for three slots side by side (iterate_window)."""
    room = synth.make_room(seed=59, n_query=n_surf, n_edge_query=max(n_edge, 1))
    P = L.make_params(flavour)
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(31), 0.15, 1.2)
    smap, emap, sq, eq = _clouds(room, flavour)
    mask = L.MASK_SURF | (L.MASK_EDGE if n_edge else 0)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, smap)
    m.set_input_cloud(L.KIND_EDGE, emap)
    for k in range(3):
        m.set_queries(k, L.KIND_SURF, sq)
        if n_edge:
            m.set_queries(k, L.KIND_EDGE, eq[:n_edge])
    gpu_ctx.set_debug(True)
    res = {}
    try:
        for mode in (0, 1, 1):
            gpu_ctx.set_option("persistent_iterate", mode)
            m.pose_set(0, t0, q0)
            m.iterate(0, 7, mask)
            t, q, st = m.pose_get(0)
            rec = m.surf_records(0, sq.shape[0])
            lin = m.linearize(0, t, q, mask)
            if mode in res:
                t1, q1, st1, rec1, lin1 = res[mode]                 # deterministic: the second persistent run repeats every bit
                assert np.array_equal(t, t1) and np.array_equal(q, q1) and np.array_equal(lin[0], lin1[0])
            res[mode] = (t, q, st, rec, lin)
        (ta, qa, sa, ra, la), (tb_, qb_, sb, rb, lb) = res[0], res[1]
        assert sa == 0 and sb == 0 and np.abs(ta - t0).max() > 1e-3
        assert np.abs(ta - tb_).max() < 1e-10 and np.abs(qa - qb_).max() < 1e-10
        assert ra["count"] == rb["count"] > 0.3 * n_surf and np.array_equal(ra["query_index"], rb["query_index"])
        assert np.abs(ra["n"] - rb["n"]).max() < 1e-6 and np.array_equal(la[2], lb[2])
        assert np.abs(la[0] - lb[0]).max() <= 1e-7 * np.abs(la[0]).max()
        # three slots side by side: the same bits as one after the other (the lanes-per-query rule does not depend on what else runs)
        gpu_ctx.set_option("persistent_iterate", 1)
        starts = [synth.perturbed_pose(tb, qb, np.random.default_rng(40 + k), 0.1, 0.8) for k in range(3)]
        one = []
        for k in range(3):
            m.pose_set(k, *starts[k])
            m.iterate(k, 5, mask)
            one.append(m.pose_get(k))
        for k in range(3):
            m.pose_set(k, *starts[k])
        m.iterate_window([0, 1, 2], 5, mask)
        for k in range(3):
            t, q, st = m.pose_get(k)
            assert st == 0 and np.array_equal(t, one[k][0]) and np.array_equal(q, one[k][1])
    finally:
        gpu_ctx.set_option("persistent_iterate", 0)
        gpu_ctx.set_debug(False)
