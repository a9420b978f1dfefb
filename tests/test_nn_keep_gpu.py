"""The verified neighbour cache of the one-lane association kernels (lili_s2m_dev.h, round 4; option "nn_cache", default on).

Claim: an association served from the cache returns EXACTLY what a full search returns — the same five neighbours, the same f32 distances in the same
(distance, original index) order, hence the same records, counts, Grams and poses, bit for bit.  Here:
  * whole registrations (three flavours, surf + edge) with the cache on and off: every record, neighbour row and pose bit-identical at every iteration,
    while the records of the cache show that most queries were indeed served without a search;
  * adversarial geometry: a lattice map (exact distance ties everywhere), duplicated map points, moves from nanometres to centimetres in a random walk —
    neighbours against brute force after every move;
  * the records are dropped when the map or the queries change."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu
MASK = L.MASK_SURF | L.MASK_EDGE


@pytest.fixture()
def one_lane(gpu_ctx):
    """the cache lives in the one-lane kernels; small test scenes would otherwise go to the cooperative ones"""
    gpu_ctx.set_option("assoc_lpq", 1)
    gpu_ctx.set_option("fuse_lin", 0)
    yield gpu_ctx
    gpu_ctx.set_option("assoc_lpq", 0)
    gpu_ctx.set_option("fuse_lin", 1)
    gpu_ctx.set_option("nn_cache", 1)
    gpu_ctx.set_debug(False)


def _scene(flavour, seed, n_q=6000, n_e=400):
    room = synth.make_room(seed=seed, n_query=n_q, n_edge_query=n_e)
    P = L.make_params(flavour)
    if flavour == "frontend":
        tb, qb = np.asarray(room["t_true"], np.float64), np.asarray(room["q_true"], np.float64)
    else:
        tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(seed + 1), 0.08, 0.8)
    return room, P, t0, q0


def _matcher(ctx, room, P, flavour):
    m = L.ScanToMapMatcher(ctx, P)
    if flavour == "livox":
        m.set_input_cloud(L.KIND_SURF, np.c_[room["map_xyz"], room["map_refl"]])
        m.set_queries(0, L.KIND_SURF, np.c_[room["q_xyz"], room["q_refl"]])
    else:
        m.set_input_cloud(L.KIND_SURF, room["map_xyz"])
        m.set_queries(0, L.KIND_SURF, room["q_xyz"])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    return m


def _snapshot(m, room):
    ns, ne = room["q_xyz"].shape[0], room["eq_xyz"].shape[0]
    rs, re_ = m.surf_records(0, ns), m.edge_records(0, ne)
    i_s, d_s = m.neighbors(0, L.KIND_SURF, ns)
    i_e, d_e = m.neighbors(0, L.KIND_EDGE, ne)
    t, q, st = m.pose_get(0)
    return dict(rs=rs, re=re_, i_s=i_s, d_s=d_s, i_e=i_e, d_e=d_e, t=np.array(t), q=np.array(q), st=st)


def _same(a, b):
    for k in ("cp", "n", "d", "score", "query_index"):
        assert np.array_equal(a["rs"][k], b["rs"][k]), k
    for k in ("cp", "a", "b", "s", "query_index"):
        assert np.array_equal(a["re"][k], b["re"][k]), k
    for k in ("i_s", "i_e"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("d_s", "d_e"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(a["t"], b["t"]) and np.array_equal(a["q"], b["q"]) and a["st"] == b["st"] == 0


@pytest.mark.parametrize("flavour", ["rot", "livox", "frontend"])
def test_registration_with_and_without_the_cache_is_bit_identical(one_lane, flavour):
    ctx = one_lane
    room, P, t0, q0 = _scene(flavour, seed=71)
    runs = {}
    served = []
    for on in (1, 0):
        ctx.set_option("nn_cache", on)
        ctx.set_debug(True)
        m = _matcher(ctx, room, P, flavour)
        m.pose_set(0, t0, q0)
        snaps, prev = [], None
        for it in range(9):
            m.iterate(0, 1, MASK)
            snaps.append(_snapshot(m, room))
            if on:
                rec = m.nn_cache_records(0, L.KIND_SURF, room["q_xyz"].shape[0])
                if prev is not None:
                    served.append(float((np.array_equal(rec, prev) and 1.0) or (rec == prev).all(1).mean()))
                prev = rec
        runs[on] = snaps
    for a, b in zip(runs[1], runs[0]):
        _same(a, b)
    assert runs[1][-1]["rs"]["count"] > 3000
    # iteration 1 follows a pose reset (write-only), the first moves are centimetres: searches; from the fourth iteration on the moves are far below the
    # margins and (nearly) every wave keeps its neighbours
    assert max(served[:1]) < 0.5 and min(served[4:]) > 0.9, served


def test_records_of_the_cache_are_dropped_with_the_map_and_the_queries(one_lane):
    ctx = one_lane
    room, P, t0, q0 = _scene("rot", seed=73, n_q=3000, n_e=200)
    m = _matcher(ctx, room, P, "rot")
    m.pose_set(0, t0, q0)
    m.iterate(0, 6, MASK)
    n = room["q_xyz"].shape[0]
    assert (m.nn_cache_records(0, L.KIND_SURF, n)[:, 3] > 0).mean() > 0.7           # most queries left a usable record
    m.set_input_cloud(L.KIND_SURF, room["map_xyz"][::2].copy())                        # another map: positions in the sorted array mean something else
    with pytest.raises(L.LiliError):
        m.nn_cache_records(0, L.KIND_SURF, n)
    m.pose_set(0, t0, q0)
    m.iterate(0, 4, MASK)
    a = _snapshot_pose(m)
    ctx.set_option("nn_cache", 0)
    m2 = L.ScanToMapMatcher(ctx, P)
    m2.set_input_cloud(L.KIND_SURF, room["map_xyz"][::2].copy())
    m2.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m2.set_queries(0, L.KIND_SURF, room["q_xyz"]); m2.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    m2.pose_set(0, t0, q0)
    m2.iterate(0, 4, MASK)
    b = _snapshot_pose(m2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ctx.set_option("nn_cache", 1)
    m.set_queries(0, L.KIND_SURF, room["q_xyz"][:1500])                                # other queries: the records go as well
    with pytest.raises(L.LiliError):
        m.nn_cache_records(0, L.KIND_SURF, 1500)


def _snapshot_pose(m):
    t, q, st = m.pose_get(0)
    assert st == 0
    return np.array(t), np.array(q)


def _brute5(map_xyz, q):
    """exact 5-NN in f32 L2_Simple arithmetic (x, then y, then z; no FMA), ties by index"""
    d = np.zeros((q.shape[0], map_xyz.shape[0]), np.float32)
    for k in range(3):
        diff = (q[:, k:k + 1] - map_xyz[None, :, k]).astype(np.float32)
        d = (d + (diff * diff).astype(np.float32)).astype(np.float32)
    key = d.view(np.uint32).astype(np.uint64) << np.uint64(32) | np.arange(map_xyz.shape[0], dtype=np.uint64)[None, :]
    order = np.argsort(key, axis=1)[:, :5]
    return order.astype(np.int32), np.take_along_axis(d, order, 1)


@pytest.mark.parametrize("kind_of_map", ["lattice", "duplicates", "noisy"])
def test_random_walk_on_adversarial_maps_against_brute_force(one_lane, kind_of_map):
    """Neighbours after every move of a random walk (1 nm ... 3 cm per step, occasionally a jump) equal brute force: on a LATTICE map every query sits in a
    web of exact distance ties (the margin logic must refuse or order them by index), DUPLICATED map points tie at distance zero difference, a noisy map
    has the generic near-ties."""
    ctx = one_lane
    rng = np.random.default_rng({"lattice": 5, "duplicates": 6, "noisy": 7}[kind_of_map])
    g = np.arange(-6, 6.01, 0.4, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="ij")
    plane = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size, np.float32)], 1)
    wall = np.stack([X.ravel(), np.full(X.size, 6.0, np.float32), Y.ravel() + 6.0], 1)
    mp = np.concatenate([plane, wall]).astype(np.float32)
    if kind_of_map == "duplicates":
        mp = np.concatenate([mp, mp[rng.choice(mp.shape[0], mp.shape[0] // 3, replace=False)]])
    if kind_of_map == "noisy":
        mp = (mp + rng.normal(0, 0.03, mp.shape)).astype(np.float32)
    qs = np.concatenate([plane[rng.choice(plane.shape[0], 900)] + rng.uniform(-0.2, 0.2, (900, 3)), wall[rng.choice(wall.shape[0], 380)] + rng.uniform(-0.2, 0.2, (380, 3))]).astype(np.float32)
    if kind_of_map == "lattice":
        qs[:300] = plane[rng.choice(plane.shape[0], 300)] + np.array([0.2, 0.2, 0.0], np.float32)       # cell centres: four-fold exact ties
    P = L.make_params("frontend")
    ctx.set_option("nn_cache", 1)
    ctx.set_debug(True)
    m = L.ScanToMapMatcher(ctx, P)
    m.set_input_cloud(L.KIND_SURF, mp)
    m.set_queries(0, L.KIND_SURF, qs)
    t, q = np.zeros(3), np.array([1.0, 0, 0, 0])
    kept = []
    prev = None
    for step in range(40):
        scale = 10.0 ** rng.uniform(-9, -1.5) if step % 11 != 10 else 0.3
        t = t + rng.normal(0, scale, 3)
        ang = rng.normal(0, scale / 5.0, 3)
        dq = np.r_[1.0, 0.5 * ang]; dq /= np.linalg.norm(dq)
        q = synth.quat_mul(dq, q); q /= np.linalg.norm(q)
        m.find_corresponding_surf_features(0, q, t, want_count=False)
        idx, d2 = m.neighbors(0, L.KIND_SURF, qs.shape[0])
        pm = (synth.quat_rot(q, qs.astype(np.float64)) + t).astype(np.float32)
        bi, bd = _brute5(mp, pm)
        inside = bd[:, 4] <= np.float32(P.kd_max_radius)          # (beyond the gate the bounded search reports "no fifth neighbour")
        assert inside.mean() > 0.8
        assert np.array_equal(idx[inside], bi[inside]), (kind_of_map, step)
        assert np.array_equal(d2[inside].view(np.uint32), bd[inside].view(np.uint32)), (kind_of_map, step)
        rec = m.nn_cache_records(0, L.KIND_SURF, qs.shape[0])
        if prev is not None:
            kept.append(float((rec == prev).all(1).mean()))
        prev = rec
    if kind_of_map == "noisy":
        assert max(kept) > 0.9, kept          # tiny moves are served from the cache ...
    assert min(kept) < 0.2, kept              # ... jumps are searched
