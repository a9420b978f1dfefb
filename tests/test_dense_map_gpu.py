"""SURVEY §8d Config 2, variant B — a map far denser than the gate radius (0.05 m spacing against a 1 m gate: ~170 points per gate-sized
cell).  lili_map_set measures the density and builds the second, density-sized index; the association searches THAT index alone (round 6:
inner 27 fine cells per lane, then rings of super-rows by 16 lanes per query for the queries that are not settled; lili_s2m_dense.hip).
Exactness: neighbours (indices, f32 distances), records and the Gram are those of the oracle's exact kd-tree, and identical — bit for bit —
to the gate-sized index alone (option fine_grid = 0)."""
import time

import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


def _dense_room(seed=3, step=0.05, size=(12.0, 9.0, 4.0)):
    rng = np.random.default_rng(seed)
    sx, sy, sz = size

    def grid(u0, u1, v0, v1):
        U, V = np.meshgrid(np.arange(u0, u1, step), np.arange(v0, v1, step), indexing="ij")
        return U.ravel() + rng.uniform(-0.02, 0.02, U.size), V.ravel() + rng.uniform(-0.02, 0.02, V.size)
    faces = []
    for z in (0.0, sz):
        U, V = grid(-sx / 2, sx / 2, -sy / 2, sy / 2)
        faces.append(np.stack([U, V, np.full_like(U, z) + rng.normal(0, 0.003, U.size)], 1))
    for sg in (-1, 1):
        U, V = grid(-sy / 2, sy / 2, 0, sz)
        faces.append(np.stack([np.full_like(U, sg * sx / 2) + rng.normal(0, 0.003, U.size), U, V], 1))
        U, V = grid(-sx / 2, sx / 2, 0, sz)
        faces.append(np.stack([U, np.full_like(U, sg * sy / 2) + rng.normal(0, 0.003, U.size), V], 1))
    pts = np.concatenate(faces).astype(np.float32)
    # a sparse patch as well (a hole in the dense coverage): one wall is thinned to ~0.5 m spacing in a strip
    strip = (np.abs(pts[:, 0] - sx / 2) < 0.05) & (pts[:, 1] > 0) & (pts[:, 1] < 3)
    keep = ~strip | (rng.uniform(size=pts.shape[0]) < 0.01)
    return pts[keep]


def test_dense_map_fine_index_is_exact(gpu_ctx, oracle):
    mp = _dense_room()
    assert mp.shape[0] > 140_000
    rng = np.random.default_rng(7)
    pick = rng.choice(mp.shape[0], 6000)
    qw = mp[pick].astype(np.float64) + rng.normal(0, 0.01, (6000, 3)) + rng.uniform(-0.1, 0.1, (6000, 3))
    qw[:300] += rng.uniform(0.6, 2.5, (300, 3))                       # far from every surface: fail the gate / need the gate-sized index
    qw[300:600, 0] = 6.0 + rng.uniform(-0.05, 0.05, 300); qw[300:600, 1] = rng.uniform(0, 3, 300)   # next to the thinned strip
    t_true = np.array([0.4, -0.3, 1.5]); ang = np.radians(15.0)
    q_true = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)])
    q_local = synth.quat_rot(q_true * np.array([1, -1, -1, -1]), qw - t_true).astype(np.float32)
    P, PO = L.make_params("rot"), oracle.params("rot")
    res = {}
    try:
        gpu_ctx.set_option("map_guess_box", 0)      # both builds on the measured box: with 1 862 cells the timing below depends on where the walls fall inside the cells
        for fine in (1, 0):
            gpu_ctx.set_option("fine_grid", fine)
            gpu_ctx.set_debug(True)
            m = L.ScanToMapMatcher(gpu_ctx, P)
            m.set_input_cloud(L.KIND_SURF, mp)
            occ, fcell, fr2 = m.map_density(L.KIND_SURF)
            if fine:
                assert occ > 100 and 0.04 < fcell < 0.3 and fr2 > 4 * 0.04 ** 2, (occ, fcell, fr2)
            else:
                assert fcell == 0.0
            m.set_queries(0, L.KIND_SURF, q_local)
            n = m.find_corresponding_surf_features(0, q_true, t_true)
            idx, d2 = m.neighbors(0, L.KIND_SURF, 6000)
            rec = m.surf_records(0, 6000)
            tb, qb = L.api.body_pose_from_lidar(t_true, q_true, P)
            G, cost, counts = m.linearize(0, tb, qb, L.MASK_SURF)
            # timing on the queries a scan consists of — points ON the mapped surfaces (slot 1; the 600 far / border queries above need the
            # gate-sized index whatever comes first, and one such lane costs its wave the whole dense neighbourhood)
            gpu_ctx.set_debug(False)
            m.set_queries(1, L.KIND_SURF, q_local[600:])
            m.find_corresponding_surf_features(1, q_true, t_true)
            gpu_ctx.sync()
            tic = time.perf_counter()
            for _ in range(10):
                m.find_corresponding_surf_features(1, q_true, t_true, want_count=False)
            gpu_ctx.sync()
            dt = (time.perf_counter() - tic) / 10
            res[fine] = (n, idx, d2, rec, G, cost, dt, occ, fcell)
    finally:
        gpu_ctx.set_option("map_guess_box", 1)
        gpu_ctx.set_option("fine_grid", 1)
        gpu_ctx.set_debug(False)
    print(f"dense map: {mp.shape[0]} points, mean occupancy {res[1][7]:.0f} per gate-sized cell, fine cell {res[1][8]:.3f} m; "
          f"associate 5400 on-surface queries: fine index {res[1][6] * 1e6:.0f} us, gate-sized index alone {res[0][6] * 1e6:.0f} us")
    # against the oracle's exact kd-tree
    tree = oracle.KdTree(mp)
    o = oracle.associate_surf(tree, None, q_local, None, q_true, t_true, PO)
    n, idx, d2, rec, G, cost, _, _, _ = res[1]
    assert n == o["count"] and n > 4000
    inside = o["nn_d2"][:, 4] < 1.0
    assert inside.sum() > 5000 and (~inside).sum() > 100
    assert np.array_equal(idx[inside], o["nn_idx"][inside]) and np.array_equal(d2[inside], o["nn_d2"][inside])
    assert np.array_equal(rec["query_index"], np.nonzero(o["valid"])[0])
    # ... and identical to the gate-sized index alone, bit for bit (same neighbours -> same fits -> same records -> same Gram)
    a, b = res[1], res[0]
    assert a[0] == b[0] and np.array_equal(a[1][inside], b[1][inside]) and np.array_equal(a[2][inside], b[2][inside])
    for k in ("query_index", "n", "d", "score"):
        assert np.array_equal(a[3][k], b[3][k]), k
    assert np.array_equal(a[4], b[4]) and a[5] == b[5]
    assert a[6] < b[6]                                                # and it is the faster of the two on this map


def _brute_knn5(mp, qw):
    """exact (d2, index)-ordered 5-NN with FLANN's f32 L2_Simple arithmetic, by brute force (small query sets)."""
    idx = np.zeros((qw.shape[0], 5), np.int32); d2 = np.zeros((qw.shape[0], 5), np.float32)
    m = mp.astype(np.float32)
    for i, q in enumerate(qw.astype(np.float32)):
        dx = q[0] - m[:, 0]; dy = q[1] - m[:, 1]; dz = q[2] - m[:, 2]
        d = (dx * dx + dy * dy) + dz * dz
        o = np.lexsort((np.arange(m.shape[0]), d))[:5]
        idx[i] = o; d2[i] = d[o]
    return idx, d2


def test_dense_map_borders_focus_box_and_both_selectors(gpu_ctx, oracle):
    """Round 6: the dense-map association searches the fine index alone — inner 27 fine cells per lane, then rings of super-rows by 16 lanes per query.  Exercised here:
    queries outside the grid and at its faces, a NaN query, queries between 0.2 and 0.9 m off every surface (several ring levels), a focus box that leaves most of the
    map without super-rows (blocks walked row by row), both kinds, and the exact-key selector against the bucket-key one (LILI_DEBUG bit 32768)."""
    import os
    mp = _dense_room(seed=5, size=(6.0, 5.0, 3.0))
    rng = np.random.default_rng(11)
    n_q = 1500
    qw = mp[rng.choice(mp.shape[0], n_q)].astype(np.float64) + rng.normal(0, 0.01, (n_q, 3))
    qw[:400] += rng.uniform(-1, 1, (400, 3)) * rng.uniform(0.2, 0.9, (400, 1))     # off the surfaces, inside and outside the room
    qw[400:430] = rng.uniform(-1, 1, (30, 3)) * np.array([3.6, 3.1, 0.4]) + np.array([0, 0, -0.5])      # below the floor: outside the grid in z
    qw[430:460, 0] = 3.0 + rng.uniform(0.0, 2.6, 30)                                 # beyond the +x wall, some of them beyond the gate
    qw[460] = np.nan
    t_true = np.array([0.2, -0.1, 1.2]); q_true = np.array([1.0, 0, 0, 0])
    q_local = (qw - t_true).astype(np.float32)
    qm = q_local.astype(np.float64) + t_true                                         # what the device sees after the (identity-rotation) transform, up to f32 rounding
    P = L.make_params("rot")
    want_idx, want_d2 = _brute_knn5(mp, (q_local.astype(np.float64) + t_true).astype(np.float32))
    results = {}
    try:
        gpu_ctx.set_option("map_guess_box", 0)
        gpu_ctx.set_debug(True)
        for name, focus, dbg in (("whole", None, "0"), ("focus", ((1.0, 0.5, 1.0), 1.2), "0"), ("exact", None, "32768")):
            os.environ["LILI_DEBUG"] = dbg
            m = L.ScanToMapMatcher(gpu_ctx, P)
            m.map_focus(*focus) if focus else m.map_focus(None)
            m.set_input_cloud(L.KIND_SURF, mp)
            m.set_input_cloud(L.KIND_EDGE, mp, max_sq_radius=1.0)
            assert m.map_density(L.KIND_SURF)[1] > 0 and m.map_density(L.KIND_EDGE)[1] > 0      # both kinds got the fine index
            out = {}
            for kind, find in ((L.KIND_SURF, m.find_corresponding_surf_features), (L.KIND_EDGE, m.find_corresponding_corner_features)):
                m.set_queries(0, kind, q_local)
                n = find(0, q_true, t_true)
                idx, d2 = m.neighbors(0, kind, n_q)
                out[kind] = (n, idx, d2)
            rec = m.surf_records(0, n_q)
            results[name] = (out, rec)
    finally:
        os.environ.pop("LILI_DEBUG", None)
        gpu_ctx.set_option("map_guess_box", 1)
        gpu_ctx.set_debug(False)
        L.ScanToMapMatcher(gpu_ctx, P).map_focus(None)
    gate = 1.0
    inside = want_d2[:, 4] < gate
    inside[460] = False
    assert inside.sum() > 1300 and (~inside).sum() > 10
    for name, (out, rec) in results.items():
        for kind in (L.KIND_SURF, L.KIND_EDGE):
            n, idx, d2 = out[kind]
            assert np.array_equal(idx[inside], want_idx[inside]), (name, kind, np.nonzero((idx != want_idx).any(1) & inside)[0][:10])
            assert np.array_equal(d2[inside], want_d2[inside]), (name, kind)
            assert (idx[460] == -1).all()
    base_out, base_rec = results["whole"]
    for name in ("focus", "exact"):
        out, rec = results[name]
        for kind in (L.KIND_SURF, L.KIND_EDGE):
            assert out[kind][0] == base_out[kind][0], (name, kind)
        for k in ("query_index", "n", "d", "score"):
            assert np.array_equal(rec[k], base_rec[k]), (name, k)


def test_dense_map_livox_flavour_reads_reflectivity_through_the_fine_index(gpu_ctx, oracle):
    """The Livox back-end flavour weights the plane fit by reflectivity differences (L/src/BackendFusion.cpp:1617-1638): on a dense map the neighbours' reflectivity comes from the
    fine index's own sorted auxiliary array.  Records equal the gate-sized index alone bit for bit, counts and neighbours equal the oracle's."""
    mp = _dense_room(seed=9, size=(7.0, 6.0, 3.0))
    rng = np.random.default_rng(21)
    refl = (10.0 + 2.0 * np.sin(0.7 * mp[:, 0]) + 2.0 * np.cos(0.5 * mp[:, 1]) + rng.normal(0, 0.3, mp.shape[0])).astype(np.float32)      # smooth: the five neighbours pass reflect_thres
    n_q = 3000
    pick = rng.choice(mp.shape[0], n_q)
    qw = mp[pick].astype(np.float64) + rng.normal(0, 0.01, (n_q, 3)) + rng.uniform(-0.15, 0.15, (n_q, 3))
    q_refl = (refl[pick] + rng.normal(0, 0.4, n_q)).astype(np.float32)
    t_true = np.array([0.3, 0.2, 1.4]); ang = np.radians(-10.0)
    q_true = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)])
    q_local = synth.quat_rot(q_true * np.array([1, -1, -1, -1]), qw - t_true).astype(np.float32)
    P, PO = L.make_params("livox"), oracle.params("livox")
    res = {}
    try:
        gpu_ctx.set_option("map_guess_box", 0)
        for fine in (1, 0):
            gpu_ctx.set_option("fine_grid", fine)
            m = L.ScanToMapMatcher(gpu_ctx, P)
            m.set_input_cloud(L.KIND_SURF, np.c_[mp, refl])
            assert (m.map_density(L.KIND_SURF)[1] > 0) == bool(fine)
            m.set_queries(0, L.KIND_SURF, np.c_[q_local, q_refl])
            n = m.find_corresponding_surf_features(0, q_true, t_true)
            res[fine] = (n, m.surf_records(0, n_q))
    finally:
        gpu_ctx.set_option("map_guess_box", 1)
        gpu_ctx.set_option("fine_grid", 1)
    assert res[1][0] == res[0][0] > 500
    for k in ("query_index", "cp", "n", "d", "score"):
        assert np.array_equal(res[1][1][k], res[0][1][k]), k
    o = oracle.associate_surf(oracle.KdTree(mp), refl, q_local, q_refl, q_true, t_true, PO)
    assert res[1][0] == o["count"] and np.array_equal(res[1][1]["query_index"], np.nonzero(o["valid"])[0])
