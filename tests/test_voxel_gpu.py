"""GPU parity of the VoxelGrid filter and the local-map assembly vs the oracle's PCL restatement
(in-order accumulation mode): voxel membership, order, counts and f32 centroids bit-exact."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,leaf,scale", [(1000, 0.4, 5.0), (200_000, 0.4, 60.0), (1_500_000, 0.2, 40.0), (50_000, 0.6, 300.0)])
def test_voxel_filter_matches_pcl_restatement(gpu_ctx, oracle, n, leaf, scale):
    rng = np.random.default_rng(n)
    pts = np.concatenate([rng.uniform(-scale, scale, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
    pts[: n // 10] = pts[n // 10: 2 * (n // 10)][: n // 10]      # exact duplicates -> multi-point voxels for sure
    g, gc = L.api.voxel_filter(gpu_ctx, pts, leaf)
    o, oc = oracle.voxel_grid(pts, leaf, stable=True)
    assert g.shape == o.shape and g.shape[0] < n
    assert np.array_equal(gc, oc)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    # PCL's own (unstable) sort may only change the summation order inside a voxel
    o2, oc2 = oracle.voxel_grid(pts, leaf, stable=False)
    assert np.array_equal(oc2, oc)
    np.testing.assert_allclose(o2, o, rtol=1e-6, atol=1e-5 * scale)


@pytest.mark.parametrize("n", [1, 255, 1025, 20_000, 262_144, 262_145, 1_100_000])
def test_short_sorts_without_the_scan_launch(gpu_ctx, oracle, n):
    """Radix passes of at most 256 tiles (sorts up to ~1 M keys) derive the scatter offsets inside k_sort_scatter8 (option sort_fused_scan, on) instead of
    in a scan launch between the two kernels: same order, same centroids, on both sides of the tile-size switch (262 144 keys), beyond the limit
    (1.1 M keys = 269 tiles: the scan launch) and against the oracle."""
    rng = np.random.default_rng(n + 7)
    pts = np.concatenate([rng.uniform(-60, 60, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
    res = []
    try:
        for opt in (1, 0):
            gpu_ctx.set_option("sort_fused_scan", opt)
            res.append(L.api.voxel_filter(gpu_ctx, pts, 0.4))
    finally:
        gpu_ctx.set_option("sort_fused_scan", 1)
    o, oc = oracle.voxel_grid(pts, 0.4, stable=True)
    for g, gc in res:
        assert np.array_equal(gc, oc) and np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_voxel_filter_edge_cases(gpu_ctx, oracle):
    one = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    g, c = L.api.voxel_filter(gpu_ctx, one, 0.4)
    assert np.array_equal(g, one) and c[0] == 1
    g, c = L.api.voxel_filter(gpu_ctx, np.zeros((0, 4), np.float32), 0.4)
    assert g.shape[0] == 0
    with pytest.raises(L.LiliError):                              # PCL refuses too ("leaf size is too small")
        L.api.voxel_filter(gpu_ctx, np.array([[0, 0, 0, 0], [1e6, 1e6, 1e6, 0]], np.float32), 0.01)


def test_voxel_filter_skips_non_finite_points(gpu_ctx, oracle):
    """NaN / Inf points used to get an arbitrary voxel key (UB cast) and could split or duplicate voxels.  PCL's
    VoxelGrid skips non-finite points of a non-dense cloud: the result must equal the filter of the finite subset."""
    rng = np.random.default_rng(77)
    n = 60_000
    pts = np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
    bad = rng.choice(n, 900, replace=False)
    pts[bad[:300], 0] = np.nan
    pts[bad[300:600], 1] = np.inf
    pts[bad[600:], 2] = -np.inf
    fin = np.isfinite(pts[:, :3]).all(1)
    g, gc = L.api.voxel_filter(gpu_ctx, pts, 0.4)
    o, oc = oracle.voxel_grid(np.ascontiguousarray(pts[fin]), 0.4, stable=True)
    assert g.shape == o.shape and np.array_equal(gc, oc)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    assert gc.sum() == fin.sum()
    with pytest.raises(L.LiliError):                              # nothing finite at all
        L.api.voxel_filter(gpu_ctx, np.full((4, 4), np.nan, np.float32), 0.4)


def test_local_map_assembly_and_match(gpu_ctx, oracle):
    """push 5 keyframes (ring of width 4), commit, then match a scan against the device-built map: identical to
    assembling / filtering on the host with the oracle and calling set_input_cloud."""
    room = synth.make_room(seed=23, n_query=3000, n_edge_query=50)
    rng = np.random.default_rng(1)
    kfs, poses = [], []
    for k in range(5):
        sel = rng.choice(room["map_xyz"].shape[0], 6000, replace=False)
        t = np.array([0.3 * k, -0.1 * k, 0.02 * k])
        ang = 0.05 * k
        q = np.array([np.cos(ang / 2), 0.0, 0.0, np.sin(ang / 2)])
        world = room["map_xyz"][sel].astype(np.float64)
        local = synth.quat_rot(q * np.array([1, -1, -1, -1]), world - t)                  # keyframe-frame features
        kfs.append(np.concatenate([local, rng.uniform(1, 20, (6000, 1))], 1).astype(np.float32))
        poses.append((t, q))
    lm = L.LocalMap(gpu_ctx, L.KIND_SURF, width=4, leaf=0.4, max_sq_radius=1.0)
    for f, (t, q) in zip(kfs, poses):
        lm.push(f, t, q)
    n_raw, n_map = lm.commit()
    assert n_raw == 4 * 6000                                       # the oldest keyframe was dropped
    # host restatement: transformCloud (f64 -> f32), concatenate oldest..newest, VoxelGrid
    parts = []
    for f, (t, q) in list(zip(kfs, poses))[1:]:
        w = np.stack([oracle.qrot(q, p[:3].astype(np.float64)) + t for p in f[:200]])     # spot-check the transform
        parts.append(np.concatenate([(synth.quat_rot(q, f[:, :3].astype(np.float64)) + t).astype(np.float32), f[:, 3:4]], 1))
        assert np.array_equal(parts[-1][:200, :3], w.astype(np.float32))
    cat = np.concatenate(parts, 0)
    ref_map, _ = oracle.voxel_grid(cat, 0.4, stable=True)
    assert n_map == ref_map.shape[0]
    P, PO = L.make_params("frontend"), oracle.params("frontend")
    m = L.ScanToMapMatcher(gpu_ctx, P)
    gpu_ctx.set_debug(True)
    m.set_queries(0, L.KIND_SURF, room["q_xyz"])
    n1 = m.find_corresponding_surf_features(0, room["q_true"], room["t_true"])
    idx1, d1 = m.neighbors(0, L.KIND_SURF, room["q_xyz"].shape[0])
    rs = oracle.associate_surf(oracle.KdTree(ref_map[:, :3]), None, room["q_xyz"], None, room["q_true"], room["t_true"], PO)
    assert n1 == rs["count"] > 500
    inside = rs["nn_d2"][:, 4] < 1.0
    assert np.array_equal(idx1[inside], rs["nn_idx"][inside]) and np.array_equal(d1[inside], rs["nn_d2"][inside])


def test_device_resident_pipeline_equals_host_round_trip(gpu_ctx, oracle):
    """scan -> lili_extract_rot -> (device views) -> VoxelGrid(0.4) -> queries -> 5 outer iterations, without the
    features leaving HBM, must equal the same pipeline with host round trips between the stages."""
    import ctypes as C
    import torch
    w = synth.make_workload(n_map=400_000, n_az=600, half_extent=(150.0, 150.0))
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 9.0, np.float32)], 1).astype(np.float32)
    P = L.make_params("rot")
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(1), 0.2, 1.0)
    ex = L.RotExtractor(gpu_ctx, ds_rate=2)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
    # host round trips
    feats = ex.extract(raw)
    qh, _ = L.api.voxel_filter(gpu_ctx, feats["surf"], 0.4)
    m.set_queries(0, L.KIND_SURF, qh)
    m.pose_set(0, t0, q0); m.iterate(0, 5, L.MASK_SURF)
    t_h, q_h, st = m.pose_get(0)
    assert st == 0
    # device resident
    d_scan = torch.from_numpy(raw).cuda()
    cloud = L.api.cloud_from_device(d_scan.data_ptr(), raw.shape[0], 16, 12)
    outs = [L.api.FeatureOut(None, 0, 16, L.api.MEM_HOST, 0) for _ in range(3)]
    qi, ql = np.array([1.0, 0, 0, 0]), np.array([1.0, 0, 0, 0])
    gpu_ctx._chk(gpu_ctx.lib.lili_extract_rot(gpu_ctx.h, C.byref(cloud), qi.ctypes.data_as(C.c_void_p), ql.ctypes.data_as(C.c_void_p),
                                              C.byref(ex.params), C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
    _, _, d_surf = L.api.extract_rot_device(gpu_ctx)
    assert d_surf.n == feats["surf"].shape[0]
    d_q = torch.empty((d_surf.n, 4), dtype=torch.float32, device="cuda")
    qd = L.api.voxel_filter_device(gpu_ctx, d_surf, 0.4, d_q.data_ptr(), d_surf.n)
    assert qd.n == qh.shape[0]
    m.set_queries(0, L.KIND_SURF, qd)
    m.pose_set(0, t0, q0); m.iterate(0, 5, L.MASK_SURF)
    t_d, q_d, st = m.pose_get(0)
    assert st == 0
    assert np.array_equal(t_h, t_d) and np.array_equal(q_h, q_d)
    assert np.linalg.norm(t_d - tb) < 0.5 * np.linalg.norm(t0 - tb)


def test_incremental_local_map_equals_full_rebuild(gpu_ctx):
    """lili_localmap_commit keeps the ring sorted by voxel and merges ONE keyframe per step (the reference pops / pushes one
    keyframe per step, L/src/BackendFusion.cpp:1407-1477).  A/B against the rebuild-every-time path (option localmap_incremental = 0): the
    same map, bit for bit and in the same order, at every step — while the ring fills, in the pop / push steady state, with keyframes of
    different sizes, an empty one, NaN points, two keyframes pushed between commits, a leaf change, and with both radix digit widths."""
    rng = np.random.default_rng(5)

    def keyframe(k):
        n = int(rng.integers(300, 4000)) if k != 6 else 0
        pts = np.concatenate([rng.uniform(-12, 12, (n, 2)) + [0.9 * k, -0.4 * k], rng.uniform(-1.5, 2.5, (n, 1)), rng.uniform(0, 30, (n, 1))], 1).astype(np.float32)
        if n and k % 4 == 1:
            pts[rng.integers(0, n, 5), rng.integers(0, 3, 5)] = np.nan
        ang = 0.07 * k
        return pts, np.array([0.5 * k, 0.2 * k, 0.01 * k]), np.array([np.cos(ang / 2), 0.0, 0.0, np.sin(ang / 2)])
    kfs = [keyframe(k) for k in range(18)]
    # schedule: (keyframes pushed before this commit, leaf)
    sched = [(1, 0.4)] * 3 + [(2, 0.4)] + [(1, 0.4)] * 6 + [(1, 0.2)] + [(1, 0.2)] * 2 + [(3, 0.2)]
    runs = {}
    try:
        for mode, digits in (("full", 8), ("incremental", 8), ("incremental4", 4)):
            gpu_ctx.set_option("localmap_incremental", 0 if mode == "full" else 1)
            gpu_ctx.set_option("sort_digit_bits", digits)
            lm = L.LocalMap(gpu_ctx, L.KIND_SURF, width=5, leaf=0.4, max_sq_radius=1.0)
            inc0, full0 = lm.stats()
            k, maps = 0, []
            for n_push, leaf in sched:
                for _ in range(n_push):
                    lm.push(*kfs[k]); k += 1
                lm.leaf = leaf
                n_raw, n_map = lm.commit()
                maps.append((n_raw, lm.get(n_map + 1)))
            inc1, full1 = lm.stats()
            runs[mode] = (maps, inc1 - inc0, full1 - full0)
    finally:
        gpu_ctx.set_option("localmap_incremental", 1)
        gpu_ctx.set_option("sort_digit_bits", 8)
    ref, n_inc_ref, n_full_ref = runs["full"]
    assert n_inc_ref == 0 and n_full_ref == len(sched)
    for mode in ("incremental", "incremental4"):
        maps, n_inc, n_full = runs[mode]
        assert n_inc >= 9 and n_full <= 5, (n_inc, n_full)              # rebuilds only: first commit, leaf change, three keyframes pending (+ nothing else)
        for (ra, a), (rb, b) in zip(ref, maps):
            assert ra == rb and a.shape == b.shape and a.shape[0] > 100
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_index_behind_an_incremental_commit_equals_a_plain_map_set(gpu_ctx):
    """After an incremental step lili_localmap_commit indexes the centroids where they lie, with the bounding box that travelled with their count
    (lili_map_set_hinted: no ingestion copy, no read-back before the grid).  The index must be the one lili_map_set builds from the same points:
    neighbours, distances, records and the Gram of a scan matched against it are identical bit for bit."""
    rng = np.random.default_rng(11)

    def keyframe(k):
        n = int(rng.integers(1500, 4000))
        pts = np.concatenate([rng.uniform(-15, 15, (n, 2)) + [0.8 * k, 0.3 * k], rng.normal(0, 0.03, (n, 1)), rng.uniform(0, 30, (n, 1))], 1).astype(np.float32)
        pts[: n // 3, 2] = rng.uniform(0, 3, n // 3)          # some structure above the ground plane
        return pts, np.array([0.0, 0.0, 0.0]), np.array([1.0, 0.0, 0.0, 0.0])
    P = L.make_params("rot")
    lm = L.LocalMap(gpu_ctx, L.KIND_SURF, width=6, leaf=0.4, max_sq_radius=1.0)
    inc0, _ = lm.stats()
    for k in range(9):
        lm.push(*keyframe(k))
        n_raw, n_map = lm.commit()
    inc1, _ = lm.stats()
    assert inc1 - inc0 >= 7                                   # the last commits were incremental steps: the hinted build
    pts = lm.get(n_map + 1)
    assert pts.shape[0] == n_map > 3000
    q = (pts[rng.integers(0, n_map, 3000), :3] + rng.normal(0, 0.05, (3000, 3))).astype(np.float32)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    t0, q0 = np.array([0.02, -0.01, 0.03]), np.array([1.0, 0.0, 0.0, 0.0])
    res = []
    gpu_ctx.set_debug(True)
    try:
        for plain in (False, True):
            if plain:
                m.set_input_cloud(L.KIND_SURF, np.ascontiguousarray(pts))     # lili_map_set on a host copy of the same centroids
            m.set_queries(0, L.KIND_SURF, q)
            m.pose_set(0, t0, q0)
            m.associate_dev(0, L.MASK_SURF)
            idx, d2 = m.neighbors(0, L.KIND_SURF, q.shape[0])
            rec = m.surf_records(0, q.shape[0])
            G, cost, counts = m.linearize(0, t0, q0, L.MASK_SURF)
            res.append((idx, d2, rec, G, cost, counts))
    finally:
        gpu_ctx.set_debug(False)
    a, b = res
    inside = a[1][:, 4] < 1.0
    assert inside.sum() > 1500 and a[5][0] > 500
    assert np.array_equal(a[0][inside], b[0][inside]) and np.array_equal(a[1][inside], b[1][inside])
    assert a[2]["count"] == b[2]["count"] and np.array_equal(a[2]["n"], b[2]["n"]) and np.array_equal(a[2]["query_index"], b[2]["query_index"])
    assert np.array_equal(a[3], b[3]) and a[4] == b[4] and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("n,leaf,scale", [(1, 0.4, 5.0), (63, 0.4, 5.0), (64, 0.4, 30.0), (65, 0.2, 30.0), (3000, 0.4, 60.0), (4097, 0.4, 100.0), (8192, 0.6, 200.0), (8192, 0.05, 2.0)])
def test_single_launch_filter_of_small_clouds_equals_the_general_chain(oracle, n, leaf, scale):
    """k_voxel_small (round 5: clouds of <= 8192 points — a Livox frame's features — in ONE single-workgroup launch) against the general chain of launches
    (option voxel_small = 0) and against the oracle's PCL restatement: same voxels, same order, same counts, centroids bit for bit; duplicates, non-finite
    points, many points per voxel (leaf >> spacing) and one point per voxel."""
    rng = np.random.default_rng(1000 + n)
    pts = np.concatenate([rng.uniform(-scale, scale, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
    if n >= 64:
        pts[: n // 8] = pts[n // 8: 2 * (n // 8)][: n // 8]          # exact duplicates
        pts[n // 2] = [np.nan, 0, 0, 1]; pts[n // 2 + 1] = [0, np.inf, 0, 1]; pts[n - 1] = [0, 0, -np.inf, 1]
    res = []
    ctx = L.Context(0)          # (its own context: the session's shared one may carry another test's A/B options)
    try:
        # a larger cloud first, so that the filter's scratch buffers hold another cloud's keys / flags / counts when the small one arrives
        big = np.concatenate([rng.uniform(-80, 80, (50_000, 2)), rng.normal(0, 0.5, (50_000, 1)), rng.uniform(0, 25, (50_000, 1))], 1).astype(np.float32)
        L.api.voxel_filter(ctx, big, 0.3)
        for opt in (1, 0):
            ctx.set_option("voxel_small", opt)
            res.append(L.api.voxel_filter(ctx, pts, leaf))
    finally:
        ctx.close()
    (g1, c1), (g0, c0) = res
    assert g1.shape == g0.shape and np.array_equal(c1, c0) and np.array_equal(g1.view(np.uint32), g0.view(np.uint32))
    fin = np.isfinite(pts[:, :3]).all(1)
    o, oc = oracle.voxel_grid(np.ascontiguousarray(pts[fin]), leaf, stable=True)          # PCL skips non-finite points of a non-dense cloud
    assert g1.shape == o.shape and np.array_equal(c1, oc) and np.array_equal(g1.view(np.uint32), o.view(np.uint32))


def test_single_launch_filter_hands_an_index_overflow_to_the_general_path(gpu_ctx):
    """a leaf far too small for the extent: PCL's "leaf size too small" (voxel index beyond int32) is reported whichever path sees the cloud first"""
    pts = np.array([[0, 0, 0, 1], [5000, 5000, 5000, 1]], np.float32)
    with pytest.raises(L.LiliError):
        L.api.voxel_filter(gpu_ctx, pts, 0.001)


def test_filter_with_guessed_key_bits_equals_the_measured_filter(oracle):
    """Round 5: a VoxelGrid of more than 8192 points whose leaf size has been filtered before keeps its bounding box on the device and sorts by as many key bits as that
    filter needed (k_vox_key_dev; option voxel_guess_bits).  Same voxels, order, counts and centroids as the measured filter and the oracle — for a cloud the guess
    holds for, one that needs FEWER bits, one that needs MORE (the device notices, the filter is repeated the measured way), non-finite points, and the error cases."""
    rng = np.random.default_rng(2024)

    def cloud(n, scale):
        pts = np.concatenate([rng.uniform(-scale, scale, (n, 2)), rng.normal(0, 0.5, (n, 1)), rng.uniform(0, 25, (n, 1))], 1).astype(np.float32)
        pts[: n // 10] = pts[n // 10: 2 * (n // 10)][: n // 10]
        pts[n // 2] = [np.nan, 0, 0, 1]; pts[n // 3] = [0, -np.inf, 0, 1]
        return pts
    seq = [cloud(20_000, 50.0), cloud(24_000, 52.0), cloud(20_000, 4.0), cloud(30_000, 900.0), cloud(30_000, 880.0), cloud(9_000, 50.0)]
    ctx = L.Context(0)
    try:
        got = [L.api.voxel_filter(ctx, p, 0.4) for p in seq]
        guesses, misses = L.api.voxel_filter_stats(ctx)
        # the first filter measures; the others guess; 50 -> 4 m needs fewer bits (the guess holds), 4 -> 900 m more (a miss)
        assert guesses == len(seq) - 1 and misses == 1, (guesses, misses)
        with pytest.raises(L.LiliError):      # PCL's int32 overflow, reported through the guessed path's fallback
            L.api.voxel_filter(ctx, np.concatenate([seq[0], np.array([[3e5, 3e5, 3e5, 0]], np.float32)]), 0.4)
        with pytest.raises(L.LiliError):      # no finite point at all
            L.api.voxel_filter(ctx, np.full((9000, 4), np.nan, np.float32), 0.4)
        again = L.api.voxel_filter(ctx, seq[0], 0.4)      # and the filter still works afterwards
        ctx.set_option("voxel_guess_bits", 0)
        ref = [L.api.voxel_filter(ctx, p, 0.4) for p in seq]
        assert L.api.voxel_filter_stats(ctx)[0] == guesses + 3      # (the two error cases and `again` guessed; nothing since)
    finally:
        ctx.close()
    for p, (g, c), (g0, c0) in zip(seq, got, ref):
        assert g.shape == g0.shape and np.array_equal(c, c0) and np.array_equal(g.view(np.uint32), g0.view(np.uint32))
        fin = np.isfinite(p[:, :3]).all(1)
        o, oc = oracle.voxel_grid(np.ascontiguousarray(p[fin]), 0.4, stable=True)
        assert g.shape == o.shape and np.array_equal(c, oc) and np.array_equal(g.view(np.uint32), o.view(np.uint32))
    assert np.array_equal(again[0].view(np.uint32), got[0][0].view(np.uint32))
