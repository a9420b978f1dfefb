"""GPU parity of the LOAM-style (LiLi-OM-ROT) extractor vs the oracle: feature INDICES bit-exact (north star),
deskewed cloud / curvature bit-exact f32, voxel-filtered surf points bit-exact vs the oracle's in-order mode."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


def _raw_scan(n_az, seed=0):
    w = synth.make_workload(n_map=300_000, n_az=n_az, half_extent=(150.0, 150.0), seed=synth.SEED_SCENE + seed)
    scan = w["scan_xyz"]
    refl = np.random.default_rng(seed).integers(1, 255, scan.shape[0]).astype(np.float32)
    return np.concatenate([scan, refl[:, None]], 1).astype(np.float32)


def _compare(g, o):
    assert np.array_equal(g["full_src"], o["full_src"])
    assert np.array_equal(g["ring_start"], o["ring_start"]) and np.array_equal(g["ring_end"], o["ring_end"])
    assert np.array_equal(g["full"].view(np.uint32), o["full"].view(np.uint32))          # deskewed cloud, bit-exact
    assert np.array_equal(g["curvature"].view(np.uint32), o["curvature"].view(np.uint32))
    assert np.array_equal(g["label"], o["label"])
    for k in ("edge_idx", "sharp_idx", "flat_idx", "lessflat_idx"):
        assert np.array_equal(g[k], o[k]), k
    assert np.array_equal(g["edge"].view(np.uint32), o["full"][o["edge_idx"]].view(np.uint32))
    assert np.array_equal(g["surf_cnt"], o["surf_cnt"])
    assert np.array_equal(g["surf"].view(np.uint32), o["surf"].view(np.uint32))


@pytest.mark.parametrize("n_az,ds_rate", [(391, 4), (1042, 1), (3125, 4)])
def test_rot_extractor_parity(gpu_ctx, oracle, n_az, ds_rate):
    raw = _raw_scan(n_az)
    q_lb = [0.7071, 0.0, 0.0, 0.7071]                      # R/config/config_fr_iosb.yaml:38-41 (not unit norm)
    ang = 0.03
    q_imu = [np.cos(ang / 2), np.sin(ang / 2) * 0.3, -np.sin(ang / 2) * 0.5, np.sin(ang / 2) * 0.81]   # un-normalised, like deltaQ products
    ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=ds_rate)
    g = ex.extract(raw, q_imu, q_lb, debug=True)
    o = oracle.extract_rot(raw, q_imu, q_lb, oracle.rot_params(ds_rate=ds_rate, atan_mode=2, stable_sort=1))   # glibc float atan / atan2 restated
    assert len(o["edge_idx"]) > 20 and len(o["surf"]) > 200
    _compare(g, o)
    # the other definition ("rot_atan" = 1: f64 functions rounded to f32) against the oracle's mode 1
    gpu_ctx.set_option("rot_atan", 1)
    try:
        g1 = ex.extract(raw, q_imu, q_lb, debug=True)
    finally:
        gpu_ctx.set_option("rot_atan", 2)
    _compare(g1, oracle.extract_rot(raw, q_imu, q_lb, oracle.rot_params(ds_rate=ds_rate, atan_mode=1, stable_sort=1)))
    # the literal oracle (THIS libm's float overloads, std::sort): the deskewed cloud is bit-identical to the GPU's — the restated
    # routines ARE this libm's — and without sort ties so is every selected feature
    lit = oracle.extract_rot(raw, q_imu, q_lb, oracle.rot_params(ds_rate=ds_rate, atan_mode=0, stable_sort=0))
    assert np.array_equal(lit["full_src"], o["full_src"])
    assert np.array_equal(lit["full"].view(np.uint32), g["full"].view(np.uint32))
    if lit["n_ties"] == 0:
        assert np.array_equal(lit["edge_idx"], o["edge_idx"]) and np.array_equal(lit["label"], o["label"])
        assert np.array_equal(lit["lessflat_idx"], o["lessflat_idx"]) and np.array_equal(lit["surf_cnt"], o["surf_cnt"])
        np.testing.assert_allclose(lit["surf"], o["surf"], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("fold,wait", [(0, 1), (1, 0), (0, 0)])
def test_rot_extractor_fallback_paths(gpu_ctx, oracle, fold, wait):
    """Round 6 moved two steps into kernels that wait for other workgroups (the ring stage writes the scan's lists behind a look-back over the lower rings; a segment
    whose pick may lie under its predecessor's marks waits for them).  Both waits are bounded and fall back to the launches of rounds 3-5 — k_rot_compact, the redo in
    k_rot_ring — which options "rot_fold" / "rot_segment_wait" = 0 select outright: the same features, bit for bit."""
    raw = _raw_scan(3125)      # (the 200 k-point scan: several of its segments need the redo)
    q_lb = [0.7071, 0.0, 0.0, 0.7071]
    q_imu = [0.9998, 0.004, -0.007, 0.012]
    o = oracle.extract_rot(raw, q_imu, q_lb, oracle.rot_params(ds_rate=1, atan_mode=2, stable_sort=1))
    gpu_ctx.set_option("rot_fold", fold); gpu_ctx.set_option("rot_segment_wait", wait)
    try:
        ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=1)
        for _ in range(2):
            _compare(ex.extract(raw, q_imu, q_lb, debug=True), o)
    finally:
        gpu_ctx.set_option("rot_fold", 1); gpu_ctx.set_option("rot_segment_wait", 1)


def test_rot_extractor_edge_cases(gpu_ctx, oracle):
    ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=1)
    raw = _raw_scan(391, seed=3)
    # NaNs, near points, points above/below the ring table, an empty scan
    bad = raw.copy()
    bad[10:20, 0] = np.nan
    bad[30:40, :3] *= 0.01
    bad[50:60, 2] = 50.0
    g = ex.extract(bad, debug=True)
    o = oracle.extract_rot(bad, P=oracle.rot_params(ds_rate=1, atan_mode=2, stable_sort=1))
    _compare(g, o)
    g0 = ex.extract(np.zeros((0, 4), np.float32), debug=True)
    assert g0["full"].shape[0] == 0 and g0["edge"].shape[0] == 0 and g0["surf"].shape[0] == 0
    allnan = np.full((100, 4), np.nan, np.float32)
    g1 = ex.extract(allnan, debug=True)
    assert g1["full"].shape[0] == 0
    # 16-ring table
    ex16 = L.RotExtractor(gpu_ctx, n_scans=16, ds_rate=1)
    g16 = ex16.extract(raw, debug=True)
    o16 = oracle.extract_rot(raw, P=oracle.rot_params(n_scans=16, ds_rate=1, atan_mode=2, stable_sort=1))
    _compare(g16, o16)


def test_rot_extractor_voxel_keys_beyond_the_packed_range(gpu_ctx, oracle):
    """The voxel ordering normally comes from rank counting on packed 9 | 11 | 11-bit voxel coordinates (k_rot_scatter / k_rot_segments).  With a
    5 cm leaf the coordinates of a 150 m scene leave that range: the scan is flagged (RotState::vox_overflow) and the radix ordering over
    the ring's bounding box takes over in a second pass — same features, same centroids."""
    raw = _raw_scan(391, seed=5)
    for ds_v in (0.05, 0.6):
        ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=2, ds_v=ds_v)
        g = ex.extract(raw, debug=True)
        o = oracle.extract_rot(raw, P=oracle.rot_params(ds_rate=2, ds_v=ds_v, atan_mode=2, stable_sort=1))
        assert len(o["surf"]) > 200
        _compare(g, o)


def test_rot_extractor_rings_beyond_the_lds_working_set(gpu_ctx, oracle):
    """A 16-ring sensor at 0.04 deg azimuth resolution: ~9000 points per ring — more than the 4096-point LDS working set of k_rot_select
    (such rings used to be refused; the reference takes any ring up to its 400 000-point arrays,
    R/src/Preprocessing.cpp:9-12).  The global-memory pass gives the oracle's indices, labels and centroids bit for bit."""
    sc = synth.OutdoorScene()
    elev = -15.0 + 2.0 * np.arange(16)                       # the reference's 16-ring table: id = int((angle + 15) / 2 + 0.5)
    dirs, ring, rel = synth.spinning_rays(9000, elev)
    t = sc.raycast(np.array([0.0, 0.0, 1.8]), dirs)
    ok = np.isfinite(t)
    t = t + np.random.default_rng(5).normal(0, 0.02, t.shape)
    pts = (dirs * t[:, None])[ok].astype(np.float32)
    raw = np.concatenate([pts, np.full((pts.shape[0], 1), 10.0, np.float32)], 1)
    assert raw.shape[0] > 100_000
    for ds_rate in (1, 4):
        ex = L.RotExtractor(gpu_ctx, n_scans=16, ds_rate=ds_rate)
        g = ex.extract(raw, [0.9999, 0.003, -0.004, 0.002], [1.0, 0, 0, 0], debug=True)
        o = oracle.extract_rot(raw, [0.9999, 0.003, -0.004, 0.002], [1.0, 0, 0, 0], oracle.rot_params(n_scans=16, ds_rate=ds_rate, atan_mode=2, stable_sort=1))
        assert (o["ring_end"] - o["ring_start"]).max() > 4096 + 1000          # really beyond the LDS cap
        assert len(o["edge_idx"]) > 20 and len(o["surf"]) > 1000
        _compare(g, o)


@pytest.mark.parametrize("seed", range(6))
def test_rot_extractor_ragged_scans(gpu_ctx, oracle, seed):
    """The paths of the five-launch chain that a clean scan does not reach: more than a thousand dropped points at either end of the scan
    (the first / last surviving point is found in several trips), rings of every length from none to a few dozen points (fewer than six
    usable points: not selected; fewer than ~75: the reference's serial order through the six segments in k_rot_ring), a scan cut to a
    random stretch (a ring that starts / ends in the middle of a segment), whole rings inside ONE voxel (a single run of hundreds of
    points) and points repeated many times (equal curvatures: ties by index; equal voxel keys: runs)."""
    rng = np.random.default_rng(100 + seed)
    raw = _raw_scan(391, seed=seed)
    n = raw.shape[0]
    kind = seed % 3
    if kind == 0:          # dropped points at both ends, random NaNs and near points inside
        a, b = int(rng.integers(1100, 3000)), int(rng.integers(1100, 2500))
        raw[:a, rng.integers(0, 3)] = np.nan
        raw[n - b:, :3] *= 0.001
        raw[rng.integers(0, n, 500), 0] = np.nan
    elif kind == 1:        # a short stretch of the scan: a few dozen points per ring, some rings empty
        lo = int(rng.integers(0, n // 2))
        raw = raw[lo:lo + int(rng.integers(300, 2500))].copy()
        raw = raw[rng.random(raw.shape[0]) < 0.7]
    else:                  # dense: the scene shrunk into a few voxels, points repeated
        raw[:, :3] *= 0.04
        raw[:, :3] += np.array([3.5, 0.2, 0.1], np.float32)
        rep = rng.integers(0, n, n // 3)
        raw[rep] = raw[(rep // 7) * 7 % n]
    for ds_rate, ds_v in ((1, 0.6), (2, 0.3)):
        ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=ds_rate, ds_v=ds_v)
        g = ex.extract(raw, debug=True)
        o = oracle.extract_rot(raw, P=oracle.rot_params(ds_rate=ds_rate, ds_v=ds_v, atan_mode=2, stable_sort=1))
        assert o["full"].shape[0] > 100
        _compare(g, o)


def test_rot_extractor_page_locked_buffers(gpu_ctx):
    """Page-locked host memory takes its own ways through lili_extract_rot (round 4): the scan is read by the conversion kernel across PCIe instead of being copied
    first, the deskewed cloud leaves through a thin copy kernel on a side stream, the feature lists are written by k_rot_send behind the concatenation and the call
    synchronises once.  Same clouds, bit for bit, as the staged copies pageable memory takes — for every mix of page-locked and pageable arguments, with 32-byte
    PointXYZI rows (only the first 16 bytes of a row are written) and with a capacity below the count."""
    import ctypes as C
    w = synth.make_workload(n_map=50_000, n_az=1100, half_extent=(60.0, 60.0))
    raw = np.ascontiguousarray(np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 10.0, np.float32)], 1))
    ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=4)
    ref = ex.extract(raw)                                                  # pageable in, pageable out
    assert len(ref["edge"]) > 50 and len(ref["surf"]) > 500
    pin_in = L.api.PinnedArray(raw.shape, np.float32)
    pin_in.array[...] = raw
    for src in (pin_in.array, raw):
        got = ex.extract(src, reuse=True)                                   # page-locked out
        for k in ("full", "edge", "surf"):
            assert np.array_equal(got[k], ref[k]), k
    got = ex.extract(pin_in.array)                                          # page-locked in, pageable out
    for k in ("full", "edge", "surf"):
        assert np.array_equal(got[k], ref[k]), k
    n = raw.shape[0]
    qi = np.array([1.0, 0.0, 0.0, 0.0]); ql = np.array([1.0, 0.0, 0.0, 0.0])
    cloud = L.api.Cloud(pin_in.array.ctypes.data, n, 16, 12, L.api.MEM_HOST)
    pins = [L.api.PinnedArray((n, 8), np.float32) for _ in range(3)]       # 32-byte rows
    page = np.zeros((n, 8), np.float32)
    small = 100
    for bufs, caps, strides in (((pins[0].array, pins[1].array, pins[2].array), (n, n, n), (16, 32, 32)),
                                ((pins[0].array, page, pins[2].array), (n, n, n), (16, 32, 32)),
                                ((pins[0].array, pins[1].array, pins[2].array), (n, n, small), (16, 16, 32))):
        for b in bufs:
            b[...] = -7.0
        outs = [L.api.FeatureOut(b.ctypes.data, c, s, L.api.MEM_HOST, 0) for b, c, s in zip(bufs, caps, strides)]
        gpu_ctx._chk(gpu_ctx.lib.lili_extract_rot(gpu_ctx.h, C.byref(cloud), qi.ctypes.data_as(C.c_void_p), ql.ctypes.data_as(C.c_void_p), C.byref(ex.params),
                                                  C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        for b, o_, c, s, k in zip(bufs, outs, caps, strides, ("full", "edge", "surf")):
            assert o_.count == ref[k].shape[0], k
            m = min(o_.count, c)
            rows = b.reshape(-1)[: m * (s // 4)].reshape(m, s // 4) if s == 16 else b[:m]
            assert np.array_equal(rows[:, :4], ref[k][:m]), (k, c, s)
            if s == 32:
                assert np.all(rows[:, 4:] == -7.0), k                        # bytes 16..31 of a row are the caller's
            if m < b.shape[0] and s == 32:
                assert np.all(b[m:] == -7.0), k                              # nothing behind the records that were asked for
    for p in pins:
        p.close()
    pin_in.close()
