"""GPU parity of the Livox Horizon extractor vs the oracle: grid ownership, emitted cells (feature indices) and the
order of both lists exact; point payloads bit-exact; stored normals / directions to 1e-6 (two independent f64
eigen-solvers behind a float store)."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


def _check(g, o):
    assert np.array_equal(g["cut_src"], o["cut_src"])
    assert np.array_equal(g["cutted"].view(np.uint32), o["cutted"].view(np.uint32))
    assert np.array_equal(g["cell_src"], o["cell_src"])
    assert np.array_equal(g["edge_cell"], o["edge_cell"])
    assert np.array_equal(g["surf_cell"], o["surf_cell"])
    for k in ("edge", "surf"):
        a, b = g[k], o[k]
        assert a.shape == b.shape
        assert np.array_equal(a[:, [0, 1, 2, 6, 7]].view(np.uint32), b[:, [0, 1, 2, 6, 7]].view(np.uint32))
        np.testing.assert_allclose(a[:, 3:6], b[:, 3:6], rtol=0, atol=2e-6)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_livox_extractor_parity(gpu_ctx, oracle, seed):
    scan = synth.make_livox_scan(seed)
    ang = 0.02
    q_imu = [np.cos(ang / 2), 0.6 * np.sin(ang / 2), -0.3 * np.sin(ang / 2), 0.74 * np.sin(ang / 2)]
    ex = L.LivoxExtractor(gpu_ctx)
    g = ex.extract(scan, q_imu, debug=True)
    o = oracle.extract_livox(scan, q_imu)
    assert len(o["surf"]) > 5000 and len(o["edge"]) > 5
    _check(g, o)
    # pcl::PointXYZINormal layout carries the same payload
    gp = ex.extract(scan, q_imu, pcl_layout=True)
    assert np.array_equal(gp["surf"][:, [0, 1, 2, 4, 5, 6, 8, 9]], g["surf"])
    assert np.all(gp["surf"][:, 3] == 1.0)


def test_livox_extractor_edge_cases(gpu_ctx, oracle):
    ex = L.LivoxExtractor(gpu_ctx, surf_thres=0.2, edge_thres=2.0)
    P = oracle.livox_params(0.2, 2.0, 0.1)
    scan = synth.make_livox_scan(5)
    bad = scan.copy()
    bad[100:110, 3] = -0.5        # negative line -> dropped before lidar_cloud_cutted
    bad[200:210, 3] = 7.25        # line >= 6: rejected (the reference would write out of bounds)
    bad[300:400, :3] *= 50.0      # beyond 200 m: in cutted, not in the grid
    g = ex.extract(bad, debug=True)
    o = oracle.extract_livox(bad, P=P)
    _check(g, o)
    g0 = ex.extract(np.zeros((0, 5), np.float32), debug=True)
    assert g0["cutted"].shape[0] == 0 and g0["surf"].shape[0] == 0 and (g0["cell_src"] == -1).all()


def test_livox_extractor_layouts_and_repeats(gpu_ctx, oracle):
    """The rows are read as they are (k_livox_prep picks x, y, z, intensity, curvature at their byte offsets): the packed 20-byte rows from host memory,
    the same rows already on the device, and pcl::PointXYZINormal rows (48 bytes, intensity at 32, curvature at 36) on the device give the same
    features.  And the ownership table is re-armed by every scan (no init launch): scans of different sizes and an empty one in between change nothing."""
    import ctypes as C
    import torch
    scan = synth.make_livox_scan(7)
    ex = L.LivoxExtractor(gpu_ctx)
    ref = ex.extract(scan, debug=True)
    o = oracle.extract_livox(scan)
    _check(ref, o)
    for other in (scan[:5000], np.zeros((0, 5), np.float32), scan[::3], scan):
        g = ex.extract(np.ascontiguousarray(other), debug=True)
    for k in ("cutted", "edge", "surf", "cut_src", "cell_src", "edge_cell", "surf_cell"):
        assert np.array_equal(g[k], ref[k]), k
    n = scan.shape[0]
    wide = np.zeros((n, 12), np.float32)
    wide[:, :3] = scan[:, :3]; wide[:, 3] = 1.0; wide[:, 8] = scan[:, 3]; wide[:, 9] = scan[:, 4]
    cap = max(n, 24000)
    for rows, stride, off_i, off_c in ((scan, 20, 12, 16), (wide, 48, 32, 36)):
        d = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
        cloud = L.api.Cloud(d.data_ptr(), n, stride, off_i, L.api.MEM_DEVICE)
        bufs = [np.zeros((cap, 8), np.float32) for _ in range(3)]
        outs = [L.api.FeatureOut(b.ctypes.data, cap, 32, L.api.MEM_HOST, 0) for b in bufs]
        qi = np.array([1.0, 0.0, 0.0, 0.0])
        gpu_ctx._chk(gpu_ctx.lib.lili_extract_livox(gpu_ctx.h, C.byref(cloud), off_c, qi.ctypes.data_as(C.c_void_p), C.byref(ex.params),
                                                    C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
        for b, o_, k in zip(bufs, outs, ("cutted", "edge", "surf")):
            assert o_.count == ref[k].shape[0] and np.array_equal(b[:o_.count], ref[k]), (stride, k)


def test_livox_extractor_page_locked_buffers(gpu_ctx):
    """Page-locked host memory takes its own ways through lili_extract_livox (round 4): the scan is read by k_livox_prep across PCIe instead of being copied first,
    and the three outputs are written by ONE packing launch (k_livox_pack3) straight into the caller's buffers.  Same records, bit for bit, as the staged paths
    pageable memory takes — in both record layouts, for every mix of page-locked and pageable arguments, and with a capacity below the count."""
    import ctypes as C
    scan = synth.make_livox_scan(9)
    ex = L.LivoxExtractor(gpu_ctx)
    for pcl_layout in (False, True):
        ref = ex.extract(scan, pcl_layout=pcl_layout)                 # pageable in, pageable out
        pin_in = L.api.PinnedArray(scan.shape, np.float32)
        pin_in.array[...] = scan
        for src in (pin_in.array, scan):
            got = ex.extract(src, pcl_layout=pcl_layout, reuse=True)  # page-locked out (all three: one packing launch)
            for k in ("cutted", "edge", "surf"):
                assert np.array_equal(got[k], ref[k]), (pcl_layout, k)
        got = ex.extract(pin_in.array, pcl_layout=pcl_layout)         # page-locked in, pageable out
        for k in ("cutted", "edge", "surf"):
            assert np.array_equal(got[k], ref[k]), (pcl_layout, k)
        # one output pageable, two page-locked; and a surf capacity below the count: `count` still reports what was available
        w = 12 if pcl_layout else 8
        n = scan.shape[0]
        pins = [L.api.PinnedArray((24000, w), np.float32) for _ in range(2)]
        page = np.zeros((24000, w), np.float32)
        small = 1000
        cloud = L.api.Cloud(pin_in.array.ctypes.data, n, 20, 12, L.api.MEM_HOST)
        qi = np.array([1.0, 0.0, 0.0, 0.0])
        for bufs, caps in (((pins[0].array, page, pins[1].array), (24000, 24000, 24000)), ((pins[0].array, pins[1].array, page), (24000, 24000, 24000)),
                           ((pins[0].array, page, pins[1].array), (24000, 24000, small))):
            for b in bufs:
                b[...] = -7.0
            outs = [L.api.FeatureOut(b.ctypes.data, c, 4 * w, L.api.MEM_HOST, 0) for b, c in zip(bufs, caps)]
            gpu_ctx._chk(gpu_ctx.lib.lili_extract_livox(gpu_ctx.h, C.byref(cloud), 16, qi.ctypes.data_as(C.c_void_p), C.byref(ex.params),
                                                        C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2])))
            for b, o_, c, k in zip(bufs, outs, caps, ("cutted", "edge", "surf")):
                assert o_.count == ref[k].shape[0], k
                m = min(o_.count, c)
                assert np.array_equal(b[:m], ref[k][:m]), (k, c)
        for p in pins:
            p.close()
        pin_in.close()
