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
