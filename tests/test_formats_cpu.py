"""Host-side callers either side of the hot path — no GPU needed:
gyro integration (C-ABI host function vs the pure-Python restatement, bit-exact), the CustomMsg restatement's
known answers, and the marginalisation assembly from a Gram record vs the reference's per-residual accumulation."""
import numpy as np

import lili_om_amd as L
from lili_om_amd import synth


def _imu_stream(seed, n=400, rate=200.0, t0=1000.0):
    rng = np.random.default_rng(seed)
    stamps = t0 + np.arange(n) / rate + rng.normal(0, 2e-4, n)
    stamps.sort()
    gyr = rng.normal(0, 0.3, (n, 3)) + np.array([0.05, -0.02, 0.4])
    return stamps, gyr


def test_imu_integration_matches_restatement_bit_exact(oracle):
    stamps, gyr = _imu_stream(1)
    g, o = L.api.ImuIntegrator(), oracle.ImuIntegrator()
    # scans every 0.1 s; the IMU buffer grows between scans like imu_buf does
    for k in range(1, 16):
        t_next = 1000.0 + 0.1 * k + 0.013
        n_avail = min(len(stamps), int(np.searchsorted(stamps, t_next + 0.02)) + 1)
        qg = g.integrate(stamps[:n_avail], gyr[:n_avail], t_next)
        qo = o.integrate(stamps[:n_avail], gyr[:n_avail], t_next)
        assert np.array_equal(qg.view(np.uint64), qo.view(np.uint64))
        assert g.state.idx == o.idx and g.state.t_cur == o.t_cur
        assert abs(qg[0] - 1.0) < 1e-2 and 1e-4 < np.linalg.norm(qg[1:]) < 0.1     # ~0.04 rad per scan about z
    # known answer: constant rate w about z for 0.1 s -> un-normalised small-angle products, first order = [1, 0, 0, w*0.05]
    st = 5.0 + np.arange(40) * 0.005
    w = np.tile([0.0, 0.0, 0.2], (40, 1))
    q = L.api.ImuIntegrator().integrate(st, w, 5.1)
    assert abs(q[3] - 0.2 * 0.1 / 2) < 2e-5 and abs(q[1]) == 0 and abs(q[2]) == 0


def test_imu_edge_cases(oracle):
    g, o = L.api.ImuIntegrator(), oracle.ImuIntegrator()
    assert np.array_equal(g.integrate([], np.zeros((0, 3)), 1.0), [1, 0, 0, 0])          # no IMU yet: identity
    stamps, gyr = _imu_stream(2, n=30)
    # every sample older than the scan end: the loop runs off the buffer (L:151-153), then idx >= size next time (L:138-139)
    for t_next in (stamps[-1] + 0.05, stamps[-1] + 0.15):
        qg, qo = g.integrate(stamps, gyr, t_next), o.integrate(stamps, gyr, t_next)
        assert np.array_equal(qg.view(np.uint64), qo.view(np.uint64))
        assert g.state.idx == o.idx
    # NaN rate -> q_iMU reset to identity (L:232-234)
    bad = gyr.copy(); bad[3, 1] = np.nan
    assert np.array_equal(L.api.ImuIntegrator().integrate(stamps, bad, stamps[10]), [1, 0, 0, 0])
    assert np.array_equal(oracle.ImuIntegrator().integrate(stamps, bad, stamps[10]), [1, 0, 0, 0])


def test_custom_msg_restatement_known_answers(oracle):
    pts = np.zeros(3, oracle.CUSTOM_POINT)
    pts["offset_time"] = [0, 500, 1000]
    pts["x"], pts["y"], pts["z"] = [1.5, 2.5, 3.5], [-1, -2, -3], [0.25, 0.5, 0.75]
    pts["reflectivity"], pts["line"] = [0, 100, 255], [0, 3, 5]
    c = oracle.livox_custom_to_cloud(pts)
    assert c.shape == (3, 12) and np.array_equal(c[:, 0], pts["x"]) and np.all(c[:, 3] == 1.0) and np.all(c[:, 4:8] == 0)
    assert c[1, 8] == np.float32(3.0 + float(np.float32(0.5)) * 0.1) and c[2, 8] == np.float32(5.0 + 1.0 * 0.1) and c[0, 8] == 0.0
    assert c[1, 9] == np.float32(0.1 * 100) and c[2, 9] == np.float32(0.1 * 255)
    assert oracle.CUSTOM_POINT.itemsize == 19


def test_marginalisation_assembly_from_gram(oracle):
    """A, b of MarginalizationInfo built from ONE Gram record per keyframe equal the reference's per-residual
    accumulation (ThreadsConstructA) over the same robustified rows, for a 3-keyframe window layout."""
    room = synth.make_room(seed=21, n_query=4000, n_edge_query=400)
    P = L.make_params("livox")
    PO = oracle.params("livox")
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(9), 0.05, 0.3)
    Q2, T2 = L.api.assoc_transform(t0, q0, P)
    tree, etree = oracle.KdTree(room["map_xyz"]), oracle.KdTree(room["edge_map_xyz"])
    rs = oracle.associate_surf(tree, room["map_refl"], room["q_xyz"], room["q_refl"], Q2, T2, PO)
    re_ = oracle.associate_edge(etree, room["eq_xyz"], Q2, T2, PO)
    assert rs["count"] > 500 and re_["count"] > 20
    pos, idx_t, idx_q = 45, 15, 18            # pose blocks of the 2nd keyframe in a 3 x 15 window
    A_ref, b_ref = np.zeros((pos, pos)), np.zeros(pos)
    G = np.zeros((8, 8))
    for kind, rec, lin in (("surf", rs, oracle.linearize_surf), ("edge", re_, oracle.linearize_edge)):
        rows = oracle.linearize_rows(rec, t0, q0, PO, 1.0, kind)
        assert rows.shape[0] == rec["count"]
        oracle.marg_accumulate(rows[:, :7], rows[:, 7], pos, idx_t, idx_q, A_ref, b_ref)
        G += lin(rec, t0, q0, PO, 1.0)[0]
    A, b = np.zeros((pos, pos)), np.zeros(pos)
    L.api.marg_add_lidar(G, A, b, idx_t, idx_q)
    np.testing.assert_allclose(A, A_ref, rtol=1e-12, atol=1e-12 * np.abs(A_ref).max())
    np.testing.assert_allclose(b, b_ref, rtol=1e-12, atol=1e-12 * np.abs(b_ref).max())
    assert np.array_equal(A, A.T) and np.count_nonzero(A) == 36 and np.count_nonzero(b) == 6
    # bad indices are refused
    import pytest
    with pytest.raises(L.LiliError):
        L.api.marg_add_lidar(G, A, b, 43, 18)
