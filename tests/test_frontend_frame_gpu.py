"""lili_frontend_frame: the whole per-scan chain of the reference's odometry node as ONE device-resident C call.

* against the reference's OWN front-end node (tests/golden/ref_frontend.npz: LiLi-OM/src/LidarOdometry.cpp compiled unmodified) with the node's start-up
  (frame 0 stored only, frame 1 matched against its own features with 8 iterations) — no oracle in the loop;
* against the chain of separate C calls with host copies in between (round 4's structure): same poses bit for bit;
* argument errors."""
import importlib.util
import math
import os

import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(G, "make_ref_golden.py"))
M = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(M)


def _q_imu(integ, stamps, imu_t, gyr, k):
    m = imu_t <= stamps[k + 2]
    return integ.integrate(imu_t[m], gyr[m], stamps[k + 1])


def test_frontend_frame_follows_the_reference_node():
    """Raw Livox scans in, poses out, one lili_frontend_frame call per scan: the poses the reference node (LidarOdometry.cpp compiled as is, one GN step per
    ceres::Solve) held after every frame, within the north star's 1e-4 m / 1e-4 rad.  poseInitialization / computeRelative (host scalar code of the node) are the
    harness's (tests/frontend_chain.py)."""
    from tests import frontend_chain as F
    g = np.load(os.path.join(G, "ref_frontend.npz"))
    frames, stamps, imu_t, gyr = M.frontend_inputs()
    ctx = L.Context(0)
    try:
        integ = L.api.ImuIntegrator()
        odo = L.FrontendOdometry(ctx, scan_match_cnt=int(M.FRONTEND_PARAMS["/lidar_odometry/scan_match_cnt"]), first_match_cnt=8, reference_startup=True)
        odo.reset()
        abs_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        rel_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        poses, out_abs, out_rel, matched = [], [], [], []
        for k in range(M.FRONTEND_FRAMES):
            q_imu = _q_imu(integ, stamps, imu_t, gyr, k)
            scan = np.ascontiguousarray(frames[k], np.float32)
            if k == 0:                      # savePoses + checkInitialization
                t, q, info = odo.frame(scan, abs_pose[4:], abs_pose[:4], q_imu)
                assert not info["matched"] and np.array_equal(t, abs_pose[4:]) and np.array_equal(q, abs_pose[:4])
            else:
                q0, t0, dq, dt = abs_pose[:4], abs_pose[4:], rel_pose[:4], rel_pose[4:]
                t0 = F.eigen_qrot(q0, dt[None, :])[0] + t0      # poseInitialization
                q0 = F.eigen_qmul(q0, dq)
                t, q, info = odo.frame(scan, t0, q0, q_imu)
                assert info["gn_status"] == 0
                abs_pose = np.r_[q, t]
            matched.append(info["matched"])
            poses.append(abs_pose.copy())
            if k > 0:                       # computeRelative
                q1, t1 = poses[-2][:4], poses[-2][4:]
                q1i = F.eigen_qinv(q1)
                rel_pose = np.r_[F.eigen_qmul(q1i, abs_pose[:4]), F.eigen_qrot(q1i, (abs_pose[4:] - t1)[None, :])[0]]
            out_abs.append(abs_pose.copy()); out_rel.append(rel_pose.copy())
        a, r = np.array(out_abs), np.array(out_rel)
        ref = g["abs_pose"]
        assert all(matched[1:]), matched
        assert np.abs(a[:, 4:] - ref[:, 4:]).max() < 1e-4, np.abs(a[:, 4:] - ref[:, 4:]).max()
        assert np.abs(a[:, :4] - ref[:, :4]).max() < 5e-5
        assert np.abs(r - g["rel_pose"]).max() < 1e-4
        print("lili_frontend_frame vs reference node: max |dt| = %.3g m, max |dq| = %.3g" % (np.abs(a[:, 4:] - ref[:, 4:]).max(), np.abs(a[:, :4] - ref[:, :4]).max()))
    finally:
        ctx.close()


def test_frontend_frame_rot_follows_the_reference_rot_node():
    """Round 6: raw 64-ring scans in, poses out, ONE lili_frontend_frame_rot call per scan — the poses LiLi-OM-ROT's own odometry node
    (R/src/LidarOdometry.cpp compiled as is behind R/src/Preprocessing.cpp, tests/golden/ref_frontend_R.npz) held after every frame, within 1e-4 m / 1e-4 rad; feature
    counts equal the reference node's."""
    from tests import frontend_chain as F
    g = np.load(os.path.join(G, "ref_frontend_R.npz"))
    scans, stamps, imu_t, gyr = M.frontend_rot_inputs()
    ctx = L.Context(0)
    try:
        integ = L.api.ImuIntegrator()
        odo = L.RotFrontendOdometry(ctx, n_scans=64, ds_rate=4, q_lb=M.ROT_QLB, scan_match_cnt=int(M.FRONTEND_R_PARAMS["/lidar_odometry/scan_match_cnt"]), first_match_cnt=8,
                                    reference_startup=True)
        odo.reset()
        abs_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        rel_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        poses, out_abs, out_rel = [], [], []
        for k in range(M.FRONTEND_R_FRAMES):
            q_imu = _q_imu(integ, stamps, imu_t, gyr, k)
            scan = np.ascontiguousarray(scans[k], np.float32)
            if k == 0:
                t, q, info = odo.frame(scan, abs_pose[4:], abs_pose[:4], q_imu)
                assert not info["matched"]
            else:
                q0, t0, dq, dt = abs_pose[:4], abs_pose[4:], rel_pose[:4], rel_pose[4:]
                t0 = F.eigen_qrot(q0, dt[None, :])[0] + t0      # poseInitialization
                q0 = F.eigen_qmul(q0, dq)
                t, q, info = odo.frame(scan, t0, q0, q_imu)
                assert info["gn_status"] == 0 and info["matched"]
                abs_pose = np.r_[q, t]
            assert (info["n_edge"], info["n_surf"]) == (int(g["n_edge"][k]), int(g["n_surf"][k])), (k, info)
            poses.append(abs_pose.copy())
            if k > 0:                       # computeRelative
                q1, t1 = poses[-2][:4], poses[-2][4:]
                q1i = F.eigen_qinv(q1)
                rel_pose = np.r_[F.eigen_qmul(q1i, abs_pose[:4]), F.eigen_qrot(q1i, (abs_pose[4:] - t1)[None, :])[0]]
            out_abs.append(abs_pose.copy()); out_rel.append(rel_pose.copy())
        a, r = np.array(out_abs), np.array(out_rel)
        ref = g["abs_pose"]
        assert np.abs(a[:, 4:] - ref[:, 4:]).max() < 1e-4, np.abs(a[:, 4:] - ref[:, 4:]).max()
        assert np.abs(a[:, :4] - ref[:, :4]).max() < 5e-5
        assert np.abs(r - g["rel_pose"]).max() < 1e-4
        assert np.linalg.norm(a[-1][4:] - a[1][4:]) > 1.5
        print("lili_frontend_frame_rot vs reference ROT node: max |dt| = %.3g m, max |dq| = %.3g" % (np.abs(a[:, 4:] - ref[:, 4:]).max(), np.abs(a[:, :4] - ref[:, :4]).max()))
    finally:
        ctx.close()


def test_frontend_frame_rot_with_the_callers_maps_equals_the_separate_calls():
    """LILI_FRAME_EXTERNAL_MAP | LILI_FRAME_EDGES, leaf_query = 0 (BASELINE configs[0] as ONE call): extraction, both feature kinds as queries, one outer iteration against the
    caller's surf + edge indices — the pose equals lili_extract_rot -> lili_s2m_set_queries x 2 -> lili_s2m_pose_set -> lili_s2m_iterate bit for bit."""
    import torch
    w = synth.make_workload(n_map=200_000, n_az=400, half_extent=(150.0, 150.0), verbose=False)
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 50.0, np.float32)], 1)
    P = L.make_params("rot")
    q_lb = np.array(list(P.q_lb))
    ctx = L.Context(0)
    try:
        m = L.ScanToMapMatcher(ctx, P)
        m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
        m.set_input_cloud(L.KIND_EDGE, w["edge_map_xyz"])
        tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
        t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.2, 1.0)
        d_raw = torch.from_numpy(raw).cuda()
        ex = L.RotExtractor(ctx, n_scans=64, ds_rate=4)
        ex.extract_device(d_raw.data_ptr(), raw.shape[0], (1.0, 0, 0, 0), q_lb)
        _, d_edge, d_surf = L.api.extract_rot_device(ctx)
        m.set_queries(0, L.KIND_SURF, d_surf); m.set_queries(0, L.KIND_EDGE, d_edge)
        m.pose_set(0, t0, q0)
        m.iterate(0, 2, L.MASK_SURF | L.MASK_EDGE)
        t_sep, q_sep, st = m.pose_get(0)
        assert st == 0
        odo = L.RotFrontendOdometry(ctx, params=P, n_scans=64, ds_rate=4, q_lb=q_lb, leaf_query=0.0, scan_match_cnt=2, external_map=True, edges=True, slot=1)
        cloud = L.api.cloud_from_device(d_raw.data_ptr(), raw.shape[0], 16, 12)
        t_one, q_one, info = odo.frame(cloud, t0, q0)
        assert info["matched"] and info["gn_status"] == 0 and info["n_surf"] == int(d_surf.n) and info["n_edge"] == int(d_edge.n) and info["n_query"] == int(d_surf.n)
        assert np.array_equal(t_one, t_sep) and np.array_equal(q_one * np.sign(q_one[0]), q_sep * np.sign(q_sep[0]))
        assert np.linalg.norm(t_one - t0) > 0.01
    finally:
        ctx.close()


def test_frontend_frame_rot_guessed_feature_counts_change_nothing():
    """Round 6: with the caller's maps and the features as queries the matcher is enqueued behind the extractor for GUESSED feature counts (the previous scan's + 1/8 + 64,
    padding rows are NaN) and the call synchronises once.  The pose must be the one of the exactly sized slot — on a hit (same scan again), on a miss (a scan with many
    more features than the one before: matched again the plain way) and with the option off."""
    import torch
    w = synth.make_workload(n_map=200_000, n_az=400, half_extent=(150.0, 150.0), verbose=False)
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 50.0, np.float32)], 1)
    n_az = raw.shape[0] // 64
    thin = np.ascontiguousarray(raw.reshape(n_az, 64, 4)[:, ::4].reshape(-1, 4))      # a quarter of the rings: about a quarter of the features
    P = L.make_params("rot")
    q_lb = np.array(list(P.q_lb))
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.2, 1.0)
    d_raw, d_thin = torch.from_numpy(raw).cuda(), torch.from_numpy(thin).cuda()
    cloud = L.api.cloud_from_device(d_raw.data_ptr(), raw.shape[0], 16, 12)
    cloud_thin = L.api.cloud_from_device(d_thin.data_ptr(), thin.shape[0], 16, 12)

    def run(guess, seq):
        ctx = L.Context(0)
        try:
            ctx.set_option("frame_guess_counts", guess)
            m = L.ScanToMapMatcher(ctx, P)
            m.set_input_cloud(L.KIND_SURF, w["map_xyz"]); m.set_input_cloud(L.KIND_EDGE, w["edge_map_xyz"])
            odo = L.RotFrontendOdometry(ctx, params=P, n_scans=64, ds_rate=1, q_lb=q_lb, leaf_query=0.0, scan_match_cnt=2, external_map=True, edges=True, slot=1)
            out = []
            for c in seq:
                t, q, info = odo.frame(c, t0, q0)
                assert info["matched"] and info["gn_status"] == 0
                out.append((t.copy(), q.copy(), info["n_surf"], info["n_edge"], info["n_query"]))
            # an EMPTY scan behind scans with features (a guess exists): nothing to match, the predicted pose comes back
            t, q, info = odo.frame(L.api.cloud_from_device(d_raw.data_ptr(), 0, 16, 12), t0, q0)
            assert not info["matched"] and info["n_surf"] == 0 and np.array_equal(t, t0) and np.array_equal(q, q0)
            return out
        finally:
            ctx.close()

    plain = run(0, [cloud, cloud_thin])
    full, small = plain
    assert small[2] < 0.6 * full[2] and full[2] > 1000              # the thin scan has far fewer features: the full one behind it overflows the guess
    got = run(1, [cloud, cloud, cloud_thin, cloud, cloud_thin, cloud_thin])      # first (no guess) | hit | hit (fewer than guessed) | MISS | hit | hit
    for g, want in zip(got, [full, full, small, full, small, small]):
        assert g[2:] == want[2:]
        assert np.array_equal(g[0], want[0]) and np.array_equal(g[1], want[1])


@pytest.mark.parametrize("edges", [False, True])
def test_frontend_frame_rot_guess_behind_a_second_pass_of_the_extractor(edges):
    """The guessed-count frame when the extractor REWRITES its lists after the matcher has been enqueued: with a 5 cm VoxelGrid leaf the voxel keys of a 150 m scene leave
    the packed range, lili_extract_rot_complete runs the radix ordering pass and the ring stage again — the frame must notice (the lists its matcher read are stale), match
    again the plain way and return the pose of the plain chain; also without edge queries (no edge sink)."""
    import torch
    w = synth.make_workload(n_map=200_000, n_az=400, half_extent=(150.0, 150.0), verbose=False)
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 50.0, np.float32)], 1)
    P = L.make_params("rot")
    q_lb = np.array(list(P.q_lb))
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.2, 1.0)
    d_raw = torch.from_numpy(raw).cuda()
    cloud = L.api.cloud_from_device(d_raw.data_ptr(), raw.shape[0], 16, 12)

    def run(guess):
        ctx = L.Context(0)
        try:
            ctx.set_option("frame_guess_counts", guess)
            m = L.ScanToMapMatcher(ctx, P)
            m.set_input_cloud(L.KIND_SURF, w["map_xyz"]); m.set_input_cloud(L.KIND_EDGE, w["edge_map_xyz"])
            odo = L.RotFrontendOdometry(ctx, params=P, n_scans=64, ds_rate=2, ds_v=0.05, q_lb=q_lb, leaf_query=0.0, scan_match_cnt=2, external_map=True, edges=edges, slot=1)
            out = []
            for _ in range(3):
                t, q, info = odo.frame(cloud, t0, q0)
                assert info["matched"] and info["gn_status"] == 0 and info["n_surf"] > 1000
                out.append((t.copy(), q.copy(), info["n_surf"], info["n_edge"]))
            return out
        finally:
            ctx.close()

    plain, guessed = run(0), run(1)
    for a, b in zip(plain, guessed):
        assert a[2:] == b[2:] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _circuit(f, radius=4.0, step=0.03):
    a = step * f
    yaw = a + math.pi / 2
    return np.array([radius * math.cos(a), radius * math.sin(a), 1.8]), np.array([math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)]), yaw


def _predict(poses):
    if len(poses) == 1:
        return poses[-1]
    (ta, qa), (tb, qb) = poses[-2], poses[-1]
    qi = qa * np.array([1, -1, -1, -1]) / np.dot(qa, qa)
    dq = synth.quat_mul(qi, qb)
    q0 = synth.quat_mul(qb, dq)
    return tb + synth.quat_rot(qb, synth.quat_rot(qi, tb - ta)), q0 / np.linalg.norm(q0)


@pytest.mark.parametrize("pinned", [False, True])
def test_frontend_frame_equals_the_chain_of_separate_calls(pinned):
    """The fused call moves no arithmetic: extraction, VoxelGrid, ring, index and iterations are the library's own stages, so its poses equal those of
    lili_extract_livox -> lili_voxel_filter -> lili_localmap_commit -> lili_s2m_set_queries / _pose_set / _iterate / _pose_get -> lili_localmap_push driven
    through host buffers BIT FOR BIT (the replay tools' start-up: frame 0 enters the ring, 12 iterations on frame 1)."""
    n_frames = 9
    frames = [synth.make_livox_scan(100 + f, origin=_circuit(f)[0], yaw=_circuit(f)[2], inject_bad=(f == 3)) for f in range(n_frames)]
    ctx = L.Context(0)
    pins = []
    try:
        P = L.make_params("frontend")
        # --- separate calls
        ex = L.LivoxExtractor(ctx)
        m = L.ScanToMapMatcher(ctx, P)
        local = L.api.LocalMap(ctx, L.KIND_SURF, 5, 0.4, P.kd_max_radius)
        staged, nq_staged = [], []
        for f in range(n_frames):
            feats = ex.extract(frames[f])
            qry, _ = L.api.voxel_filter(ctx, np.ascontiguousarray(feats["surf"][:, [0, 1, 2, 7]]), 0.4)
            if f == 0:
                t, q = _circuit(0)[:2]
            else:
                t0, q0 = _predict(staged)
                local.commit()
                m.set_queries(0, L.KIND_SURF, qry)
                m.pose_set(0, t0, q0)
                m.iterate(0, 12 if f == 1 else 6, L.MASK_SURF)
                t, q, st = m.pose_get(0)
                assert st == 0
                if q[0] < 0:
                    q = -q
            staged.append((np.asarray(t, np.float64), np.asarray(q, np.float64)))
            nq_staged.append(qry.shape[0])
            local.push(qry, t, q)
        # --- one call per frame (a fresh ring; ring width 5 so that keyframes are popped inside the sequence)
        odo = L.FrontendOdometry(ctx, P, width=5, scan_match_cnt=6, first_match_cnt=12, reference_startup=False)
        odo.reset()
        fused = []
        for f in range(n_frames):
            scan = frames[f]
            if pinned:
                pa = L.api.PinnedArray(scan.shape, np.float32)
                pa.array[...] = scan
                pins.append(pa)
                scan = pa.array
            t0, q0 = _circuit(0)[:2] if f == 0 else _predict(fused)
            t, q, info = odo.frame(scan, t0, q0, timing=True)
            assert info["gn_status"] == 0 and info["matched"] == (f > 0) and info["n_query"] == nq_staged[f]
            assert len(info["stage_us"]) == 4 and all(b >= a for a, b in zip(info["stage_us"], info["stage_us"][1:]))
            fused.append((t, q))
        for f, (a, b) in enumerate(zip(fused, staged)):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (f, a, b)
        err = max(float(np.linalg.norm(p[0] - _circuit(f)[0])) for f, p in enumerate(fused))
        assert err < 0.15, err
    finally:
        for p in pins:
            p.close()
        ctx.close()


def test_frontend_frame_survives_small_featureless_and_stretched_frames():
    """The frame call's deferred stages on inputs that leave the steady state: a SMALL frame (its surf features go through the one-workgroup filter, enqueued behind the
    index build like the general chain), a frame WITHOUT features (no queries: not matched, the pose is the prediction, an empty keyframe joins the ring), a frame
    with three times the ranges (another bounding box and voxel count than the guess was made from; the range gate keeps a Livox frame within the guessed 24 key bits,
    so the guess holds — the miss itself is tests/test_voxel_gpu.py's).  Wherever the chain of separate calls is defined the fused call equals it bit for bit; the
    sequence goes on afterwards."""
    def scan(f, **kw):
        return synth.make_livox_scan(100 + f, origin=_circuit(f)[0], yaw=_circuit(f)[2], inject_bad=False, **kw)
    frames = [scan(f) for f in range(9)]
    frames[3] = np.ascontiguousarray(frames[3][:6000])                     # a quarter of the sweep: a few thousand features at most
    frames[5] = np.ascontiguousarray(frames[5][:40])                       # 40 points of one line: no block reaches 25 valid cells
    wide = frames[7].copy(); wide[:, :3] *= 3.0; frames[7] = wide          # same directions, three times the ranges
    ctx = L.Context(0)
    try:
        P = L.make_params("frontend")
        ex = L.LivoxExtractor(ctx)
        m = L.ScanToMapMatcher(ctx, P)
        local = L.api.LocalMap(ctx, L.KIND_SURF, 5, 0.4, P.kd_max_radius)
        staged, nq_staged = [], []
        for f in range(len(frames)):
            feats = ex.extract(frames[f])
            surf = np.ascontiguousarray(feats["surf"][:, [0, 1, 2, 7]])
            qry = L.api.voxel_filter(ctx, surf, 0.4)[0] if surf.shape[0] else np.zeros((0, 4), np.float32)
            if f == 0:
                t, q = _circuit(0)[:2]
            else:
                t0, q0 = _predict(staged)
                local.commit()
                if qry.shape[0]:
                    m.set_queries(0, L.KIND_SURF, qry)
                    m.pose_set(0, t0, q0)
                    m.iterate(0, 12 if f == 1 else 6, L.MASK_SURF)
                    t, q, st = m.pose_get(0)
                    if q[0] < 0:
                        q = -q
                else:
                    t, q = t0, q0                                           # (updateTransformationWithCeres returns before it touches the pose)
            staged.append((np.asarray(t, np.float64), np.asarray(q, np.float64)))
            nq_staged.append(qry.shape[0])
            local.push(qry, t, q)
        assert nq_staged[5] == 0 and 0 < nq_staged[3] < nq_staged[2]
        guesses0 = L.api.voxel_filter_stats(ctx)[0]
        odo = L.FrontendOdometry(ctx, P, width=5, scan_match_cnt=6, first_match_cnt=12, reference_startup=False)
        odo.reset()
        fused = []
        for f in range(len(frames)):
            t0, q0 = _circuit(0)[:2] if f == 0 else _predict(fused)
            t, q, info = odo.frame(frames[f], t0, q0)
            assert info["n_query"] == nq_staged[f] and info["matched"] == (f > 0 and nq_staged[f] > 0), (f, info)
            fused.append((t, q))
        assert L.api.voxel_filter_stats(ctx)[0] > guesses0          # the filters of the large frames kept their boxes on the device
        for f, (a, b) in enumerate(zip(fused, staged)):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (f, a, b)
    finally:
        ctx.close()


def test_frontend_frame_rejects_bad_arguments(gpu_ctx):
    odo = L.FrontendOdometry(gpu_ctx)
    scan = synth.make_livox_scan(5, inject_bad=False)
    odo.opt.width = 0
    with pytest.raises(L.LiliError):
        odo.frame(scan, np.zeros(3), np.array([1.0, 0, 0, 0]))
    odo.opt.width = 20
    odo.opt.leaf_query = 0.0
    with pytest.raises(L.LiliError):
        odo.frame(scan, np.zeros(3), np.array([1.0, 0, 0, 0]))


def write_frames_bin(path, frames, first_pose, reference_startup):
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", len(frames), 1 if reference_startup else 0))
        for fr in frames:
            fr = np.ascontiguousarray(fr, "<f4")
            f.write(struct.pack("<i", fr.shape[0]))
            f.write(np.asarray(first_pose[0], "<f8").tobytes()); f.write(np.asarray(first_pose[1], "<f8").tobytes())
            f.write(fr.tobytes())


def test_cpp_host_runs_the_frame_loop(tmp_path):
    """examples/frontend_demo.cpp — the loop a merged Preprocessing + LidarOdometry nodelet would run, plain C++ on the C ABI — ends every frame at the pose the
    Python binding's loop reaches with the same prediction arithmetic (Eigen's operation order on both sides): bit for bit."""
    import subprocess
    from tests import frontend_chain as F
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    demo = os.path.join(root, "examples", "frontend_demo")
    if not os.path.exists(demo):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
    n_frames = 6
    frames = [synth.make_livox_scan(100 + f, origin=_circuit(f)[0], yaw=_circuit(f)[2], inject_bad=False) for f in range(n_frames)]
    t_first, q_first = np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0])      # the reference node's world frame IS the first scan's frame (abs_pose starts at identity): the
    path = tmp_path / "frames.bin"                                       # start-up's self-map (frame 1 against its own features) presumes it
    write_frames_bin(path, frames, (t_first, q_first), reference_startup=True)
    out = subprocess.run([demo, str(path), "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = [l.split() for l in out.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == n_frames and "ms_per_frame" in out.stdout
    ctx = L.Context(0)
    try:
        odo = L.FrontendOdometry(ctx, reference_startup=True)
        odo.reset()
        abs_pose = np.r_[q_first, t_first]
        rel_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
        prev = None
        for k in range(n_frames):
            if k == 0:
                t, q, info = odo.frame(frames[k], abs_pose[4:], abs_pose[:4])
            else:
                t0 = F.eigen_qrot(abs_pose[:4], rel_pose[4:][None, :])[0] + abs_pose[4:]
                q0 = F.eigen_qmul(abs_pose[:4], rel_pose[:4])
                t, q, info = odo.frame(frames[k], t0, q0)
            prev, abs_pose = abs_pose, np.r_[q, t]
            if k > 0:
                q1i = F.eigen_qinv(prev[:4])
                rel_pose = np.r_[F.eigen_qmul(q1i, abs_pose[:4]), F.eigen_qrot(q1i, (abs_pose[4:] - prev[4:])[None, :])[0]]
            tok = lines[k]
            pose_c = np.array([float(v) for v in tok[3:10]])
            assert np.array_equal(pose_c, np.r_[t, q]), (k, pose_c, t, q)
            assert (int(tok[11]), int(tok[13]), int(tok[17])) == (info["gn_status"], int(info["matched"]), info["n_query"]) and info["gn_status"] == 0, (k, tok[10:], info)
    finally:
        ctx.close()
