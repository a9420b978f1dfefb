"""HIP path vs the REFERENCE'S OWN outputs (tests/golden/ref_*.npz, produced by oracle/_ref from the reference's unmodified
Preprocessing.cpp / LidarKeyframeFactor.h — see tests/golden/make_ref_golden.py).  No oracle in the loop.

Bars: Livox — every published point bit-exact in its payload (x, y, z, intensity, curvature) and in list order, stored
normals within 2e-6 up to Eigen's arbitrary sign.  ROT — the deskewed cloud and the corner features bit for bit and the corner
INDICES equal to the ones the reference pushed (the extractor runs glibc's float atan / atan2 statement for statement); the
voxel-filtered surf centroids equal up to PCL's unspecified in-voxel summation order (last bit, voxels of >= 3 points).
Factors — the analytic residual/Jacobian rows behind lili_s2m_linearize vs Create()->Evaluate() of the three functors."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest

import lili_om_amd as L

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(G, "make_ref_golden.py"))
M = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(M)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _q_imu(integ, stamps, imu_t, gyr, k):
    m = imu_t <= stamps[k + 2]
    return integ.integrate(imu_t[m], gyr[m], stamps[k + 1])


def _surf_voxel_counts(full, lessflat_idx, leaf=0.6, n_rings=64):
    """Points per output row of the per-ring pcl::VoxelGrid(leaf) over the less-flat points (R/src/Preprocessing.cpp:502-506): voxel indices in
    PCL's own f32 arithmetic (floor(p * inverse_leaf), box offsets), rows in ascending voxel id ring after ring — the order of /surf_features."""
    inv = np.float32(1.0) / np.float32(leaf)
    ring = full[lessflat_idx, 3].astype(np.int32)
    out = []
    for r in range(n_rings):
        p = full[lessflat_idx[ring == r]]
        if p.shape[0] == 0:
            continue
        ijk = np.floor(p[:, :3].astype(np.float32) * inv).astype(np.int64)
        mn = ijk.min(0)
        d = ijk.max(0) - mn + 1
        idx = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * d[0] + (ijk[:, 2] - mn[2]) * d[0] * d[1]
        out.append(np.unique(idx, return_counts=True)[1])
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def test_gpu_rot_extractor_vs_reference(gpu_ctx):
    g = np.load(os.path.join(G, "ref_rot.npz"))
    scans, stamps, imu_t, gyr = M.rot_inputs()
    integ = L.api.ImuIntegrator()
    ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=4)
    for k in range(int(g["n_processed"])):
        out = ex.extract(scans[k], _q_imu(integ, stamps, imu_t, gyr, k), M.ROT_QLB, debug=True)
        for mine, ref in ((out["full"], g[f"cutted{k}"]), (out["edge"], g[f"edge{k}"]), (out["surf"], g[f"surf{k}"])):
            assert mine.shape == ref.shape                                   # same features, same order
            np.testing.assert_allclose(mine, ref, rtol=2e-6, atol=2e-5)
        # the ring / relTime arithmetic follows glibc's float atan / atan2 statement for statement: the deskewed cloud and the
        # corner features are the reference build's bits, and the feature INDICES are the ones the reference pushed
        assert np.array_equal(_bits(out["full"]), _bits(g[f"cutted{k}"]))
        assert np.array_equal(_bits(out["edge"]), _bits(g[f"edge{k}"]))
        assert np.array_equal(out["edge_idx"], g[f"edge_src{k}"])
        # /surf_features are VoxelGrid centroids: PCL sums the points of a voxel in std::sort's (unspecified) order, the GPU in index
        # order.  A sum of ONE or TWO f32 values does not depend on the order, so every row that differs must belong to a voxel of >= 3
        # points — and differ by rounding only (the former `> 0.9` said neither).
        same = (_bits(out["surf"]) == _bits(g[f"surf{k}"])).all(1)
        counts = _surf_voxel_counts(out["full"], out["lessflat_idx"])
        assert counts.shape[0] == out["surf"].shape[0] and (counts >= 1).all()
        assert same.mean() > 0.95 and (counts[~same] >= 3).all(), (same.mean(), counts[~same].min() if (~same).any() else None)
        ulp = np.abs(_bits(out["surf"])[~same].astype(np.int64) - _bits(g[f"surf{k}"])[~same].astype(np.int64))
        assert ulp.max(initial=0) <= 4, ulp.max(initial=0)


def test_gpu_livox_extractor_vs_reference(gpu_ctx):
    g = np.load(os.path.join(G, "ref_livox.npz"))
    scans, stamps, imu_t, gyr = M.livox_inputs()
    integ = L.api.ImuIntegrator()
    ex = L.LivoxExtractor(gpu_ctx)
    pay = [0, 1, 2, 6, 7]
    for k in range(int(g["n_processed"])):
        out = ex.extract(scans[k], _q_imu(integ, stamps, imu_t, gyr, k), debug=True)
        for name in ("cutted", "surf"):
            a = out[name]
            assert a.shape[0] == int(g[f"{name}{k}_n"])
            assert _sha(a[:, pay]) == str(g[f"{name}{k}_sha_payload"]), name          # every point, bit-exact, in order
            np.testing.assert_allclose(np.abs(a[::8, 3:6]), np.abs(g[f"{name}{k}_every8"][:, 3:6]), rtol=0, atol=2e-6)
        e, ge = out["edge"], g[f"edge{k}"]
        assert e.shape == ge.shape
        assert np.array_equal(_bits(e[:, pay]), _bits(ge[:, pay]))
        np.testing.assert_allclose(np.abs(e[:, 3:6]), np.abs(ge[:, 3:6]), rtol=0, atol=2e-6)


def _cauchy_rows(rows):
    """ResidualBlockInfo::Evaluate's corrector for CauchyLoss(1.0) (L/src/MarginalizationFactor.cpp:44-70): rho'' < 0 always,
    so residual and Jacobian are both scaled by sqrt(rho') = 1 / sqrt(1 + r^2).  rows: (n, 8) = [J(7), r]."""
    return rows / np.sqrt(1.0 + rows[:, 7:8] ** 2)


@pytest.mark.parametrize("variant", ["livox", "rot"])
def test_gpu_linearize_vs_reference_functors(gpu_ctx, variant):
    """lili_s2m_linearize (analytic Jacobians, MFMA Gram) vs the reference's OWN functors: the records the GPU association
    produced are evaluated one by one through LidarPlaneNormFactor / LidarEdgeFactor ::Create()->Evaluate() of
    oracle/_ref/libref_factors.so (prebuilt from /root/reference's header; it travels with the snapshot), robustified and
    summed in numpy.  No oracle restatement in the loop."""
    from oracle import ref as R
    from lili_om_amd import synth
    # The functor rows of the records THIS kernel produces are committed (tests/golden/ref_functor_rows.npz, made by tools/dump_functor_inputs.py on
    # the GPU + tests/golden/make_ref_golden.py::functor_rows here): the check runs even where the git-ignored oracle/_ref did not travel
    # (VERDICT r2 #7).  Where it did, the functors are evaluated live as before.
    fix = np.load(os.path.join(G, "ref_functor_rows.npz")) if os.path.exists(os.path.join(G, "ref_functor_rows.npz")) else None
    if not R.available() and fix is None:
        pytest.skip("neither oracle/_ref nor tests/golden/ref_functor_rows.npz")
    room = synth.make_room(seed=12, n_query=1500, n_edge_query=200)
    P = L.make_params(variant)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, np.c_[room["map_xyz"], room["map_refl"]] if variant == "livox" else room["map_xyz"])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_SURF, np.c_[room["q_xyz"], room["q_refl"]] if variant == "livox" else room["q_xyz"])
    m.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(3), 0.05, 0.5)
    Q2, T2 = L.api.assoc_transform(t0, q0, P)
    ns = m.find_corresponding_surf_features(0, Q2, T2)
    ne = m.find_corresponding_corner_features(0, Q2, T2)
    assert ns > 500 and ne > 20
    rs, re_ = m.surf_records(0, ns), m.edge_records(0, ne)
    qlb, tlb = np.array(list(P.q_lb)), np.array(list(P.t_lb))
    ss = P.scale_surf_num / ns if P.scale_surf_num else 1.0
    se = P.scale_edge_num / ne if P.scale_edge_num else 1.0
    live = R.available()
    if live:
        rows_s, rows_e = M.functor_rows(R, rs, re_, qlb, tlb, ss, se, t0, q0)
    if fix is not None:
        # the fixture belongs to exact records: the kernel is deterministic, so they must be the ones it was made from (regenerate it with the two
        # scripts above when a kernel change moves a record — the assertion says so instead of comparing rows of other records)
        for key, arr in (("s_cp", rs["cp"]), ("s_n", rs["n"]), ("s_d", rs["d"]), ("s_score", rs["score"]), ("e_cp", re_["cp"]), ("e_a", re_["a"]), ("e_b", re_["b"]), ("e_s", re_["s"])):
            assert np.array_equal(np.asarray(arr), fix[f"{variant}_{key}"]), f"records changed ({key}): regenerate tests/golden/ref_functor_rows.npz"
        if live:
            assert np.array_equal(rows_s, fix[f"{variant}_rows_s"]) and np.array_equal(rows_e, fix[f"{variant}_rows_e"])      # the committed rows ARE the reference's
        rows_s, rows_e = fix[f"{variant}_rows_s"], fix[f"{variant}_rows_e"]
    for mask, rows in ((L.MASK_SURF, rows_s), (L.MASK_EDGE, rows_e)):
        Gg, cost, counts = m.linearize(0, t0, q0, mask)
        rr = _cauchy_rows(rows)
        Gr = rr.T @ rr
        assert np.abs(Gg - Gr).max() <= 1e-9 * np.abs(Gr).max(), (variant, mask, np.abs(Gg - Gr).max() / np.abs(Gr).max())
        cost_ref = 0.5 * np.log1p(rows[:, 7] ** 2).sum()
        assert abs(cost - cost_ref) <= 1e-9 * max(1.0, cost_ref)


class _GpuBackend:
    """tests/frontend_chain.py backend on the HIP path: lili_voxel_filter, lili_map_set, lili_s2m_iterate (front-end flavour)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.m = L.ScanToMapMatcher(ctx, L.make_params("frontend"))

    def voxel(self, pts4, leaf=0.4):
        return L.api.voxel_filter(self.ctx, np.ascontiguousarray(pts4, np.float32), leaf)[0] if pts4.shape[0] else pts4

    def set_map(self, clouds_world, leaf=0.4):
        cat = np.ascontiguousarray(np.concatenate(clouds_world, 0), np.float32)
        mp = self.voxel(cat, leaf)
        self.m.set_input_cloud(L.KIND_SURF, mp)
        return mp.shape[0]

    def match(self, queries4, q, t, n_iter):
        self.m.set_queries(0, L.KIND_SURF, np.ascontiguousarray(queries4, np.float32))
        self.m.pose_set(0, t, q)
        self.m.iterate(0, n_iter, L.MASK_SURF)
        t2, q2, st = self.m.pose_get(0)
        assert st == 0
        return (q2 if q2[0] >= 0 else -q2), t2


def test_gpu_frontend_chain_vs_reference_node(gpu_ctx):
    """Pose parity with the reference's OWN front-end node (LiLi-OM/src/LidarOdometry.cpp compiled unmodified, one GN step per
    ceres::Solve — tests/golden/ref_frontend.npz): Livox extractor -> voxel filter -> local map of the last 20 frames ->
    6 (8 on the first frame) re-association + Gauss-Newton iterations per frame, all on the HIP path, against the poses the
    reference node held after every frame.  Tolerance = north star (1e-4 m / 1e-4 rad)."""
    from tests import frontend_chain as F
    g = np.load(os.path.join(G, "ref_frontend.npz"))
    frames, stamps, imu_t, gyr = M.frontend_inputs()
    integ = L.api.ImuIntegrator()
    ex = L.LivoxExtractor(gpu_ctx)
    surf = [ex.extract(frames[k], _q_imu(integ, stamps, imu_t, gyr, k))["surf"][:, [0, 1, 2, 7]] for k in range(M.FRONTEND_FRAMES)]
    a, r = F.run_frontend_chain(_GpuBackend(gpu_ctx), surf, scan_match_cnt=int(M.FRONTEND_PARAMS["/lidar_odometry/scan_match_cnt"]))
    ref = g["abs_pose"]
    assert np.abs(a[:, 4:] - ref[:, 4:]).max() < 1e-4, np.abs(a[:, 4:] - ref[:, 4:]).max()
    assert np.abs(a[:, :4] - ref[:, :4]).max() < 5e-5           # quaternion components: < 1e-4 rad
    assert np.abs(r - g["rel_pose"]).max() < 1e-4
    print("GPU front-end chain vs reference node: max |dt| = %.3g m, max |dq| = %.3g" % (np.abs(a[:, 4:] - ref[:, 4:]).max(), np.abs(a[:, :4] - ref[:, :4]).max()))


@pytest.mark.parametrize("flavour", ["livox", "rot"])
def test_gpu_backend_matcher_vs_reference(gpu_ctx, flavour):
    """k_associate_surf / k_associate_edge / k_linearize_* vs the reference's OWN back-end association functions and residual
    blocks (tests/golden/ref_backend.npz: BackendFusion.cpp member functions compiled from the reference text): the same
    correspondences in the same order (cp bit-exact), plane / line parameters within the f32-record tolerance, scores to
    3e-7, and the Gram of lili_s2m_linearize equal to the robustified sum over the reference's raw residual blocks."""
    g = np.load(os.path.join(G, "ref_backend.npz"))
    i = M.backend_inputs(flavour)
    P = L.make_params(flavour)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, i["surf_map"] if flavour == "livox" else np.ascontiguousarray(i["surf_map"][:, :3]))
    m.set_input_cloud(L.KIND_EDGE, np.ascontiguousarray(i["edge_map"][:, :3]))
    m.set_queries(0, L.KIND_SURF, i["surf_q"] if flavour == "livox" else np.ascontiguousarray(i["surf_q"][:, :3]))
    m.set_queries(0, L.KIND_EDGE, np.ascontiguousarray(i["edge_q"][:, :3]))
    Q2, T2 = L.api.assoc_transform(i["t0"], i["q0"], P)             # the product's own derivation of the association pose
    assert np.abs(np.asarray(Q2) - i["Q2"]).max() < 1e-15 and np.abs(np.asarray(T2) - i["T2"]).max() < 1e-14
    ns = m.find_corresponding_surf_features(0, Q2, T2)
    ne = m.find_corresponding_corner_features(0, Q2, T2)
    rs, re_ = m.surf_records(0, ns), m.edge_records(0, ne)
    ref_s, ref_sc, ref_e = g[f"{flavour}_surf_rec"], g[f"{flavour}_surf_score"], g[f"{flavour}_edge_rec"]
    assert ns == ref_s.shape[0] and ne == ref_e.shape[0]
    assert np.array_equal(rs["cp"], ref_s[:, 0:3]) and np.array_equal(re_["cp"], ref_e[:, 0:3])     # same queries kept, same order
    np.testing.assert_allclose(rs["n"], ref_s[:, 3:6], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(rs["d"], ref_s[:, 6], rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(rs["score"], ref_sc, rtol=3e-7)
    ga, gb = re_["a"], re_["b"]
    swap = np.abs(ga - ref_e[:, 3:6]).max(1) > np.abs(ga - ref_e[:, 6:9]).max(1)        # eigenvector sign is arbitrary
    ra = np.where(swap[:, None], ref_e[:, 6:9], ref_e[:, 3:6]); rb = np.where(swap[:, None], ref_e[:, 3:6], ref_e[:, 6:9])
    np.testing.assert_allclose(ga, ra, rtol=0, atol=2e-6)
    np.testing.assert_allclose(gb, rb, rtol=0, atol=2e-6)
    assert np.array_equal(re_["s"], ref_e[:, 9])
    for mask, rows in ((L.MASK_SURF, g[f"{flavour}_surf_rows"]), (L.MASK_EDGE, g[f"{flavour}_edge_rows"])):
        Gg, cost, counts = m.linearize(0, i["t0"], i["q0"], mask)
        rr = _cauchy_rows(np.c_[rows[:, 1:8], rows[:, 0]])
        Gr = rr.T @ rr
        assert np.abs(Gg - Gr).max() <= 2e-6 * np.abs(Gr).max(), (flavour, mask, np.abs(Gg - Gr).max() / np.abs(Gr).max())
        cost_ref = 0.5 * np.log1p(rows[:, 0] ** 2).sum()
        assert abs(cost - cost_ref) <= 2e-6 * max(1.0, cost_ref)


def test_gpu_format_convert_vs_reference(gpu_ctx):
    """k_custom_to_pcl (lili_livox_custom_to_cloud) vs the reference's livoxLidarHandler (tests/golden/ref_format.npz)."""
    g = np.load(os.path.join(G, "ref_format.npz"))
    pts, zero = M.format_inputs()
    a = L.api.livox_custom_to_cloud(gpu_ctx, pts)
    assert a.shape[0] == int(g["n"]) and _sha(a) == str(g["sha"])
    assert np.array_equal(_bits(a[::16]), _bits(g["every16"]))
    assert np.array_equal(_bits(L.api.livox_custom_to_cloud(gpu_ctx, zero)), _bits(g["zero_case"]))


@pytest.mark.parametrize("kind,name,map_leaf,leaf", [(L.KIND_SURF, "surf", M.LM_SURF_MAP_LEAF, M.LM_SURF_LEAF), (L.KIND_EDGE, "edge", M.LM_EDGE_MAP_LEAF, M.LM_EDGE_LEAF)])
def test_gpu_local_map_vs_reference_backend(gpu_ctx, kind, name, map_leaf, leaf):
    """lili_localmap_push / _commit (device ring buffer + transform + VoxelGrid + map index) fed the way INTEGRATION.md binds
    buildLocalMapWithLandMark, vs the maps the reference's own text produced for the same 7 keyframes (tests/golden/ref_localmap.npz).
    The device map is read back through the matcher: every reference map point, used as a query at the identity pose, must find ITSELF
    (same index = same voxel order) at distance 0; points of >= 3-point voxels may differ in the last bit of the centroid (PCL's unstable
    sort chooses the summation order), hence the 1e-11 m^2 allowance on a minority of points."""
    g = np.load(os.path.join(G, "ref_localmap.npz"))
    i = M.localmap_inputs()
    feats = i[name]
    P = L.make_params("rot")        # only the neighbour lists are read back (the Livox flavour would require reflectivity on the queries)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    gpu_ctx.set_debug(True)
    find = m.find_corresponding_surf_features if kind == L.KIND_SURF else m.find_corresponding_corner_features
    ident_q, ident_t = np.array([1.0, 0, 0, 0]), np.zeros(3)

    def check(ref_map, n_map):
        assert n_map == ref_map.shape[0]
        m.set_queries(0, kind, np.ascontiguousarray(ref_map[:, :3]))
        find(0, ident_q, ident_t)
        idx, d2 = m.neighbors(0, kind, ref_map.shape[0])
        assert np.array_equal(idx[:, 0], np.arange(ref_map.shape[0]))
        assert d2[:, 0].max() < 1e-11 and (d2[:, 0] == 0).mean() > 0.75

    # first keyframe: its own raw features moved by T_bl (L:1390-1404)
    lm = L.LocalMap(gpu_ctx, kind, width=M.LM_WIDTH, leaf=map_leaf)
    lm.push(feats[0], i["t_bl"], i["q_bl"])
    n_raw, n_map = lm.commit()
    assert n_raw == feats[0].shape[0]
    check(g[f"kf0_{name}_map"], n_map)
    # from then on: the down-sampled features of the previous keyframes under their optimised poses, ring of local_map_width
    lm = L.LocalMap(gpu_ctx, kind, width=M.LM_WIDTH, leaf=map_leaf)
    for k in range(1, len(feats)):
        ds_prev = g[f"kf{k - 1}_{name}_ds"]
        t, q = L.api.keyframe_map_pose(i["poses"][k - 1][4:7], i["poses"][k - 1][:4], i["t_bl"], i["q_bl"])
        lm.push(ds_prev, t, q)
        n_raw, n_map = lm.commit()
        assert n_raw == sum(g[f"kf{j}_{name}_ds"].shape[0] for j in range(max(0, k - M.LM_WIDTH), k))
        check(g[f"kf{k}_{name}_map"], n_map)
    # the keyframe's own down-sampling (ds_filter_surf / ds_filter_edge, L:1502-1511)
    for k in range(len(feats)):
        ds, cnt = L.api.voxel_filter(gpu_ctx, feats[k], leaf)
        ref = g[f"kf{k}_{name}_ds"]
        assert ds.shape == ref.shape
        same = (_bits(ds) == _bits(ref)).all(1)
        assert same[cnt <= 2].all() and same.mean() > 0.85
        np.testing.assert_allclose(ds, ref, rtol=1e-6, atol=1e-5)


def test_gpu_marginalisation_feed_vs_reference(gpu_ctx):
    """lili_s2m_linearize + lili_marg_add_lidar on the GPU's own records of the ref_backend Livox keyframe vs the A, b the
    reference's ResidualBlockInfo::Evaluate + ThreadsConstructA build from the reference's records (f32-record tolerance)."""
    g = np.load(os.path.join(G, "ref_marg.npz"))
    i = M.backend_inputs("livox")
    P = L.make_params("livox")
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, i["surf_map"])
    m.set_input_cloud(L.KIND_EDGE, np.ascontiguousarray(i["edge_map"][:, :3]))
    m.set_queries(0, L.KIND_SURF, i["surf_q"])
    m.set_queries(0, L.KIND_EDGE, np.ascontiguousarray(i["edge_q"][:, :3]))
    Q2, T2 = L.api.assoc_transform(i["t0"], i["q0"], P)
    m.find_corresponding_surf_features(0, Q2, T2)
    m.find_corresponding_corner_features(0, Q2, T2)
    Gm, cost, counts = m.linearize(0, i["t0"], i["q0"], L.MASK_SURF | L.MASK_EDGE)
    assert int(counts[0]) + int(counts[1]) == int(g["n_rows"])
    A, b = np.zeros((M.MARG_POS, M.MARG_POS)), np.zeros(M.MARG_POS)
    L.api.marg_add_lidar(Gm, A, b, M.MARG_IDX_T, M.MARG_IDX_Q)
    assert np.abs(A - g["A"]).max() <= 2e-5 * np.abs(g["A"]).max()
    assert np.abs(b - g["b"]).max() <= 2e-5 * np.abs(g["b"]).max()


@pytest.mark.parametrize("flavour", ["livox", "rot"])
def test_ceres_seam_batch_factor_equals_reference_blocks(gpu_ctx, tmp_path, flavour):
    """The drop-in claim at the Ceres seam, in C++ and with the reference's own factor header: inside one
    ceres::Problem (stand-in) ONE lili::LidarBatchFactor — include/lili_ceres_adapter.h, the binding of INTEGRATION.md §1 —
    presents the solver with the same J^T J, J^T r and cost as the thousands of AutoDiffCostFunction<LidarEdgeFactor |
    LidarPlaneNormFactor> + CauchyLoss(1.0) blocks the reference adds for the same correspondences
    (oracle/refshim/ref_seam.cpp -> oracle/_ref/seam_check, prebuilt against /root/reference's LidarKeyframeFactor.h)."""
    import subprocess
    exe = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "seam_check")
    exe = os.path.abspath(exe)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/seam_check not built")
    i = M.backend_inputs(flavour)
    P = L.make_params(flavour)
    f = tmp_path / "seam.bin"
    with open(f, "wb") as fh:
        fh.write(np.array([i["surf_map"].shape[0], i["edge_map"].shape[0], i["surf_q"].shape[0], i["edge_q"].shape[0],
                           1 if flavour == "rot" else 0], np.int32).tobytes())
        fh.write(bytes(P))
        for k in ("surf_map", "edge_map", "surf_q", "edge_q"):
            fh.write(np.ascontiguousarray(i[k], np.float32).tobytes())
        fh.write(np.r_[i["t0"], i["q0"]].astype(np.float64).tobytes())
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(line.split("=", 1) for line in out.stdout.strip().splitlines() if "=" in line)
    assert "error" not in kv, kv
    g = np.load(os.path.join(G, "ref_backend.npz"))
    assert int(kv["n_surf"]) == g[f"{flavour}_surf_rec"].shape[0] and int(kv["n_edge"]) == g[f"{flavour}_edge_rec"].shape[0]
    assert int(kv["blocks_batch"]) == 1 and int(kv["blocks_reference"]) == int(kv["n_surf"]) + int(kv["n_edge"]) > 1000
    assert float(kv["H_rel_diff"]) < 1e-9 and float(kv["g_rel_diff"]) < 1e-9, kv
    cb, cr = float(kv["cost_batch"]), float(kv["cost_reference"])
    assert abs(cb - cr) <= 1e-9 * max(1.0, cr), kv
    # the window binding: ONE lili::LidarWindowFactor over three keyframes == one LidarBatchFactor per keyframe, bit for bit, block-diagonal
    assert int(kv["window_blocks"]) == 3 and int(kv["window_residuals"]) == 27
    assert float(kv["window_vs_batch_max_abs_diff"]) == 0.0 and float(kv["window_off_diagonal_max"]) == 0.0, kv


def test_ros_node_seam_gpu_node_equals_reference_node(gpu_ctx, tmp_path):
    """The drop-in claim at the ROS-node seam, in C++: the reference's LiLi-OM-ROT Preprocessing node (compiled
    unmodified) and a node with the same topics whose cloud handler is the binding of INTEGRATION.md §3 (lili_imu_integrate +
    lili_extract_rot through the C ABI) get the same ROS messages; every cloud the reference publishes comes out of the GPU node
    on the same topic, with the same stamp, the same number of points and — up to the in-voxel summation order of the surf centroids — the same bits
    (oracle/refshim/ref_seam_pre.cpp -> oracle/_ref/seam_pre_check)."""
    import subprocess
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "seam_pre_check"))
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/seam_pre_check not built")
    scans, stamps, imu_t, gyr = M.rot_inputs()
    f = tmp_path / "pre.bin"
    with open(f, "wb") as fh:
        fh.write(np.array([len(scans), len(imu_t), 64, 4], np.int32).tobytes())
        fh.write(np.array(M.ROT_QLB, np.float64).tobytes())
        for s_, t_ in zip(scans, stamps):
            fh.write(np.float64(t_).tobytes()); fh.write(np.int32(s_.shape[0]).tobytes()); fh.write(np.ascontiguousarray(s_, np.float32).tobytes())
        fh.write(np.ascontiguousarray(imu_t, np.float64).tobytes()); fh.write(np.ascontiguousarray(gyr, np.float64).tobytes())
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(line.split("=", 1) for line in out.stdout.strip().splitlines() if "=" in line)
    assert "error" not in kv, kv
    assert int(kv["messages_reference"]) == int(kv["messages_gpu"]) == 6            # 2 processed scans x 3 topics
    assert int(kv["same_topics_stamps"]) == 1 and int(kv["same_point_counts"]) == 1
    assert int(kv["points"]) > 20_000 and int(kv["bit_identical_points"]) > 0.99 * int(kv["points"])
    assert float(kv["max_abs_diff"]) < 2e-5, kv


def test_ros_node_seam_livox_chain(gpu_ctx, tmp_path):
    """The Livox chain at the ROS-node seam, in C++: the reference's FormatConvert + Preprocessing nodes (both compiled unmodified)
    next to ONE node that converts the CustomMsg into device memory (lili_livox_custom_to_cloud), integrates the gyro
    (lili_imu_integrate) and extracts from the device cloud (lili_extract_livox): same messages in, the same six messages out —
    payload (x, y, z, intensity, curvature) of every point bit-identical, stored normals within 2e-6 up to sign
    (oracle/refshim/ref_seam_livox.cpp -> oracle/_ref/seam_livox_check)."""
    import subprocess
    from lili_om_amd import synth
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "seam_livox_check"))
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/seam_livox_check not built")
    rng = np.random.default_rng(17)
    stamps = 70.0 + 0.1 * np.arange(4)
    imu_t = 69.97 + 0.005 * np.arange(100)
    gyr = 0.2 * rng.standard_normal((100, 3)) + np.array([0.1, -0.05, 0.3])
    f = tmp_path / "livox.bin"
    dt = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1")])
    with open(f, "wb") as fh:
        fh.write(np.array([4, len(imu_t)], np.int32).tobytes())
        for k in range(4):
            s = synth.make_livox_scan(80 + k, inject_bad=False)                 # x y z intensity(line + 0.1 t) curvature(0.1 refl)
            line = np.floor(s[:, 3]).astype(np.int64)
            tfrac = np.clip((s[:, 3] - line) / 0.1, 0.0, 1.0)
            order = np.argsort(tfrac, kind="stable")                            # CustomMsg points arrive in time order
            pts = np.zeros(s.shape[0], dt)
            pts["offset_time"] = np.round(tfrac[order] * 99_000_000).astype(np.uint32)
            pts["x"], pts["y"], pts["z"] = s[order, 0], s[order, 1], s[order, 2]
            pts["reflectivity"] = np.clip(np.round(s[order, 4] * 10), 0, 255).astype(np.uint8)
            pts["line"] = np.clip(line[order], 0, 5).astype(np.uint8)
            fh.write(np.float64(stamps[k]).tobytes()); fh.write(np.int32(pts.shape[0]).tobytes()); fh.write(pts.tobytes())
        fh.write(np.ascontiguousarray(imu_t, np.float64).tobytes()); fh.write(np.ascontiguousarray(gyr, np.float64).tobytes())
    assert dt.itemsize == 19
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(line.split("=", 1) for line in out.stdout.strip().splitlines() if "=" in line)
    assert "error" not in kv, kv
    assert int(kv["messages_reference"]) == int(kv["messages_gpu"]) == 6
    assert int(kv["same_topics_stamps"]) == 1 and int(kv["same_point_counts"]) == 1
    assert int(kv["points"]) > 60_000 and int(kv["bit_identical_payload"]) == int(kv["points"]), kv
    assert float(kv["max_abs_normal_diff"]) < 2e-6, kv
