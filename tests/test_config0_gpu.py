"""BASELINE configs[0] on the GPU path: one HDL-64E-like scan (~130 k points) through the LiLi-OM-ROT extractor
(ds_rate 4, 64 rings) and ONE back-end outer iteration (edge + surf association, linearisation, GN update) against a
500 k-point surf map (+ edge map) — feature indices bit-exact, pose delta within 1e-4 m / 1e-4 rad of the oracle."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu


def test_config0_extract_and_one_iteration(gpu_ctx, oracle):
    w = synth.make_workload(n_map=500_000, n_az=2031, half_extent=(150.0, 150.0))
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 50.0, np.float32)], 1)
    assert 120_000 < raw.shape[0] < 135_000
    P, PO = L.make_params("rot"), oracle.params("rot")
    q_lb = list(P.q_lb)
    # --- extraction: indices bit-exact
    ex = L.RotExtractor(gpu_ctx, n_scans=64, ds_rate=4)
    g = ex.extract(raw, (1.0, 0, 0, 0), q_lb, debug=True)
    o = oracle.extract_rot(raw, (1.0, 0, 0, 0), q_lb, oracle.rot_params(ds_rate=4, atan_mode=2, stable_sort=1))
    for k in ("full_src", "edge_idx", "flat_idx", "lessflat_idx", "label"):
        assert np.array_equal(g[k], o[k]), k
    assert np.array_equal(g["surf"].view(np.uint32), o["surf"].view(np.uint32))
    assert np.array_equal(g["edge"].view(np.uint32), o["full"][o["edge_idx"]].view(np.uint32))
    assert len(g["edge"]) > 100 and len(g["surf"]) > 2000
    # --- one outer iteration of the back-end matcher on the extracted features (body frame after the q_lb deskew)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
    m.set_input_cloud(L.KIND_EDGE, w["edge_map_xyz"])
    surf_q, edge_q = g["surf"][:, :3], g["edge"][:, :3]
    m.set_queries(0, L.KIND_SURF, surf_q)
    m.set_queries(0, L.KIND_EDGE, edge_q)
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
    m.pose_set(0, t0, q0)
    m.iterate(0, 1, L.MASK_SURF | L.MASK_EDGE)
    tg, qg, st = m.pose_get(0)
    assert st == 0
    # oracle: same iteration
    Q2, T2 = L.api.assoc_transform(t0, q0, P)
    tree, etree = oracle.KdTree(w["map_xyz"]), oracle.KdTree(w["edge_map_xyz"])
    rs = oracle.associate_surf(tree, None, surf_q, None, Q2, T2, PO, nthreads=8)
    re_ = oracle.associate_edge(etree, edge_q, Q2, T2, PO)
    assert rs["count"] > 500
    Gs, _, _ = oracle.linearize_surf(rs, t0, q0, PO, (1000.0, max(rs["count"], 1)))
    Ge, _, _ = oracle.linearize_edge(re_, t0, q0, PO, (200.0, max(re_["count"], 1)))
    sto, to, qo, _ = oracle.gn_step(Gs + Ge, t0, q0)
    assert sto == 0
    assert np.abs(tg - to).max() < 1e-4
    dq = synth.quat_mul(qg * np.array([1, -1, -1, -1]), qo)
    assert 2 * np.arcsin(min(1.0, np.linalg.norm(dq[1:]))) < 1e-4
    assert np.linalg.norm(to - t0) > 1e-3          # the step did move the pose
