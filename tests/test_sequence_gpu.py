"""BASELINE configs[1] substitute (FR_IOSB_Short is not available offline): 100 synthetic Livox-Horizon-like frames (~24 k points,
6 lines) through the Livox extractor + the front-end scan-to-map matcher.  GPU trajectory vs oracle trajectory within the
north-star tolerance (1e-4 m / 1e-4 rad) at EVERY frame, and both close to the ground truth (ATE)."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth
from tests import seq_harness as H

pytestmark = pytest.mark.gpu
NTH = min(32, __import__("os").cpu_count() or 1)


def test_livox_sequence_pose_parity(gpu_ctx, oracle):
    gt = H.gt_pose_circuit
    frames = H.make_frames(100, gt=gt)
    P = L.make_params("frontend")
    PO = oracle.params("frontend")
    ex = L.LivoxExtractor(gpu_ctx)
    m = L.ScanToMapMatcher(gpu_ctx, P)

    def gpu_match(map_xyzc, qry, t0, q0, n_outer):
        m.set_input_cloud(L.KIND_SURF, map_xyzc)
        m.set_queries(0, L.KIND_SURF, qry)
        m.pose_set(0, t0, q0)
        m.iterate(0, n_outer, L.MASK_SURF)
        t, q, st = m.pose_get(0)
        assert st == 0
        return t, q

    def cpu_match(map_xyzc, qry, t0, q0, n_outer):
        tree = oracle.KdTree(map_xyzc[:, :3])
        t, q = np.array(t0, np.float64), np.array(q0, np.float64)
        for _ in range(n_outer):
            rs = oracle.associate_surf(tree, None, qry[:, :3], None, q, t, PO, nthreads=NTH)
            G, _, _ = oracle.linearize_surf(rs, t, q, PO)
            st, t, q, _ = oracle.gn_step(G, t, q)
            assert st == 0
        return t, q

    pg = H.run_sequence(frames, lambda s: ex.extract(s)["surf"], gpu_match, gt=gt)
    pc = H.run_sequence(frames, lambda s: oracle.extract_livox(s)["surf"], cpu_match, gt=gt)
    assert len(pg) == len(pc) == 100
    for f, ((tg, qg), (tc, qc)) in enumerate(zip(pg, pc)):
        assert np.abs(tg - tc).max() < 1e-4, f
        dq = synth.quat_mul(qg * np.array([1, -1, -1, -1]), qc)
        assert 2 * np.arcsin(min(1.0, np.linalg.norm(dq[1:]))) < 1e-4, f
    rms_g, max_g = H.ate(pg, gt)
    rms_c, max_c = H.ate(pc, gt)
    print(f"ATE vs ground truth: GPU rms {rms_g:.4f} max {max_g:.4f} | oracle rms {rms_c:.4f} max {max_c:.4f}")
    assert rms_g < 0.15 and abs(rms_g - rms_c) < 1e-4
