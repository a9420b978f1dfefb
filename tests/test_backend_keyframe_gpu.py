"""lili_backend_keyframe_prepare (round 6): what BackendFusion does per keyframe before ceres::Solve — ring push, both local maps (VoxelGrid + index), the
new keyframe's down-sampling, the window's associations (L/src/BackendFusion.cpp:830-980, 1387-1528) — as ONE device-resident call.

* against the chain of separate C calls through host buffers (lili_localmap_push / _commit, lili_voxel_filter, lili_s2m_set_queries, lili_s2m_associate_window): the same
  counts and the same correspondence records bit for bit, keyframe after keyframe, with the ring popping (width 3);
* against the reference's own maps (tests/golden/ref_localmap.npz: BackendFusion.cpp's buildLocalMapWithLandMark + downSampleCloud text compiled): the same map sizes;
* the joining keyframe taken from the previous newest slot on the device (join_slot) gives the same result as handing its features over from the host."""
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

K = 3


def _assoc_poses(P, poses, ks):
    out = [L.api.assoc_transform(poses[k][4:7], poses[k][:4], P) for k in ks]
    return [a[1] for a in out], [a[0] for a in out]


@pytest.mark.parametrize("join_from_slot", [False, True])
def test_keyframe_prepare_equals_the_separate_calls_and_the_reference_maps(join_from_slot):
    g = np.load(os.path.join(G, "ref_localmap.npz"))
    i = M.localmap_inputs()
    n_kf = len(i["surf"])
    P = L.make_params("rot")
    mask = L.MASK_SURF | L.MASK_EDGE
    ca, cb = L.Context(0), L.Context(0)
    try:
        # ---- the calls one by one, every cloud through host buffers
        ma = L.ScanToMapMatcher(ca, P)
        lm = [L.LocalMap(ca, L.KIND_SURF, width=M.LM_WIDTH, leaf=M.LM_SURF_MAP_LEAF), L.LocalMap(ca, L.KIND_EDGE, width=M.LM_WIDTH, leaf=M.LM_EDGE_MAP_LEAF, max_sq_radius=P.edge_gate)]
        # ---- one call per keyframe
        mb = L.ScanToMapMatcher(cb, P)
        bk = L.BackendKeyframes(cb, P, leaf_surf=M.LM_SURF_LEAF, leaf_edge=M.LM_EDGE_LEAF, width=M.LM_WIDTH)
        ds_prev = None
        for k in range(n_kf):
            win = list(range(max(0, k - K + 1), k + 1))                 # keyframes of the window, oldest first
            slots = [j % K for j in win]
            ts, qs = _assoc_poses(P, i["poses"], win)
            # staged
            if k > 0:
                tj, qj = L.api.keyframe_map_pose(i["poses"][k - 1][4:7], i["poses"][k - 1][:4], i["t_bl"], i["q_bl"])
                lm[0].push(ds_prev[0], tj, qj); lm[1].push(ds_prev[1], tj, qj)
                sizes_a = (lm[0].commit(), lm[1].commit())
            ds = [L.api.voxel_filter(ca, i["surf"][k], M.LM_SURF_LEAF)[0], L.api.voxel_filter(ca, i["edge"][k], M.LM_EDGE_LEAF)[0]]
            ma.set_queries(slots[-1], L.KIND_SURF, ds[0]); ma.set_queries(slots[-1], L.KIND_EDGE, ds[1])
            counts_a = ma.associate_window(slots, ts, qs, mask) if k > 0 else None
            # fused
            join = None
            if k > 0:
                join = ((k - 1) % K, tj, qj) if join_from_slot else (ds_prev[0], ds_prev[1], tj, qj)
            counts_b, info = bk.prepare(join, i["surf"][k], i["edge"][k], slots, ts, qs)
            assert info["n_query"] == (ds[0].shape[0], ds[1].shape[0])
            if k == 0:
                assert not info["associated"]
            else:
                assert info["associated"] and counts_b == counts_a, (k, counts_a, counts_b)
                assert (info["n_map_raw"][0], info["n_map"][0]) == sizes_a[0] and (info["n_map_raw"][1], info["n_map"][1]) == sizes_a[1]
                assert info["n_map"] == (g[f"kf{k}_surf_map"].shape[0], g[f"kf{k}_edge_map"].shape[0])          # the reference's own maps for this keyframe
                assert sum(a + b for a, b in counts_b) > 50
                for s in slots:
                    ra, rb = ma.surf_records(s, 4096), mb.surf_records(s, 4096)
                    for key in ("query_index", "cp", "n", "d", "score"):
                        assert np.array_equal(ra[key], rb[key]), (k, s, key)
                    ea, eb = ma.edge_records(s, 4096), mb.edge_records(s, 4096)
                    for key in ("query_index", "cp", "a", "b", "s"):
                        assert np.array_equal(ea[key], eb[key]), (k, s, key)
            ds_prev = ds
    finally:
        ca.close(); cb.close()


def test_keyframe_prepare_rejects_bad_arguments(gpu_ctx):
    P = L.make_params("rot")
    bk = L.BackendKeyframes(gpu_ctx, P)
    z = np.zeros((10, 4), np.float32)
    with pytest.raises(L.LiliError):
        bk.prepare((7, np.zeros(3), np.array([1.0, 0, 0, 0])), z, z, [0], [np.zeros(3)], [np.array([1.0, 0, 0, 0])])      # a slot without queries
    with pytest.raises(L.LiliError):
        bk.prepare(None, z, z, [0, 0], [np.zeros(3)] * 2, [np.array([1.0, 0, 0, 0])] * 2)                              # duplicate slot (a map exists only after a join)
