"""BASELINE configs[4] on the GPU: the reference back-end's joint window — 3 keyframes x (t, q, speed-bias), speed-bias
priors, two IMU factors and the lidar edge + plane blocks of every keyframe with CauchyLoss(1), solved by a dense Levenberg-Marquardt
loop with Ceres' default rules on FIXED correspondences (L/src/BackendFusion.cpp:843-1007) — with the lidar part coming from the HIP path:
every cost / Jacobian evaluation of the solver is ONE lili_s2m_linearize per keyframe (Gram + robust cost) turned into the 9-residual
block of include/lili_ceres_adapter.h by lili_gram_to_factor.  The window the solver ends in must be the one it ends in when the lidar
blocks are the oracle's per-residual rows (same association, same loss): <= 1e-4 m / 1e-4 rad (north star), measured ~1e-7."""
import numpy as np
import pytest

import lili_om_amd as L
from oracle import lo_window as W
from tests import window_harness as H

pytestmark = pytest.mark.gpu


def _angle(qa, qb):
    d = W.qmul(W.qinv(qa), qb)
    return 2.0 * np.arctan2(np.linalg.norm(d[1:]), abs(d[0]))


def test_window_lm_with_gpu_lidar_blocks_equals_oracle_rows(gpu_ctx, oracle):
    win = H.make_window(n_surf=2500, n_edge=200)
    room, P = win["room"], win["P"]
    mask = L.MASK_SURF | L.MASK_EDGE
    # ---- GPU side: one slot per keyframe, correspondences found ONCE at the initial window (findCorresponding*Features, L:929-936)
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, np.c_[room["map_xyz"], room["map_refl"]])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    assoc = []
    for k, kf in enumerate(win["kfs"]):
        m.set_queries(k, L.KIND_SURF, np.c_[kf["q_xyz"], kf["q_refl"]])
        m.set_queries(k, L.KIND_EDGE, kf["eq_xyz"])
        assoc.append(L.api.assoc_transform(win["init"][k]["t"], win["init"][k]["q"], P))
    n_gpu = m.associate_window(list(range(H.N_KF)), [a[1] for a in assoc], [a[0] for a in assoc], mask)       # both finders of every keyframe, one call
    evals = [0]

    def gpu_block(k):
        def fn(t, q):
            G, cost, counts = m.linearize(k, t, q, mask)
            evals[0] += 1
            res, jac = L.api.gram_to_factor(G, cost)
            return res, [jac[:, :3], jac[:, 3:7]]
        return fn

    # ---- oracle side: the same association on the CPU, per-residual rows, loss + corrector as ceres applies them
    PO = oracle.params("livox", loss=0)
    tree_s, tree_e = oracle.KdTree(room["map_xyz"]), oracle.KdTree(room["edge_map_xyz"])
    recs = []
    for k, kf in enumerate(win["kfs"]):
        Q2, T2 = L.api.assoc_transform(win["init"][k]["t"], win["init"][k]["q"], P)
        recs.append((oracle.associate_surf(tree_s, room["map_refl"], kf["q_xyz"], kf["q_refl"], Q2, T2, PO), oracle.associate_edge(tree_e, kf["eq_xyz"], Q2, T2, PO)))
        assert (recs[-1][0]["count"], recs[-1][1]["count"]) == n_gpu[k] and n_gpu[k][0] > 1500 and n_gpu[k][1] > 50

    def oracle_block(k):
        def fn(t, q):
            rows = np.concatenate([oracle.linearize_rows(recs[k][0], t, q, PO, kind="surf"), oracle.linearize_rows(recs[k][1], t, q, PO, kind="edge")])
            J, r, cost = H.robust_rows(rows)
            return r, [J[:, :3], J[:, 3:7]], cost
        return fn

    def gpu_joint(*tq):          # the same lidar terms as ONE block fed by lili_s2m_linearize_window: one synchronisation per evaluation
        ts, qs = tq[0::2], tq[1::2]
        res, jacs = [], [np.zeros((9 * H.N_KF, 3 if i % 2 == 0 else 4)) for i in range(2 * H.N_KF)]
        for k, (G, cost, counts) in enumerate(m.linearize_window(list(range(H.N_KF)), ts, qs, mask)):
            r, jac = L.api.gram_to_factor(G, cost)
            res.append(r)
            jacs[2 * k][9 * k:9 * k + 9] = jac[:, :3]
            jacs[2 * k + 1][9 * k:9 * k + 9] = jac[:, 3:7]
        return np.concatenate(res), jacs

    sol_j, info_j = W.ceres_lm(H.build_problem(win, None, joint_lidar=gpu_joint), max_num_iterations=15)
    log_g, log_o = [], []
    sol_g, info_g = W.ceres_lm(H.build_problem(win, gpu_block), max_num_iterations=15, log=log_g)
    assert info_j["iterations"] == info_g["iterations"] and abs(info_j["cost"] - info_g["cost"]) <= 1e-9 * info_g["cost"]
    for k in range(H.N_KF):
        assert np.abs(sol_j[f"t{k}"] - sol_g[f"t{k}"]).max() < 1e-9 and np.abs(sol_j[f"q{k}"] - sol_g[f"q{k}"]).max() < 1e-9
    sol_o, info_o = W.ceres_lm(H.build_problem(win, oracle_block), max_num_iterations=15, log=log_o)
    assert evals[0] >= 3 * 10                                     # three Grams per solver evaluation
    assert info_g["iterations"] == info_o["iterations"] and info_g["successful_steps"] == info_o["successful_steps"]
    assert abs(info_g["cost"] - info_o["cost"]) <= 1e-6 * info_o["cost"]
    for a, b in zip(log_g, log_o):                                # the same accept / reject decisions, the same trust-region radii
        assert (a["rho"] > 1e-3) == (b["rho"] > 1e-3) and a["radius"] == b["radius"]
    for k in range(H.N_KF):
        dt = np.linalg.norm(sol_g[f"t{k}"] - sol_o[f"t{k}"])
        da = _angle(sol_g[f"q{k}"], sol_o[f"q{k}"])
        assert dt < 1e-4 and da < 1e-4, (k, dt, da)
        assert np.abs(sol_g[f"sb{k}"] - sol_o[f"sb{k}"]).max() < 1e-4
        assert np.linalg.norm(sol_g[f"t{k}"] - win["kfs"][k]["t_true"]) < np.linalg.norm(win["init"][k]["t"] - win["kfs"][k]["t_true"])
