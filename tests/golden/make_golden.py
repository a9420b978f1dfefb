#!/usr/bin/env python
"""Generates tests/golden/*.npz from the ORACLE on fixed-seed synthetic inputs.

The reference has no golden vectors, so these fixtures pin the oracle against regressions and give
the GPU tests an oracle-free comparison target; they do NOT pin the oracle to the reference — tests/golden/make_ref_golden.py
and its ref_*.npz fixtures (outputs of the reference's own sources) do that.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lili_om_amd import synth          # noqa: E402
import lili_om_amd as L                # noqa: E402
from oracle import oracle as O         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def s2m_case(variant):
    room = synth.make_room(seed=31, n_query=700, n_edge_query=80)
    P, PO = L.make_params(variant), O.params(variant)
    if variant == "frontend":
        tb, qb = room["t_true"], room["q_true"]
    else:
        tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(9), 0.03, 0.2)
    Q2, T2 = (q0, t0) if variant == "frontend" else L.api.assoc_transform(t0, q0, P)
    tree, etree = O.KdTree(room["map_xyz"]), O.KdTree(room["edge_map_xyz"])
    refl = variant == "livox"
    rs = O.associate_surf(tree, room["map_refl"] if refl else None, room["q_xyz"], room["q_refl"] if refl else None, Q2, T2, PO)
    re_ = O.associate_edge(etree, room["eq_xyz"], Q2, T2, PO)
    ss = (1000.0, max(rs["count"], 1)) if variant == "rot" else 1.0
    se = (200.0, max(re_["count"], 1)) if variant == "rot" else 1.0
    Gs, cs, _ = O.linearize_surf(rs, t0, q0, PO, ss)
    Ge, ce, _ = O.linearize_edge(re_, t0, q0, PO, se)
    out = dict(t0=t0, q0=q0, Q2=np.asarray(Q2), T2=np.asarray(T2), surf_valid=rs["valid"], surf_nn=rs["nn_idx"], surf_d2=rs["nn_d2"],
               surf_n=rs["n"], surf_d=rs["d"], surf_score=rs["score"], edge_valid=re_["valid"], edge_a=re_["a"], edge_b=re_["b"],
               gram_surf=Gs, cost_surf=cs, gram_edge=Ge, cost_edge=ce)
    # three device-style outer iterations (surf only)
    t, q = t0.copy(), q0.copy()
    for _ in range(3):
        Q2i, T2i = (q, t) if variant == "frontend" else L.api.assoc_transform(t, q, P)
        r = O.associate_surf(tree, room["map_refl"] if refl else None, room["q_xyz"], room["q_refl"] if refl else None, Q2i, T2i, PO)
        G, _, _ = O.linearize_surf(r, t, q, PO, (1000.0, max(r["count"], 1)) if variant == "rot" else 1.0)
        st, t, q, _ = O.gn_step(G, t, q)
        assert st == 0
    out.update(t3=t, q3=q)
    return out


def main():
    for v in ("rot", "livox", "frontend"):
        np.savez_compressed(os.path.join(HERE, f"s2m_{v}.npz"), **s2m_case(v))
    w = synth.make_workload(n_map=100_000, n_az=120, half_extent=(150.0, 150.0))
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 7.0, np.float32)], 1).astype(np.float32)
    q_imu = [0.99995, 0.004, -0.006, 0.005]
    q_lb = [0.7071, 0.0, 0.0, 0.7071]
    r = O.extract_rot(raw, q_imu, q_lb, O.rot_params(ds_rate=2, atan_mode=2, stable_sort=1))   # glibc float atan / atan2 restated = the HIP default
    np.savez_compressed(os.path.join(HERE, "extract_rot.npz"), q_imu=q_imu, q_lb=q_lb, n_in=raw.shape[0], full_src=r["full_src"],
                        ring_start=r["ring_start"], ring_end=r["ring_end"], label=r["label"].astype(np.int8), edge_idx=r["edge_idx"],
                        flat_idx=r["flat_idx"], lessflat_idx=r["lessflat_idx"], surf=r["surf"], surf_cnt=r["surf_cnt"],
                        curv_sum=np.float64(r["curvature"].astype(np.float64).sum()), full_xyz_sum=r["full"].astype(np.float64).sum(0))
    s = synth.make_livox_scan(7)
    q_imu2 = [0.99998, -0.003, 0.004, 0.002]
    rl = O.extract_livox(s, q_imu2)
    np.savez_compressed(os.path.join(HERE, "extract_livox.npz"), q_imu=q_imu2, n_in=s.shape[0], cut_src=rl["cut_src"], cell_src=rl["cell_src"],
                        edge_cell=rl["edge_cell"], surf_cell=rl["surf_cell"], edge=rl["edge"], surf_sum=rl["surf"].astype(np.float64).sum(0))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
