#!/usr/bin/env python
"""Verified neighbour cache on the bench workload: per outer iteration of one registration, how many queries hold a usable record (margin > 0), how many
waves consist of such queries only, and how many records survived the launch unchanged (= served without a search)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lili_om_amd as L
from lili_om_amd import synth
import bench

w = synth.make_workload(n_map=5_000_000, n_az=3125, half_extent=(460.0, 380.0))
P = L.make_params("rot")
scan = bench.ring_major(w["scan_xyz"], w["scan_ring"])
ctx = L.Context(0)
m = L.ScanToMapMatcher(ctx, P)
m.map_focus(w["lidar_t"], float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0)
m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
m.set_queries(0, L.KIND_SURF, scan)
tb, qb = bench.body_pose_for_lidar(L, P, w["lidar_t"])
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
n = scan.shape[0]
for reg in range(2):
    m.pose_set(0, t0, q0)
    prev = None
    for it in range(10):
        m.iterate(0, 1, L.MASK_SURF)
        try:
            rec, tail = m.nn_cache_records(0, L.KIND_SURF, n, with_tail=True)
        except Exception as e:
            print("iteration", it, "no cache yet:", e); continue
        usable = rec[:, 3] > 0
        nw = n // 64
        row = {"registration": reg, "iteration": it + 1, "usable_lanes": round(float(usable.mean()), 5), "waves_all_usable": round(float(usable[:nw * 64].reshape(nw, 64).all(1).mean()), 4),
               "margin_median_m": round(float(np.median(rec[usable, 3])), 5) if usable.any() else None}
        mo = tail[:, 2].copy().view(np.float32)
        has_fit, fit_ok = (tail[:, 3] & 256) != 0, (tail[:, 3] & 512) != 0
        row.update(has_fit=round(float(has_fit.mean()), 5), fit_ok=round(float(fit_ok.mean()), 5), six=round(float(((tail[:, 3] & 255) == 6).mean()), 4),
                   fit_margin_gt_1e_6=round(float((mo > 1e-6).mean()), 5), fit_margin_median=float(np.median(mo[mo > 0])) if (mo > 0).any() else None,
                   waves_all_fit_margin=round(float((mo > 1e-6)[:nw * 64].reshape(nw, 64).all(1).mean()), 4),
                   lanes_without_fit_margin_but_usable=int((usable & ~(mo > 1e-6)).sum()))
        if prev is not None:
            same = (rec == prev).all(1)
            row["lanes_served"] = round(float(same.mean()), 5)
            row["waves_served"] = round(float(same[:nw * 64].reshape(nw, 64).all(1).mean()), 4)
        prev = rec
        print(json.dumps(row), flush=True)
ctx.close()
