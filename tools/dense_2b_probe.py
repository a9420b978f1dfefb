"""configs[2] variant B (5 M-point map voxelised at 0.05 m, 200 k queries) launch by launch: for every outer iteration of one registration the
share of queries the fine index cannot settle (5th neighbour not strictly inside its covered radius), the association time at exactly that pose
(HIP events, 5 launches back to back) and the correspondence count; then the index build, call by call.  VERDICT r5 #1: measure before fixing.
usage: python tools/dense_2b_probe.py [name=value ...]     (lili_set_option pairs; `ips=N` sets the iterations of the registration)"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L
from lili_om_amd import synth
import bench_configs as BC

opts = dict(a.split("=") for a in sys.argv[1:])
ips = int(opts.pop("ips", 10))
quick = int(opts.pop("quick", 0))
scan_order = int(opts.pop("scan_order", 0))
mp, q_local, t_true, q_true = BC.make_variant_b(scan_order=bool(scan_order))
P = L.make_params("rot")
tb, qb = L.api.body_pose_from_lidar(t_true, q_true, P)
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.1, 0.5)
ctx = L.Context(0)
for k, v in opts.items():
    ctx.set_option(k, int(v))
m = L.ScanToMapMatcher(ctx, P)
m.map_focus(None)
d_map = torch.from_numpy(mp).cuda()
cloud = L.api.cloud_from_device(d_map.data_ptr(), mp.shape[0], 12, -1)
m.set_input_cloud(L.KIND_SURF, cloud)
torch.cuda.synchronize()
builds = []
for _ in range(4):
    tic = time.perf_counter(); m.set_input_cloud(L.KIND_SURF, cloud); torch.cuda.synchronize(); builds.append(round((time.perf_counter() - tic) * 1e3, 3))
occ, fine_cell, fine_r2 = m.map_density(L.KIND_SURF)
print(json.dumps({"map_points": int(mp.shape[0]), "index_build_ms": builds, "occupancy": round(occ, 1), "fine_cell": fine_cell, "fine_sq_radius": fine_r2,
                  "cells": m.map_info(L.KIND_SURF)}), flush=True)
n_q = q_local.shape[0]
m.set_queries(0, L.KIND_SURF, q_local)
if quick:      # wall time of back-to-back association launches at the true pose (all settled) and at the perturbed start (experiment builds: records may be wrong)
    for name, (tt, qq) in (("true_pose", (tb, qb)), ("perturbed_start", (t0, q0))):
        Q2, T2 = L.api.assoc_transform(tt, qq, P)
        for _ in range(3):
            m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
        ctx.sync(); tic = time.perf_counter()
        for _ in range(quick):
            m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
        ctx.sync()
        print(json.dumps({"pose": name, "association_us_wall": round((time.perf_counter() - tic) / quick * 1e6, 2), "launches": quick}), flush=True)
    ctx.close()
    sys.exit(0)
m.pose_set(0, t0, q0)
rows = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(ips):
    tl, ql, _ = m.pose_get(0)
    Q2, T2 = L.api.assoc_transform(tl, ql, P)
    ctx.set_debug(True)
    n_corr = m.find_corresponding_surf_features(0, Q2, T2)
    idx, d2 = m.neighbors(0, L.KIND_SURF, n_q)
    ctx.set_debug(False)
    unsettled = (idx[:, 4] < 0) | ~(d2[:, 4] < np.float32(fine_r2))
    per_wave = unsettled.reshape(-1, 64).any(1).mean() if n_q % 64 == 0 else float("nan")
    m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
    e1.record(); torch.cuda.synchronize()
    rows.append({"iteration": it, "association_us": round(e0.elapsed_time(e1) * 1e3 / 5, 2), "unsettled_by_fine_index": round(float(unsettled.mean()), 5),
                 "waves_with_an_unsettled_lane": round(float(per_wave), 4), "no_5nn_inside_gate": round(float((idx[:, 4] < 0).mean()), 5), "correspondences": int(n_corr)})
    print(json.dumps(rows[-1]), flush=True)
    m.iterate(0, 1, L.MASK_SURF)
tg, qg, st = m.pose_get(0)
print(json.dumps({"mean_association_us": round(float(np.mean([r["association_us"] for r in rows])), 2), "gn_status": int(st),
                  "dt_truth_m": float(np.abs(tg - tb).max())}), flush=True)
# whole registrations, as bench_configs.config2b times them
m.pose_set(1, t0, q0)
m.iterate_restart(0, 2 * ips, ips, 1, L.MASK_SURF)
torch.cuda.synchronize(); tic = time.perf_counter()
m.iterate_restart(0, 10 * ips, ips, 1, L.MASK_SURF)
torch.cuda.synchronize()
print(json.dumps({"us_per_iteration": round((time.perf_counter() - tic) / (10 * ips) * 1e6, 2)}), flush=True)
ctx.close()
