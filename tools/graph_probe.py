#!/usr/bin/env python
"""Would a hipGraph of the outer iterations beat the eager launches?  (VERDICT r3 #2.)  The C loop's launches on a torch stream are captured with
torch.cuda.CUDAGraph (hipStreamBeginCapture on the same stream; lili_s2m_iterate enqueues only kernels once its buffers exist) and replayed;
eager = lili_s2m_iterate_restart as the bench runs it.  us per outer iteration, ROT and front-end flavours, 2 k / 20 k / 200 k queries."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import lili_om_amd as L
from lili_om_amd import synth
import bench

w = synth.make_workload(n_map=5_000_000, n_az=3125, half_extent=(460.0, 380.0))
scan = bench.ring_major(w["scan_xyz"], w["scan_ring"])
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
ctx = L.Context(0, stream=ts.cuda_stream)
rows = []
for flavour in ("rot", "frontend"):
    P = L.make_params(flavour)
    m = L.ScanToMapMatcher(ctx, P)
    m.map_focus(w["lidar_t"], float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0)
    m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
    tb, qb = (L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P) if flavour == "rot" else (np.asarray(w["lidar_t"], np.float64), np.array([1.0, 0, 0, 0])))
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
    m.pose_set(1, t0, q0)
    for n in (2000, 20000, 200000):
        q = scan[:n] if n >= 25000 else np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
        m.set_queries(0, L.KIND_SURF, q)
        m.iterate_restart(0, 20, 10, 1, L.MASK_SURF)
        torch.cuda.synchronize()
        tic = time.perf_counter()
        m.iterate_restart(0, 200, 10, 1, L.MASK_SURF)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - tic) / 200 * 1e6
        pe = m.pose_get(0)
        row = {"flavour": flavour, "queries": int(q.shape[0]), "eager_us_per_iteration": round(eager, 2)}
        try:
            g = torch.cuda.CUDAGraph()
            m.pose_copy(0, 1)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=ts):
                m.iterate_restart(0, 10, 10, 1, L.MASK_SURF)
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize()
            tic = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            row["graph_us_per_iteration"] = round((time.perf_counter() - tic) / 200 * 1e6, 2)
            pg = m.pose_get(0)
            row["same_final_pose"] = bool(np.array_equal(pe[0], pg[0]) and np.array_equal(pe[1], pg[1]))
        except Exception as e:      # noqa: BLE001
            row["graph_error"] = repr(e)[:300]
        rows.append(row)
        print(json.dumps(row), flush=True)
ctx.close()
