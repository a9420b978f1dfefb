"""Time per OUTER iteration (associate + linearise + sum + Gauss-Newton step) of a small scan: the persistent launch (k_iterate_coop, option
persistent_iterate) against the launch-per-stage loop.    python tools/iter_time.py [out.json]"""
import json
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

w = synth.make_workload(n_map=5_000_000, half_extent=(460.0, 380.0))
order = np.argsort(w["scan_ring"], kind="stable")
scan = np.ascontiguousarray(w["scan_xyz"][order])
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
out = []
for flavour in ("rot", "frontend"):
    P = L.make_params(flavour)
    m = L.ScanToMapMatcher(ctx, P)
    m.map_focus(w["lidar_t"], float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0)
    m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.05, 0.5)
    m.pose_set(7, t0, q0)
    for n in (500, 1000, 2000, 4000, 8000, 12000):
        q = np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
        m.set_queries(0, L.KIND_SURF, q)
        row = dict(flavour=flavour, n=n)
        poses = {}
        for mode in (0, 1):
            ctx.set_option("persistent_iterate", mode)
            iters, reps = 10, 30
            def run():
                m.pose_copy(0, 7); m.iterate(0, iters, L.MASK_SURF)
            run(); torch.cuda.synchronize()
            poses[mode] = m.pose_get(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            row["persistent_us_per_iteration" if mode else "launches_us_per_iteration"] = round(e0.elapsed_time(e1) * 1e3 / reps / iters, 3)
        row["pose_diff"] = float(max(np.abs(poses[0][0] - poses[1][0]).max(), np.abs(poses[0][1] - poses[1][1]).max()))
        row["status"] = [int(poses[0][2]), int(poses[1][2])]
        out.append(row)
        print(json.dumps(row), flush=True)
ctx.set_option("persistent_iterate", 0)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
ctx.close()
