"""One size, one lanes-per-query setting, both flavours: 400 outer iterations each (bench restart schedule), for rocprofv3 --kernel-trace --stats.
    python tools/coop_profile.py <n_queries> <lanes> [n_map]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

n = int(sys.argv[1]); lanes = int(sys.argv[2]); n_map = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
w = synth.make_workload(n_map=n_map, half_extent=(460.0, 380.0) if n_map >= 4_000_000 else (150.0, 150.0))
order = np.argsort(w["scan_ring"], kind="stable")
scan = np.ascontiguousarray(w["scan_xyz"][order])
q = scan[:n] if n >= 25000 else np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
ctx.set_option("assoc_lpq", lanes)
focus_r = float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0
for flavour in ("rot", "frontend"):
    P = L.make_params(flavour)
    m = L.ScanToMapMatcher(ctx, P)
    m.map_focus(w["lidar_t"], focus_r)
    m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
    if flavour == "rot":
        tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    else:
        tb, qb = np.asarray(w["lidar_t"], np.float64), np.array([1.0, 0.0, 0.0, 0.0])
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.3, 2.0)
    m.set_queries(0, L.KIND_SURF, q)
    m.pose_set(1, t0, q0)
    m.iterate_restart(0, 400, 10, 1, L.MASK_SURF)
    print(flavour, m.pose_get(0)[2], file=sys.stderr)
ctx.close()
