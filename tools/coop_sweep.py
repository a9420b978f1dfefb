"""Association / iteration time by launch size and lanes per query, on the bench workload (5 M-point map, variant A).

    python tools/coop_sweep.py [out.json] [--n-map N]

For every size (a contiguous ring-major piece of the 200 k-point scan, i.e. what a rank's shard or a down-sampled keyframe looks like) and
every lanes-per-query setting (1 = the one-lane kernels, 0 = the library's choice):
  assoc_us          one association launch at the converged pose, HIP events over back-to-back launches
  assoc_first_us    the same at the perturbed pose (0.3 m / 2 deg off: a good part of the queries walks the shell)
  rot_iter_us       one outer iteration of the ROT back-end flavour (count scaling: associate, linearise, reduce + GN), restart schedule of bench.py
  front_iter_us     one outer iteration of the front-end flavour (no count scaling: associate + linearise in one launch, reduce + GN)
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None
    n_map = 5_000_000
    if "--n-map" in sys.argv:
        n_map = int(sys.argv[sys.argv.index("--n-map") + 1])
    w = synth.make_workload(n_map=n_map, half_extent=(460.0, 380.0) if n_map >= 4_000_000 else (150.0, 150.0))
    order = np.argsort(w["scan_ring"], kind="stable")
    scan = np.ascontiguousarray(w["scan_xyz"][order])
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = L.Context(0, stream=s.cuda_stream)
    res = {"n_map": n_map, "rows": []}
    sizes = [2000, 5000, 10000, 20000, 25000, 50000, 100000, 200000]
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
        for n in sizes:
            # a contiguous ring-major piece (whole rings first) for shard-like sizes, a strided sample for keyframe-like sizes
            q = scan[:n] if n >= 25000 else np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
            m.set_queries(0, L.KIND_SURF, q)
            for lanes in (1, 2, 4, 8, 16, 0):
                if lanes > 1 and n * lanes > 1_700_000:
                    continue
                ctx.set_option("assoc_lpq", lanes)
                row = {"flavour": flavour, "n": int(q.shape[0]), "lanes": lanes}

                def assoc_us(t, qq, reps=60):
                    Q2, T2 = L.api.assoc_transform(t, qq, P)
                    for _ in range(5):
                        m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(reps):
                        m.find_corresponding_surf_features(0, Q2, T2, want_count=False)
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) * 1e3 / reps
                m.pose_set(1, t0, q0)
                m.pose_copy(0, 1)
                m.iterate(0, 10, L.MASK_SURF)
                tc, qc, st = m.pose_get(0)
                row["gn_status"] = int(st)
                row["dt_truth_m"] = float(np.linalg.norm(tc - tb))
                row["assoc_us"] = round(assoc_us(tc, qc), 3)
                row["assoc_first_us"] = round(assoc_us(t0, q0), 3)
                m.iterate_restart(0, 20, 10, 1, L.MASK_SURF)
                torch.cuda.synchronize()
                tic = time.perf_counter()
                m.iterate_restart(0, 200, 10, 1, L.MASK_SURF)
                torch.cuda.synchronize()
                row["iter_us"] = round((time.perf_counter() - tic) / 200 * 1e6, 3)
                res["rows"].append(row)
                print(json.dumps(row), flush=True)
        ctx.set_option("assoc_lpq", 0)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
