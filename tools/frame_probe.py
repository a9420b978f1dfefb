"""configs[1]'s frame loop (100 Livox frames, one lili_frontend_frame per frame) with lili_set_option pairs from the command line: ms per frame (median of 3 passes),
the call's own stage stamps, filter / local-map statistics.  usage: python tools/frame_probe.py [name=value ...]"""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L
from lili_om_amd import synth
import bench_configs as BC

opts = dict(a.split("=") for a in sys.argv[1:])
n_frames = int(opts.pop("frames", 100))
frames = [synth.make_livox_scan(100 + f, origin=BC._circuit(f)[0], yaw=BC._circuit(f)[2], inject_bad=False) for f in range(n_frames)]
ctx = L.Context(0)
for k, v in opts.items():
    ctx.set_option(k, int(v))
P = L.make_params("frontend")
pins = []
for fr in frames:
    p = L.api.PinnedArray(fr.shape, np.float32); p.array[...] = fr; pins.append(p)
odo = L.FrontendOdometry(ctx, P, leaf_query=0.4, leaf_map=0.4, width=20, scan_match_cnt=6, first_match_cnt=12, reference_startup=False)


def predict(poses):
    if len(poses) == 1:
        return poses[-1]
    (ta, qa), (tb, qb) = poses[-2], poses[-1]
    qi = qa * np.array([1, -1, -1, -1]) / np.dot(qa, qa)
    dq = synth.quat_mul(qi, qb)
    q0 = synth.quat_mul(qb, dq)
    return tb + synth.quat_rot(qb, synth.quat_rot(qi, tb - ta)), q0 / np.linalg.norm(q0)


def run():
    odo.reset()
    poses, st = [], np.zeros(4)
    for f in range(n_frames):
        t0, q0 = BC._circuit(0)[:2] if f == 0 else predict(poses)
        t, q, info = odo.frame(pins[f].array, t0, q0, timing=True)
        poses.append((t, q))
        su = info["stage_us"]
        st += [su[0], su[1] - su[0], su[2] - su[1], su[3] - su[2]]
    return poses, st / n_frames


run()
passes = []
for _ in range(3):
    torch.cuda.synchronize(); tic = time.perf_counter()
    poses, st = run()
    torch.cuda.synchronize()
    passes.append((time.perf_counter() - tic) / n_frames * 1e3)
print(json.dumps({"options": opts, "ms_per_frame": round(float(np.median(passes)), 4), "passes": [round(p, 4) for p in passes], "stage_us": [round(float(x), 1) for x in st],
                  "voxel_filter_stats": L.api.voxel_filter_stats(ctx), "final_t": [round(float(x), 6) for x in poses[-1][0]]}))
ctx.close()
