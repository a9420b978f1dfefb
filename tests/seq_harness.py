"""Test harness (not product code): a minimal scan-to-local-map odometry loop in the style of
LidarOdometry::run (L/src/LidarOdometry.cpp:652-700) used to compare the GPU path with the oracle over a SEQUENCE
of synthetic Livox-like frames: extract features -> voxel-filter -> match against the local map of the last
`map_width` frames -> K outer Gauss-Newton iterations (re-association every iteration) -> next frame with a
constant-velocity prediction.  Map assembly and voxel filtering are host glue here (numpy), identical for both
sides; the extractor and the matcher are the systems under test."""
import math

import numpy as np

from lili_om_amd import synth


def gt_pose(f, lane_y=0.0):
    """lane_y = 0: the run the committed fixtures were generated on (short runs only — at x = 25 m it meets the pole at the lattice
    corner); the 100-frame run of BASELINE configs[1] drives 3.5 m beside the pole row."""
    yaw = 0.02 * f
    t = np.array([0.5 * f, lane_y + 0.3 * math.sin(0.2 * f), 1.8])    # 5 m/s at 10 Hz: the first step stays inside the 1 m gate
    q = np.array([math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)])
    return t, q, yaw


def gt_pose_circuit(f, radius=4.0, step=0.03):
    """100-frame trajectory of BASELINE configs[1] (substitute for FR_IOSB_Short, no rosbag offline): a circuit of `radius` m around
    the crossing at the origin, heading along the tangent.  The straight run of gt_pose is only good for short sequences: a street
    canyon is invariant along its axis, plain scan-to-map odometry cannot observe the motion there and after some tens of frames
    settles on "standing still"; on the circuit the 80 deg field of view always holds walls of both orientations."""
    a = step * f
    yaw = a + math.pi / 2
    t = np.array([radius * math.cos(a), radius * math.sin(a), 1.8])
    q = np.array([math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)])
    return t, q, yaw


def make_frames(n_frames, seed=100, gt=gt_pose):
    frames = []
    for f in range(n_frames):
        t, q, yaw = gt(f)
        frames.append(synth.make_livox_scan(seed + f, origin=t, yaw=yaw, inject_bad=False))
    return frames


def voxel(xyz_c, leaf=0.4):
    """host voxel filter on (x,y,z,curvature) rows; ascending voxel id, f64 centroids rounded to f32"""
    if xyz_c.shape[0] == 0:
        return xyz_c
    ijk = np.floor(xyz_c[:, :3].astype(np.float64) / leaf).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    uniq, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv)
    out = np.stack([np.bincount(inv, weights=xyz_c[:, k].astype(np.float64)) / cnt for k in range(xyz_c.shape[1])], 1)
    return out.astype(np.float32)


def to_world(xyz_c, t, q):
    w = synth.quat_rot(q, xyz_c[:, :3].astype(np.float64)) + t
    return np.concatenate([w, xyz_c[:, 3:4]], 1).astype(np.float32)


def run_sequence(frames, extract_fn, match_fn, n_outer=6, map_width=20, gt=gt_pose):
    """extract_fn(scan) -> surf features (n,8); match_fn(map_xyzc, query_xyzc, t0, q0, n_outer) -> (t, q).
    Returns the list of estimated poses."""
    poses, kept = [], []
    t_prev = q_prev = None
    for f, scan in enumerate(frames):
        surf = extract_fn(scan)
        qry = voxel(np.ascontiguousarray(surf[:, [0, 1, 2, 7]]))
        if f == 0:
            t, q, _ = gt(0)
        else:
            if f == 1:
                t0, q0 = poses[-1]
            else:   # constant-velocity prediction (poseInitialization, L/src/LidarOdometry.cpp:415-480)
                (ta, qa), (tb, qb) = poses[-2], poses[-1]
                qa_inv = qa * np.array([1, -1, -1, -1]) / np.dot(qa, qa)    # Eigen's inverse(): conjugate / squared norm
                dq = synth.quat_mul(qa_inv, qb)
                dt = synth.quat_rot(qa_inv, tb - ta)
                q0 = synth.quat_mul(qb, dq)
                q0 = q0 / np.linalg.norm(q0)     # without this the norm error TRIPLES per frame (|q0| = |qb|^2 / |qa|) and wrecks runs > ~35 frames
                t0 = tb + synth.quat_rot(qb, dt)
            local = np.concatenate([to_world(k, tp, qp) for (k, (tp, qp)) in zip(kept[-map_width:], poses[-map_width:])], 0)
            local = voxel(local)
            t, q = match_fn(local, qry, t0, q0, 12 if f == 1 else n_outer)   # the reference also iterates 8x on the first frames (L:501-502)
        poses.append((np.asarray(t, np.float64), np.asarray(q, np.float64)))
        kept.append(qry)
    return poses


def ate(poses, gt=gt_pose):
    err = [np.linalg.norm(p[0] - gt(f)[0]) for f, p in enumerate(poses)]
    return float(np.sqrt(np.mean(np.square(err)))), float(np.max(err))
