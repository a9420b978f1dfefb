#!/usr/bin/env python
"""Generates tests/golden/ref_*.npz by running the REFERENCE'S OWN sources (oracle/_ref, built from /root/reference by
oracle/refshim/Makefile against stand-in third-party headers) on fixed-seed synthetic inputs.  These fixtures — unlike
tests/golden/{s2m,extract}_*.npz, which come from the oracle — pin the oracle and the HIP path to code of the reference:

  ref_rot.npz      LiLi-OM-ROT/src/Preprocessing.cpp driven with 4 clouds + a 200 Hz gyro stream: the three published
                   clouds of the 2 processed scans (/lidar_cloud_cutted, /edge_features, /surf_features)
  ref_livox.npz    LiLi-OM/src/Preprocessing.cpp, same protocol (large clouds stored as sha256 + every 8th row)
  ref_frontend.npz LiLi-OM/src/LidarOdometry.cpp (the whole front-end node) fed with the Livox node's output for 6 frames of a
                   moving synthetic sequence; ceres::Solve = the documented one-GN-step stand-in (oracle/refshim/ref_lo.cpp):
                   per solve the pose in/out and a hash of the residual-block records the reference built, per frame the
                   node's abs/rel pose and keyframe flag; two solves are stored completely (map, queries, records, raw rows)
  ref_backend.npz  BackendFusion.cpp's transformPoint / findCorrespondingCornerFeatures / findCorrespondingSurfFeatures (both
                   flavours; member-function text sliced out of the file at build time, oracle/refshim/ref_backend.cpp) on a
                   room scene: the correspondence records and the residual blocks (raw r, dr/dt, dr/dq) created from them
  ref_format.npz   LiLi-OM/src/FormatConvert.cpp (livoxLidarHandler): 5 000 random livox CustomPoints -> the published cloud
                   (hash + every 16th row), plus the all-zero offset_time case (0/0 and x/0)
  ref_marg.npz     ResidualBlockInfo::Evaluate + ThreadsConstructA (MarginalizationFactor.cpp:3-71, text sliced at build time)
                   over the lidar blocks of ref_backend.npz's Livox keyframe: robustified rows, dense A and b
  ref_factors.npz  LidarEdgeFactor / LidarPlaneNormFactor / LidarPlaneNormIncreFactor ::Create()->Evaluate() on random
                   records: residual + both Jacobian blocks

  ref_cfg2.npz     the reference's OTHER configurations: the ROT node with the 32-ring table at ds_rate 2 and the identity
                   extrinsic (LiLi-OM-ROT/config/config_utbm.yaml:13-14,37-40) on HDL-32E-like scans; the ROT back-end association at
                   kd_max_radius 1.5 with utbm's extrinsic (config_utbm.yaml:34-44) on a sparser map (5th neighbours between 1.0 and 1.5 m^2);
                   the Livox node at surf_thres 0.17 and the front-end node at scan_match_cnt 2 / max_num_iter 15 and the Livox back-end
                   association with ka_urban_campus' constants (LiLi-OM/config/config_ka_urban_campus.yaml:5,9-10,17-19,29-36)

Only runs where /root/reference exists (the build container).  Run from the repository root:
    python tests/golden/make_ref_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lili_om_amd import synth          # noqa: E402
from oracle import ref as R            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

ROT_QLB = [0.7071, 0.0, 0.0, 0.7071]                     # R/config/config_fr_iosb.yaml:38-41
ROT_PARAMS = {"/preprocessing/lidar_topic": "/velodyne_points", "/preprocessing/line_num": 64, "/preprocessing/ds_rate": 4,
              "/common/frame_id": "lili_om_rot", "/backend_fusion/imu_topic": "/imu/data",
              "/backend_fusion/ql2b_w": ROT_QLB[0], "/backend_fusion/ql2b_x": ROT_QLB[1],
              "/backend_fusion/ql2b_y": ROT_QLB[2], "/backend_fusion/ql2b_z": ROT_QLB[3]}
LIVOX_PARAMS = {"/preprocessing/surf_thres": 0.28, "/preprocessing/edge_thres": 4.0, "/common/frame_id": "lili_om"}
N_SCANS_IN = 4          # the node holds back two clouds (cloudHandler's queue): 4 in -> 2 processed


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rot_inputs():
    scans = []
    for s in range(N_SCANS_IN):
        w = synth.make_workload(n_map=200_000, n_az=200, half_extent=(150.0, 150.0), seed=synth.SEED_SCENE + 40 + s)
        refl = np.random.default_rng(40 + s).integers(1, 255, w["scan_xyz"].shape[0]).astype(np.float32)
        scans.append(np.concatenate([w["scan_xyz"], refl[:, None]], 1).astype(np.float32))
    stamps = 100.0 + 0.1 * np.arange(N_SCANS_IN)
    imu_t = 99.95 + 0.005 * np.arange(100)
    gyr = 0.2 * np.random.default_rng(5).standard_normal((100, 3)) + np.array([0.1, -0.05, 0.3])
    return scans, stamps, imu_t, gyr


def livox_inputs():
    scans = [synth.make_livox_scan(20 + s) for s in range(N_SCANS_IN)]
    stamps = 50.0 + 0.1 * np.arange(N_SCANS_IN)
    imu_t = 49.97 + 0.005 * np.arange(100)
    gyr = 0.2 * np.random.default_rng(6).standard_normal((100, 3)) + np.array([0.1, -0.05, 0.3])
    return scans, stamps, imu_t, gyr


def factor_inputs(n=64, seed=77):
    rng = np.random.default_rng(seed)
    u = lambda *s: rng.uniform(-1.0, 1.0, s)                                           # noqa: E731
    q = u(n, 4) + np.array([2.0, 0, 0, 0])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    nrm = u(n, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return dict(cp=(20 * u(n, 3)).astype(np.float32).astype(np.float64), a=(20 * u(n, 3)).astype(np.float32).astype(np.float64),
                b=(20 * u(n, 3)).astype(np.float32).astype(np.float64), n=nrm.astype(np.float32).astype(np.float64),
                d=u(n).astype(np.float32).astype(np.float64), s=rng.uniform(0.1, 20.0, n), t=3 * u(n, 3), q=q,
                qlb=np.array(ROT_QLB), tlb=np.array([-0.18, 0.0, -0.095]))


PAYLOAD_ROT = [0, 1, 2, 4]                       # x y z intensity of the 32-byte PointXYZI
PAYLOAD_LIVOX = [0, 1, 2, 4, 5, 6, 8, 9]         # x y z | normal | intensity curvature of the 48-byte PointXYZINormal


def run_rot(params=None, inputs=None):
    scans, stamps, imu_t, gyr = (inputs or rot_inputs)()
    out = R.run_scans("rot", params or ROT_PARAMS, scans, stamps, imu_t, gyr)
    d = dict(n_processed=len(out))
    for k, o in enumerate(out):
        d[f"stamp{k}"] = o["stamp"]
        for name in ("cutted", "edge", "surf"):
            d[f"{name}{k}"] = o[name][:, PAYLOAD_ROT]
        # Feature INDICES of the reference run (north star: "feature indices bit-exact").  The node publishes clouds, not indices; every
        # /edge_features point is a copy of one /lidar_cloud_cutted point (R/src/Preprocessing.cpp:422-427), and a full-cloud row is unique
        # (intensity = ring + 0.1 * relTime), so the row match IS the index the reference pushed.
        full = np.ascontiguousarray(d[f"cutted{k}"], np.float32).view(np.uint32)
        key = {row.tobytes(): i for i, row in enumerate(full)}
        assert len(key) == full.shape[0], "duplicate rows in the reference's full cloud"
        edge = np.ascontiguousarray(d[f"edge{k}"], np.float32).view(np.uint32)
        d[f"edge_src{k}"] = np.array([key[row.tobytes()] for row in edge], np.int32)
    return d


def run_livox(params=None):
    scans, stamps, imu_t, gyr = livox_inputs()
    out = R.run_scans("livox", params or LIVOX_PARAMS, scans, stamps, imu_t, gyr)
    d = dict(n_processed=len(out))
    for k, o in enumerate(out):
        d[f"stamp{k}"] = o["stamp"]
        d[f"edge{k}"] = o["edge"][:, PAYLOAD_LIVOX]
        for name in ("cutted", "surf"):
            a = o[name][:, PAYLOAD_LIVOX]
            d[f"{name}{k}_n"] = a.shape[0]
            d[f"{name}{k}_sha_payload"] = sha(a[:, [0, 1, 2, 6, 7]])        # x y z intensity curvature
            d[f"{name}{k}_sha_absn"] = sha(np.abs(a[:, 3:6]))               # stored normal / direction, sign-free (Eigen's sign is arbitrary)
            d[f"{name}{k}_every8"] = a[::8]
    return d


FRONTEND_PARAMS = {"/common/frame_id": "lili_om", "/lidar_odometry/if_to_deskew": 0, "/lidar_odometry/max_num_iter": 12,
                   "/lidar_odometry/scan_match_cnt": 6}
FRONTEND_FRAMES = 6
FRONTEND_FULL_SOLVES = (0, 8, 20)      # first solve of frame 1 (8 re-associations there, L/src/LidarOdometry.cpp:498-502), frames 2 and 4


def frontend_inputs():
    """Moving Livox-like sequence (tests/seq_harness.py) + zero-rate gyro stream (the node waits for IMU data)."""
    sys.path.insert(0, ROOT)
    from tests import seq_harness as H
    n = FRONTEND_FRAMES + 2
    frames = H.make_frames(n)
    stamps = 10.0 + 0.1 * np.arange(n)
    imu_t = 9.97 + 0.005 * np.arange(20 * n + 20)
    return frames, stamps, imu_t, np.zeros((imu_t.shape[0], 3))


def run_frontend(pre_params=None, lo_params=None, full_solves=None):
    frames, stamps, imu_t, gyr = frontend_inputs()
    pre = R.run_scans("livox", pre_params or LIVOX_PARAMS, frames, stamps, imu_t, gyr)
    lo = R.LidarOdometry(lo_params or FRONTEND_PARAMS)
    d = dict(n_frames=len(pre))
    abs_pose, rel_pose, kf = [], [], []
    for o in pre:
        ap, rp, k = lo.frame(o["stamp"], o["edge"], o["surf"], o["cutted"])
        abs_pose.append(ap); rel_pose.append(rp); kf.append(k)
    S = lo.solves()
    d.update(abs_pose=np.array(abs_pose), rel_pose=np.array(rel_pose), kf=np.array(kf), n_solves=len(S),
             pose_in=np.array([s["pose_in"] for s in S]), pose_out=np.array([s["pose_out"] for s in S]),
             n_blocks=np.array([len(s["records"]) for s in S]), n_map=np.array([len(s["map"]) for s in S]),
             n_queries=np.array([len(s["queries"]) for s in S]), gn_status=np.array([s["gn_status"] for s in S]),
             records_sha=np.array([sha(s["records"]) for s in S]), rows_sha=np.array([sha(s["rows"]) for s in S]),
             map_sha=np.array([sha(s["map"]) for s in S]), queries_sha=np.array([sha(s["queries"]) for s in S]))
    for i in (FRONTEND_FULL_SOLVES if full_solves is None else full_solves):
        for k in ("map", "queries", "records", "rows"):
            d[f"solve{i}_{k}"] = S[i][k]
    odom = [(st, a) for (topic, st, a) in lo.published() if topic == "/odom"]
    d["odom_stamp"] = np.array([st for st, _ in odom]); d["odom"] = np.array([a for _, a in odom])
    lo.close()
    return d


# ---- the ROT package's odometry node (R/src/LidarOdometry.cpp compiled as is: libref_lo_R.so) behind its own Preprocessing node -> ref_frontend_R.npz (VERDICT r5 #3b)
FRONTEND_R_PARAMS = {"/common/frame_id": "lili_om_rot", "/lidar_odometry/if_to_deskew": 0, "/lidar_odometry/max_num_iter": 12,       # R/config/config_fr_iosb.yaml:16-18
                     "/lidar_odometry/scan_match_cnt": 6}
FRONTEND_R_FRAMES = 6


def frontend_rot_inputs():
    """A 64-ring spinning LiDAR driving through the outdoor scene (0.45 m and 0.6 deg per scan), 500 azimuth steps per revolution, 2 cm range noise; zero-rate gyro."""
    sc = synth.OutdoorScene()
    n = FRONTEND_R_FRAMES + 2
    scans = []
    for f in range(n):
        rng = np.random.default_rng(700 + f)
        yaw = 0.0105 * f
        origin = np.array([0.45 * f, 3.5 + 0.2 * np.sin(0.3 * f), 1.8])
        dirs, ring, rel = synth.spinning_rays(500, synth.hdl64_elevations_deg(), az0=0.004 * f)
        c, s_ = np.cos(yaw), np.sin(yaw)
        wd = np.stack([c * dirs[:, 0] - s_ * dirs[:, 1], s_ * dirs[:, 0] + c * dirs[:, 1], dirs[:, 2]], 1)
        t = sc.raycast(origin, wd)
        ok = np.isfinite(t)
        t = t + rng.normal(0, 0.02, t.shape)
        pts = (dirs * t[:, None])[ok].astype(np.float32)
        scans.append(np.concatenate([pts, np.full((pts.shape[0], 1), 30.0, np.float32)], 1).astype(np.float32))
    stamps = 300.0 + 0.1 * np.arange(n)
    imu_t = 299.97 + 0.005 * np.arange(20 * n + 20)
    return scans, stamps, imu_t, np.zeros((imu_t.shape[0], 3))


def run_frontend_rot():
    scans, stamps, imu_t, gyr = frontend_rot_inputs()
    pre = R.run_scans("rot", ROT_PARAMS, scans, stamps, imu_t, gyr)
    lo = R.LidarOdometry(FRONTEND_R_PARAMS, flavour="rot")
    d = dict(n_frames=len(pre))
    abs_pose, rel_pose, kf = [], [], []
    for o in pre:
        ap, rp, k = lo.frame(o["stamp"], o["edge"], o["surf"], o["cutted"])
        abs_pose.append(ap); rel_pose.append(rp); kf.append(k)
    S = lo.solves()
    d.update(abs_pose=np.array(abs_pose), rel_pose=np.array(rel_pose), kf=np.array(kf), n_solves=len(S),
             pose_in=np.array([s["pose_in"] for s in S]), pose_out=np.array([s["pose_out"] for s in S]),
             n_blocks=np.array([len(s["records"]) for s in S]), n_map=np.array([len(s["map"]) for s in S]),
             n_queries=np.array([len(s["queries"]) for s in S]), gn_status=np.array([s["gn_status"] for s in S]),
             records_sha=np.array([sha(s["records"]) for s in S]), n_edge=np.array([o["edge"].shape[0] for o in pre]), n_surf=np.array([o["surf"].shape[0] for o in pre]))
    lo.close()
    return d


BACKEND_PARAMS = {   # L/config/config_fr_iosb.yaml, R/config/config_fr_iosb.yaml: kd_max_radius, surf_dist_thres, lidar_const, reflect_thres, q_lb, t_lb
    "livox": dict(kd_max_radius=1.0, surf_dist_thres=0.12, lidar_const=20.0, reflect_thres=15.0, q_lb=[0.0, 0.0, 0.0, 1.0], t_lb=[-0.0265, 0.0202, 0.05309]),
    "rot": dict(kd_max_radius=1.0, surf_dist_thres=0.12, lidar_const=7.5, reflect_thres=0.0, q_lb=[0.7071, 0.0, 0.0, 0.7071], t_lb=[-0.18, 0.0, -0.095]),
    # R/config/config_utbm.yaml:30-44 (= config_urban_hk.yaml's matcher constants): the gate is 1.5 m^2; the map is voxelised at 0.8 m here so that a good
    # part of the queries has its fifth neighbour BETWEEN 1.0 and 1.5 m^2 (kept at 1.5, dropped at 1.0)
    "rot_utbm": dict(kd_max_radius=1.5, surf_dist_thres=0.12, lidar_const=7.5, reflect_thres=0.0, q_lb=[1.0, 0.0, 0.0, 0.0], t_lb=[0.5, -1.4, -1.5],
                     room=dict(seed=14, leaf=0.8, n_query=1500, n_edge_query=200)),
    # L/config/config_ka_urban_campus.yaml:17-19,29-36
    "livox_ka": dict(kd_max_radius=1.0, surf_dist_thres=0.08, lidar_const=15.0, reflect_thres=15.0, q_lb=[0.0, 0.0, 1.0, 0.0], t_lb=[-0.05, -0.0202, -0.13],
                     room=dict(seed=15, n_query=1500, n_edge_query=200)),
}


# ---- the reference's other configurations -> ref_cfg2.npz
ROT32_QLB = [1.0, 0.0, 0.0, 0.0]                       # R/config/config_utbm.yaml:37-40
ROT32_PARAMS = dict(ROT_PARAMS, **{"/preprocessing/line_num": 32, "/preprocessing/ds_rate": 2, "/backend_fusion/ql2b_w": 1.0, "/backend_fusion/ql2b_x": 0.0,
                                   "/backend_fusion/ql2b_y": 0.0, "/backend_fusion/ql2b_z": 0.0})
LIVOX_KA_PARAMS = {"/preprocessing/surf_thres": 0.17, "/preprocessing/edge_thres": 4.0, "/common/frame_id": "lili_om"}     # config_ka_urban_campus.yaml:5-6
FRONTEND_KA_PARAMS = {"/common/frame_id": "lili_om", "/lidar_odometry/if_to_deskew": 0, "/lidar_odometry/max_num_iter": 15,
                      "/lidar_odometry/scan_match_cnt": 2}                                                                   # config_ka_urban_campus.yaml:9-11


def rot32_inputs():
    """HDL-32E-like scans: 32 rings at the elevations the reference's 32-ring table maps back to their ids (R/src/Preprocessing.cpp:325-331), 700
    azimuth steps, the outdoor scene, and a few points outside the table (id < 0 or > 31: dropped by the node)."""
    sc = synth.OutdoorScene()
    scans = []
    for s in range(N_SCANS_IN):
        rng = np.random.default_rng(140 + s)
        dirs, ring, rel = synth.spinning_rays(700, synth.hdl32_elevations_deg(), az0=0.013 * s)
        t = sc.raycast(np.array([0.4 * s, 0.1 * s, 1.8]), dirs)
        ok = np.isfinite(t)
        t = t + rng.normal(0, 0.02, t.shape)
        pts = (dirs * t[:, None])[ok].astype(np.float32)
        refl = rng.integers(1, 255, pts.shape[0]).astype(np.float32)
        raw = np.concatenate([pts, refl[:, None]], 1).astype(np.float32)
        bad = rng.choice(raw.shape[0], 60, replace=False)
        raw[bad[:30], 2] = np.abs(raw[bad[:30], 2]) + 0.35 * np.linalg.norm(raw[bad[:30], :2], axis=1)      # above +10.67 deg: id > 31
        raw[bad[30:], 2] = -np.abs(raw[bad[30:], 2]) - 0.8 * np.linalg.norm(raw[bad[30:], :2], axis=1)      # below -30.67 deg: id < 0
        scans.append(raw)
    stamps = 200.0 + 0.1 * np.arange(N_SCANS_IN)
    imu_t = 199.95 + 0.005 * np.arange(100)
    gyr = 0.2 * np.random.default_rng(15).standard_normal((100, 3)) + np.array([-0.1, 0.05, 0.25])
    return scans, stamps, imu_t, gyr


def run_cfg2():
    d = {}
    for k, v in run_rot(ROT32_PARAMS, rot32_inputs).items():
        d[f"rot32_{k}"] = v
    for k, v in run_backend(("rot_utbm", "livox_ka")).items():
        d[k] = v
    lv = run_livox(LIVOX_KA_PARAMS)
    for k, v in lv.items():
        d[f"livoxka_{k}"] = v
    fe = run_frontend(LIVOX_KA_PARAMS, FRONTEND_KA_PARAMS, full_solves=())
    for k in ("n_frames", "abs_pose", "rel_pose", "kf", "n_solves", "pose_in", "pose_out", "n_blocks", "n_queries", "gn_status", "records_sha"):
        d[f"frontendka_{k}"] = fe[k]
    return d



def backend_inputs(flavour):
    """Room scene (BACKEND_PARAMS[flavour]["room"]: seed / map leaf) + a body pose 5 cm / 0.5 deg off; (Q2, T2) = the LiDAR pose handed to the find* functions
    (Q2 = q * q_lb^-1, T2 = t - Q2 t_lb, L/src/BackendFusion.cpp:929-930), all in numpy f64 with Eigen's expressions."""
    from tests import frontend_chain as F
    B = BACKEND_PARAMS[flavour]
    room = synth.make_room(**dict(dict(seed=12, n_query=1500, n_edge_query=200), **B.get("room", {})))
    qlb, tlb = np.array(B["q_lb"]), np.array(B["t_lb"])
    # body pose from the scene's LiDAR pose: q_b = q_l * q_lb, t_b = t_l + q_l * t_lb  (inverse of the two lines above for unit q_lb)
    q_l, t_l = room["q_true"], room["t_true"]
    qb, tb = F.eigen_qmul(q_l, qlb), t_l + F.eigen_qrot(q_l, tlb[None, :])[0]
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(3), 0.05, 0.5)
    Q2 = F.eigen_qmul(q0, F.eigen_qinv(qlb))
    T2 = t0 - F.eigen_qrot(Q2, tlb[None, :])[0]
    z = lambda a: np.zeros((a.shape[0], 1), np.float32)                                 # noqa: E731
    return dict(surf_map=np.c_[room["map_xyz"], room["map_refl"]].astype(np.float32), edge_map=np.c_[room["edge_map_xyz"], z(room["edge_map_xyz"])].astype(np.float32),
                surf_q=np.c_[room["q_xyz"], room["q_refl"]].astype(np.float32), edge_q=np.c_[room["eq_xyz"], z(room["eq_xyz"])].astype(np.float32),
                t0=t0, q0=q0, Q2=Q2, T2=T2, qlb=qlb, tlb=tlb)


def run_backend(flavours=("livox", "rot")):
    d = {}
    for key in flavours:
        fl = key.split("_")[0]
        i, B = backend_inputs(key), BACKEND_PARAMS[key]
        srec, erec = R.backend_associate(fl, i["surf_map"], i["edge_map"], i["surf_q"], i["edge_q"], i["Q2"], i["T2"], B["kd_max_radius"],
                                         B["surf_dist_thres"], B["lidar_const"], B["reflect_thres"])
        srows, erows = R.backend_rows(fl, srec, erec, i["qlb"], i["tlb"], i["t0"], i["q0"])
        d.update({f"{key}_t0": i["t0"], f"{key}_q0": i["q0"], f"{key}_Q2": i["Q2"], f"{key}_T2": i["T2"],
                  f"{key}_surf_rec": srec[:, :7].astype(np.float32), f"{key}_surf_score": srec[:, 7], f"{key}_edge_rec": erec.astype(np.float32),
                  f"{key}_surf_rows": srows, f"{key}_edge_rows": erows})
        assert np.array_equal(srec[:, :7].astype(np.float32).astype(np.float64), srec[:, :7]) and np.array_equal(erec.astype(np.float32).astype(np.float64), erec)
    return d


def format_inputs(n=5000, seed=3):
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    pts = np.zeros(n, O.CUSTOM_POINT)
    pts["offset_time"] = np.sort(rng.integers(0, 100_000_000, n)).astype(np.uint32)
    for k in ("x", "y", "z"):
        pts[k] = rng.normal(0, 20, n).astype(np.float32)
    pts["reflectivity"] = rng.integers(0, 256, n)
    pts["line"] = rng.integers(0, 6, n)
    pts["tag"] = rng.integers(0, 256, n)
    zero = pts[:64].copy()
    zero["offset_time"] = 0
    return pts, zero


def run_format():
    pts, zero = format_inputs()
    a, z = R.format_convert(pts), R.format_convert(zero)
    return dict(n=a.shape[0], sha=sha(a), every16=a[::16], zero_case=z)


MARG_POS, MARG_IDX_T, MARG_IDX_Q = 15, 6, 9      # a 15-dof window state with this keyframe's translation / rotation at 6 / 9


def marg_inputs():
    g = np.load(os.path.join(HERE, "ref_backend.npz"))
    i = backend_inputs("livox")
    srec = np.c_[g["livox_surf_rec"].astype(np.float64), g["livox_surf_score"]]
    erec = g["livox_edge_rec"].astype(np.float64)
    return i, srec, erec


def run_marg():
    i, srec, erec = marg_inputs()
    rows, A, b = R.marg_lidar(srec, erec, i["qlb"], i["tlb"], i["t0"], i["q0"], MARG_POS, MARG_IDX_T, MARG_IDX_Q)
    return dict(n_rows=rows.shape[0], rows_sha=sha(rows), rows_every8=rows[::8], A=A, b=b)


LM_WIDTH, LM_SURF_MAP_LEAF, LM_EDGE_MAP_LEAF, LM_SURF_LEAF, LM_EDGE_LEAF = 3, 0.4, 0.2, 0.4, 0.2


def localmap_inputs(n_kf=7, n_surf=900, n_edge=160, seed=41):
    """Keyframes of a small room (many multi-point voxels at these leaves), body poses along a gentle curve, a tilted extrinsic."""
    rng = np.random.default_rng(seed)
    q_bl = np.array([0.98, 0.05, -0.12, 0.1]); q_bl /= np.linalg.norm(q_bl)
    t_bl = np.array([0.05, -0.02, 0.11])
    surf, edge, poses = [], [], []
    for k in range(n_kf):
        surf.append(np.concatenate([rng.uniform(-3, 3, (n_surf, 2)), rng.normal(0, 0.3, (n_surf, 1)), rng.uniform(0, 25, (n_surf, 1))], 1).astype(np.float32))
        edge.append(np.concatenate([rng.uniform(-3, 3, (n_edge, 3)), rng.uniform(0, 25, (n_edge, 1))], 1).astype(np.float32))
        ang = 0.07 * k
        q = np.array([np.cos(ang / 2), 0.02 * k, -0.01 * k, np.sin(ang / 2)]); q /= np.linalg.norm(q)
        poses.append(np.concatenate([q, [0.4 * k, -0.15 * k, 0.03 * k]]))
    return dict(q_bl=q_bl, t_bl=t_bl, surf=surf, edge=edge, poses=poses)


def run_localmap():
    i = localmap_inputs()
    lm = R.LocalMapSlice(LM_WIDTH, LM_SURF_MAP_LEAF, LM_EDGE_MAP_LEAF, LM_SURF_LEAF, LM_EDGE_LEAF, i["q_bl"], i["t_bl"])
    out = {}
    for k, (s, e, p) in enumerate(zip(i["surf"], i["edge"], i["poses"])):
        r = lm.keyframe(s, e)
        for name, a in r.items():
            out[f"kf{k}_{name}"] = a
        lm.commit(p)
    lm.close()
    return out


def run_factors():
    f = factor_inputs()
    n = f["cp"].shape[0]
    e, p, pi = np.zeros((n, 8)), np.zeros((n, 8)), np.zeros((n, 8))
    for i in range(n):
        e[i] = R.edge_factor(f["cp"][i], f["a"][i], f["b"][i], f["qlb"], f["tlb"], f["s"][i], f["t"][i], f["q"][i])
        p[i] = R.plane_factor(f["cp"][i], f["n"][i], f["qlb"], f["tlb"], f["d"][i], f["s"][i], f["t"][i], f["q"][i])
        pi[i] = R.plane_incre_factor(f["cp"][i], f["n"][i], f["d"][i], f["q"][i], f["t"][i])
    return dict(edge=e, plane=p, plane_incre=pi)


def main():
    if not R.build():
        raise SystemExit("/root/reference is not present: the reference fixtures can only be generated in the build container")
    np.savez_compressed(os.path.join(HERE, "ref_rot.npz"), **run_rot())
    np.savez_compressed(os.path.join(HERE, "ref_livox.npz"), **run_livox())
    np.savez_compressed(os.path.join(HERE, "ref_factors.npz"), **run_factors())
    np.savez_compressed(os.path.join(HERE, "ref_frontend.npz"), **run_frontend())
    np.savez_compressed(os.path.join(HERE, "ref_frontend_R.npz"), **run_frontend_rot())
    np.savez_compressed(os.path.join(HERE, "ref_backend.npz"), **run_backend())
    np.savez_compressed(os.path.join(HERE, "ref_format.npz"), **run_format())
    np.savez_compressed(os.path.join(HERE, "ref_marg.npz"), **run_marg())
    np.savez_compressed(os.path.join(HERE, "ref_localmap.npz"), **run_localmap())
    np.savez_compressed(os.path.join(HERE, "ref_cfg2.npz"), **run_cfg2())
    for f in ("ref_cfg2.npz", "ref_rot.npz", "ref_livox.npz", "ref_factors.npz", "ref_frontend.npz", "ref_frontend_R.npz", "ref_backend.npz", "ref_format.npz", "ref_marg.npz", "ref_localmap.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()


# ---- reference functor rows of given correspondence records (tests/test_reference_gpu.py::test_gpu_linearize_vs_reference_functors) ----
def functor_rows(R, rs, re_, qlb, tlb, ss, se, t0, q0):
    """[J(7) r] rows of LidarPlaneNormFactor / LidarEdgeFactor ::Create()->Evaluate() (oracle/_ref/libref_factors.so = the reference's
    LidarKeyframeFactor.h compiled as is) for the surf records rs (cp, n, d, score) and the edge records re_ (cp, a, b, s)."""
    ns, ne = len(rs["d"]), len(re_["s"])
    rows_s = np.zeros((ns, 8))
    for i in range(ns):
        o = R.plane_factor(np.asarray(rs["cp"][i], np.float64), np.asarray(rs["n"][i], np.float64), qlb, tlb, float(rs["d"][i]), float(rs["score"][i]) * ss, t0, q0)
        rows_s[i] = np.r_[o[1:8], o[0]]
    rows_e = np.zeros((ne, 8))
    for i in range(ne):
        o = R.edge_factor(np.asarray(re_["cp"][i], np.float64), np.asarray(re_["a"][i], np.float64), np.asarray(re_["b"][i], np.float64), qlb, tlb, float(re_["s"][i]) * se, t0, q0)
        rows_e[i] = np.r_[o[1:8], o[0]]
    return rows_s, rows_e


def make_functor_fixture(inputs_npz, out_npz):
    """inputs: gpurun_out/functor_inputs.npz written by tools/dump_functor_inputs.py ON THE GPU (the records lili_s2m_associate produced for the
    scene of the test, both variants); output: the same records + the reference functors' rows -> tests/golden/ref_functor_rows.npz."""
    sys.path.insert(0, ROOT)
    from oracle import ref as R
    assert R.available(), "build oracle/_ref first (make -C oracle/refshim)"
    d = np.load(inputs_npz)
    out = {}
    for variant in ("livox", "rot"):
        rs = {k: d[f"{variant}_s_{k}"] for k in ("cp", "n", "d", "score")}
        re_ = {k: d[f"{variant}_e_{k}"] for k in ("cp", "a", "b", "s")}
        qlb, tlb, t0, q0 = d[f"{variant}_qlb"], d[f"{variant}_tlb"], d[f"{variant}_t0"], d[f"{variant}_q0"]
        ss, se = float(d[f"{variant}_ss"]), float(d[f"{variant}_se"])
        rows_s, rows_e = functor_rows(R, rs, re_, qlb, tlb, ss, se, t0, q0)
        for k, v in rs.items():
            out[f"{variant}_s_{k}"] = v
        for k, v in re_.items():
            out[f"{variant}_e_{k}"] = v
        out[f"{variant}_rows_s"], out[f"{variant}_rows_e"] = rows_s, rows_e
    np.savez_compressed(out_npz, **out)
    return out
