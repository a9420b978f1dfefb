"""Synthetic 3-keyframe IMU + LiDAR window (BASELINE configs[4]; FR_IOSB_Long is not available offline) shared by
tests/test_window_cpu.py and tests/test_window_gpu.py: a room map, three keyframes on a smooth body trajectory with a 200 Hz IMU
stream between them, feature queries of every keyframe expressed in its LiDAR frame, and the assembly of the reference's window
problem (L/src/BackendFusion.cpp:843-992) on the oracle's restated blocks (oracle/lo_window.py)."""
import numpy as np

import lili_om_amd as L
from lili_om_amd import synth
from oracle import lo_window as W

N_KF = 3
DT_KF, IMU_HZ = 0.2, 200.0
G_VEC = np.array([0.0, 0.0, -9.805])


def _exp_z(ang):
    return np.array([np.cos(ang / 2), 0.0, 0.0, np.sin(ang / 2)])


def trajectory(t):
    """Body pose, velocity, world acceleration and body rate at time t: constant world acceleration, constant yaw rate."""
    p0, v0, a = np.array([0.8, -0.5, 1.5]), np.array([0.9, 0.4, 0.05]), np.array([0.6, -0.8, 0.1])
    w = np.array([0.0, 0.0, 0.35])
    q = W.qmul(_exp_z(np.radians(20.0)), _exp_z(w[2] * t))
    return p0 + v0 * t + 0.5 * a * t * t, q, v0 + a * t, a, w


def imu_between(t0, t1, ba, bg):
    """IMU samples (dt, acc, gyr) for the interval, as the reference's processIMU feeds Preintegration (first sample = acc0 / gyr0)."""
    n = int(round((t1 - t0) * IMU_HZ))
    out = []
    for k in range(n + 1):
        t = t0 + k / IMU_HZ
        _, q, _, a, w = trajectory(t)
        acc = W.qrot(W.qinv(q), a - G_VEC) + ba
        out.append((1.0 / IMU_HZ, acc, w + bg))
    return out


def make_window(seed=41, n_surf=2500, n_edge=200):
    room = synth.make_room(seed=seed, n_query=10, n_edge_query=10)
    rng = np.random.default_rng(seed + 1)
    P = L.make_params("livox")
    ba, bg = np.array([0.02, -0.01, 0.015]), np.array([0.002, -0.001, 0.0015])
    kfs = []
    for k in range(N_KF):
        t = k * DT_KF
        p, q, v, _, _ = trajectory(t)
        Q2, T2 = L.api.assoc_transform(p, q, P)            # LiDAR pose of this body pose
        pick = rng.choice(room["map_xyz"].shape[0], n_surf, replace=True)
        qw = room["map_xyz"][pick].astype(np.float64) + rng.normal(0, 0.01, (n_surf, 3)) + rng.uniform(-0.15, 0.15, (n_surf, 3))
        epick = rng.choice(room["edge_map_xyz"].shape[0], n_edge, replace=True)
        ew = room["edge_map_xyz"][epick].astype(np.float64) + rng.normal(0, 0.02, (n_edge, 3))
        Qi = W.qinv(Q2)
        q_local = np.array([W.qrot(Qi, x - T2) for x in qw], np.float32)
        e_local = np.array([W.qrot(Qi, x - T2) for x in ew], np.float32)
        q_refl = (np.float32(10.0) + rng.integers(0, 30, n_surf).astype(np.float32) * np.float32(0.1) + np.float32(0.05))
        kfs.append(dict(t_true=p, q_true=q, sb_true=np.concatenate([v, ba, bg]), q_xyz=q_local, q_refl=q_refl, eq_xyz=e_local))
    pres = []
    for k in range(N_KF - 1):
        s = imu_between(k * DT_KF, (k + 1) * DT_KF, ba, bg)
        pre = W.Preintegration(s[0][1], s[0][2], ba + 0.003, bg - 0.0004)     # linearised at slightly wrong biases, like a running filter
        for dt, acc, gyr in s[1:]:
            pre.push_back(dt, acc, gyr)
        pres.append(dict(pre=pre, samples=s, ba=ba + 0.003, bg=bg - 0.0004))
    # initial guess: the window the front-end would hand over — poses a few cm / tenths of a degree off, speed-bias a little off
    init = []
    for k, kf in enumerate(kfs):
        t0, q0 = synth.perturbed_pose(kf["t_true"], kf["q_true"], np.random.default_rng(seed + 10 + k), 0.04, 0.4)
        init.append(dict(t=np.asarray(t0, np.float64), q=np.asarray(q0, np.float64), sb=kf["sb_true"] + np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.004, 3), rng.normal(0, 0.0005, 3)])))
    return dict(room=room, P=P, kfs=kfs, pres=pres, init=init)


def robust_rows(rows, a=1.0):
    """CauchyLoss(a) + Triggs corrector on raw 1-residual rows [J(7) r] (vectorised ceres::Corrector, alpha = 0 branch: rho'' < 0 always)."""
    r = rows[:, 7]
    sq = r * r
    b = a * a
    sm = 1.0 + sq / b
    cost = 0.5 * float((b * np.log(sm)).sum())
    s1 = np.sqrt(1.0 / sm)
    return rows[:, :7] * s1[:, None], r * s1, cost


def build_problem(win, lidar_block, joint_lidar=None):
    """The reference's window problem; lidar_block(k) -> residual-block function of (t_k, q_k) for keyframe k.
    joint_lidar (optional, instead): ONE residual-block function of (t_0, q_0, ..., t_K-1, q_K-1) for the lidar terms of all keyframes."""
    pb = W.Problem()
    for k, s in enumerate(win["init"]):
        pb.add_parameter(f"t{k}", s["t"])
        pb.add_parameter(f"q{k}", s["q"], quat=True)
        pb.add_parameter(f"sb{k}", s["sb"])
    for k in range(N_KF - 1):                                   # !marg: speed-bias priors on all but the newest keyframe (L:897-909)
        prior = win["init"][k]["sb"].copy()
        pb.add_residual(lambda sb, prior=prior: W.speed_bias_prior(prior, sb), [f"sb{k}"])
    for k in range(N_KF - 1):                                   # IMU factors between consecutive keyframes (L:911-921)
        pre = win["pres"][k]["pre"]
        pb.add_residual(lambda ti, qi, sbi, tj, qj, sbj, pre=pre: W.imu_factor(pre, ti, qi, sbi, tj, qj, sbj),
                        [f"t{k}", f"q{k}", f"sb{k}", f"t{k + 1}", f"q{k + 1}", f"sb{k + 1}"])
    if joint_lidar is not None:
        pb.add_residual(joint_lidar, [n for k in range(N_KF) for n in (f"t{k}", f"q{k}")])
        return pb
    for k in range(N_KF):                                       # lidar blocks of every keyframe, CauchyLoss(1) (L:923-975)
        pb.add_residual(lidar_block(k), [f"t{k}", f"q{k}"])
    return pb
