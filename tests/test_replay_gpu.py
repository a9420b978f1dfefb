"""End-to-end bag replay on the GPU path: a synthetic Livox sequence is written as a ROS bag
(livox_ros_driver/CustomMsg + sensor_msgs/Imu), read back by the pure-Python reader and pushed through
CustomMsg conversion -> gyro integration -> Livox extractor -> voxel filter -> local map -> front-end matcher."""
import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import replay, rosbag
from tests import seq_harness as H

pytestmark = pytest.mark.gpu


def _custom(scan):
    frac = (scan[:, 3] - np.floor(scan[:, 3])) / np.float32(0.1)
    order = np.argsort(frac, kind="stable")
    s = scan[order]
    pts = np.zeros(s.shape[0], rosbag.CUSTOM_POINT)
    pts["x"], pts["y"], pts["z"] = s[:, 0], s[:, 1], s[:, 2]
    pts["line"] = np.floor(s[:, 3]).astype(np.uint8)
    pts["offset_time"] = np.round(np.clip(frac[order], 0, 1) * 99_000_000).astype(np.uint32)
    pts["reflectivity"] = np.clip(np.round(s[:, 4] * 10), 0, 255).astype(np.uint8)
    return pts


def test_replay_synthetic_bag(gpu_ctx, tmp_path):
    n = 9
    frames = H.make_frames(n + 2)
    msgs = []
    for k, scan in enumerate(frames):
        t = 50.0 + 0.1 * k
        for j in range(20):
            ti = t + 0.005 * j
            msgs.append(("/livox/imu", "sensor_msgs/Imu", ti, rosbag.encode_imu(ti, (0.0, 0.0, 0.0), seq=20 * k + j)))
        msgs.append(("/livox/lidar", "livox_ros_driver/CustomMsg", t + 0.1, rosbag.encode_livox_custom(_custom(scan), t, seq=k)))
    path = str(tmp_path / "synthetic.bag")
    rosbag.write_bag(path, msgs, compression="bz2", chunk_messages=16)
    t0, q0, _ = H.gt_pose(0)
    out = replay.replay(path, gpu_ctx, "/livox/lidar", "/livox/imu", first_pose=(t0, q0))
    assert len(out) == n                                            # the last two scans stay queued, like cloudHandler
    assert all(np.array_equal(r["q_imu"], [1, 0, 0, 0]) for r in out)
    assert min(r["n_surf"] for r in out) > 3000 and min(r["n_query"] for r in out) > 500
    err = [np.linalg.norm(r["t"] - H.gt_pose(f)[0]) for f, r in enumerate(out)]
    print("replay position errors (m):", np.round(err, 3))
    assert np.sqrt(np.mean(np.square(err))) < 0.15 and max(err) < 0.3
    assert abs(out[3]["stamp"] - 50.3) < 1e-6
