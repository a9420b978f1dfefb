"""ROS bag (format 2.0) reader and message decoders of lili_om_amd.rosbag: round trips through the
module's own minimal writer (no bag files or ROS installation offline), compressed chunks, the LZ4 block decoder on a
hand-assembled stream, and malformed input."""
import struct

import numpy as np
import pytest

from lili_om_amd import rosbag


def _messages(seed=0, n_scans=5):
    rng = np.random.default_rng(seed)
    msgs = []
    for k in range(n_scans):
        t = 100.0 + 0.1 * k
        for j in range(20):
            ti = t + 0.005 * j
            msgs.append(("/imu", "sensor_msgs/Imu", ti + 1e-4, rosbag.encode_imu(ti, rng.normal(0, 0.1, 3), seq=k * 20 + j)))
        pts = np.zeros(200, rosbag.CUSTOM_POINT)
        pts["offset_time"] = np.sort(rng.integers(0, 10 ** 8, 200))
        pts["x"], pts["y"], pts["z"] = rng.normal(0, 10, (3, 200)).astype(np.float32)
        pts["reflectivity"], pts["tag"], pts["line"] = rng.integers(0, 256, 200), rng.integers(0, 256, 200), rng.integers(0, 6, 200)
        msgs.append(("/livox/lidar", "livox_ros_driver/CustomMsg", t + 0.1, rosbag.encode_livox_custom(pts, t, timebase=123456789 + k, seq=k)))
        cloud = rng.normal(0, 5, (150, 5)).astype(np.float32)
        msgs.append(("/livox_ros_points", "sensor_msgs/PointCloud2", t + 0.1,
                     rosbag.encode_pointcloud2(cloud, ["x", "y", "z", "intensity", "curvature"], t, point_step=48, offsets=[0, 4, 8, 32, 36], seq=k)))
    return msgs


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_bag_round_trip(tmp_path, compression):
    msgs = _messages()
    path = tmp_path / "t.bag"
    rosbag.write_bag(str(path), msgs, compression=compression, chunk_messages=7)
    got = list(rosbag.Bag(str(path)).messages())
    assert len(got) == len(msgs)
    for (tp, ty, t, raw), (tp2, ty2, t2, raw2) in zip(msgs, got):
        assert (tp, ty, raw) == (tp2, ty2, raw2) and abs(t - t2) < 1e-6
    only = list(rosbag.Bag(str(path)).messages(topics=["/imu"]))
    assert len(only) == 100 and all(m[0] == "/imu" for m in only)


def test_message_decoders():
    msgs = _messages(1, 2)
    imu = [m for m in msgs if m[0] == "/imu"][3]
    d = rosbag.decode_imu(imu[3])
    assert abs(d["header"]["stamp"] - (100.0 + 0.015)) < 1e-9 and d["angular_velocity"].shape == (3,) and d["linear_acceleration"][2] == 9.81
    cm = [m for m in msgs if m[0] == "/livox/lidar"][1]
    c = rosbag.decode_livox_custom(cm[3])
    assert c["point_num"] == 200 and c["points"].dtype == rosbag.CUSTOM_POINT and c["timebase"] == 123456790
    assert np.all(np.diff(c["points"]["offset_time"].astype(np.int64)) >= 0)
    pc = [m for m in msgs if m[0] == "/livox_ros_points"][0]
    p = rosbag.decode_pointcloud2(pc[3])
    assert p["point_step"] == 48 and p["width"] == 150 and p["fields"]["intensity"] == (32, 7, 1) and p["fields"]["curvature"][0] == 36
    xyzic = rosbag.pointcloud2_xyz_aux(p, "intensity", "curvature")
    rng = np.random.default_rng(1)
    for _ in range(20):
        rng.normal(0, 0.1, 3)
    # the encoder wrote the same rows back: re-encode and compare bytes
    again = rosbag.encode_pointcloud2(xyzic, ["x", "y", "z", "intensity", "curvature"], p["header"]["stamp"], point_step=48,
                                      offsets=[0, 4, 8, 32, 36], seq=p["header"]["seq"])
    assert again == pc[3]


def test_lz4_block_and_frame():
    blk = bytes([0x35]) + b"abc" + bytes([3, 0]) + bytes([0x00])          # 3 literals, match offset 3 length 5+4, empty tail
    assert rosbag._lz4_block(blk, 12) == b"abcabcabcabc"
    long_lit = bytes(range(200)) * 2
    blk2 = bytes([0xF0, 255, 130]) + long_lit                                # 15 + 255 + 130 = 400 literals, no match
    assert rosbag._lz4_block(blk2, 400) == long_lit
    frame = b"\x04\x22\x4d\x18" + bytes([0x60, 0x40, 0x00]) + struct.pack("<I", len(blk)) + blk + struct.pack("<I", 0x80000000 | 5) + b"hello" + struct.pack("<I", 0)
    assert rosbag._roslz4(frame, 17) == b"abcabcabcabchello"
    with pytest.raises(rosbag.BagError):
        rosbag._lz4_block(bytes([0x05]) + bytes([9, 0]), None)                # match before any output


def test_malformed_bags(tmp_path):
    p = tmp_path / "bad.bag"
    p.write_bytes(b"#ROSBAG V1.2\n")
    with pytest.raises(rosbag.BagError):
        rosbag.Bag(str(p))
    msgs = _messages(2, 1)
    good = tmp_path / "g.bag"
    rosbag.write_bag(str(good), msgs)
    raw = good.read_bytes()
    (tmp_path / "trunc.bag").write_bytes(raw[:len(raw) - 37])
    with pytest.raises(rosbag.BagError):
        list(rosbag.Bag(str(tmp_path / "trunc.bag")).messages())
