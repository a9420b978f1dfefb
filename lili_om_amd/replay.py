"""Bag replay through the GPU path: the FormatConvert -> Preprocessing -> LidarOdometry chain of the
Livox configuration, one scan in flight like the reference's `ros::spin` nodes.

    /livox/lidar (livox_ros_driver/CustomMsg)  --lili_livox_custom_to_cloud-->  PointXYZINormal cloud
    or /livox_ros_points (sensor_msgs/PointCloud2) directly
    /imu (sensor_msgs/Imu)                      --lili_imu_integrate-->          q_imu of the scan
    cloud, q_imu, predicted pose                --lili_frontend_frame-->         pose
        (one C call per scan, device-resident: lili_extract_livox -> VoxelGrid(0.4) of the surf features = queries (L/src/LidarOdometry.cpp:280-323)
         -> outer iterations of the front-end variant against the local map of the last `map_width` frames (L:483-561) -> ring push at the pose
         found + the next frame's local map; `staged=True` keeps the round-4 chain of separate calls with host copies in between, for A/B)

The scan queue follows Preprocessing::cloudHandler (L/src/Preprocessing.cpp:194-215): scan k is processed when scan
k+2 has arrived, `time_scan_next` = stamp of scan k+1, and only once IMU data reaches that time.  The pose prediction
is the constant-velocity model of poseInitialization (L/src/LidarOdometry.cpp:415-480).  This is replay tooling around
the C ABI — the reference's keyframe selection, back-end and loop closure are not part of it."""
import numpy as np

from . import api as A
from . import rosbag, synth


def _predict(poses):
    if len(poses) == 1:
        return poses[-1]
    (ta, qa), (tb, qb) = poses[-2], poses[-1]
    qa_inv = qa * np.array([1, -1, -1, -1])
    dq = synth.quat_mul(qa_inv, qb)
    dt = synth.quat_rot(qa_inv, tb - ta)
    return tb + synth.quat_rot(qb, dt), synth.quat_mul(qb, dq)


def replay(bag_path, ctx, lidar_topic="/livox/lidar", imu_topic="/imu", first_pose=None, n_outer=6, n_outer_first=12,
           map_width=20, leaf=0.4, max_scans=None, on_frame=None, staged=False):
    """Returns a list of dicts (stamp, t, q, n_surf, n_edge, n_query, q_imu) — one per processed scan."""
    P = A.make_params("frontend")
    ex = A.LivoxExtractor(ctx)
    matcher = A.ScanToMapMatcher(ctx, P)
    local = A.LocalMap(ctx, A.KIND_SURF, map_width, leaf, P.kd_max_radius)
    odo = A.FrontendOdometry(ctx, P, leaf_query=leaf, leaf_map=leaf, width=map_width, scan_match_cnt=n_outer, first_match_cnt=n_outer_first, reference_startup=False)
    odo.reset()
    imu = A.ImuIntegrator()
    imu_t, imu_w = [], []
    queue, out, poses = [], [], []
    bag = rosbag.Bag(bag_path)

    def process(stamp, pts5, t_next):
        q_imu = imu.integrate(np.array(imu_t), np.array(imu_w).reshape(-1, 3), t_next)
        if not staged:
            if not poses:
                t0, q0 = first_pose if first_pose is not None else (np.zeros(3), np.array([1.0, 0, 0, 0]))
            else:
                t0, q0 = _predict(poses)
            t, q, info = odo.frame(pts5, t0, q0, q_imu)
            # (a frame whose last Gauss-Newton step was rejected keeps the pose the slot holds — the last accepted iterate, which is also the pose the frame
            #  joined the ring at; the staged path below does the same, so trajectory, next prediction and local map agree on such frames)
            poses.append((t, q))
            rec = dict(stamp=stamp, t=t, q=q, n_surf=info["n_surf"], n_edge=info["n_edge"], n_query=info["n_query"], q_imu=q_imu)
            out.append(rec)
            if on_frame:
                on_frame(rec)
            return
        f = ex.extract(pts5, q_imu)
        surf = np.ascontiguousarray(f["surf"][:, [0, 1, 2, 7]])
        qry, _ = A.voxel_filter(ctx, surf, leaf) if surf.shape[0] else (surf, None)
        if not poses:
            t, q = first_pose if first_pose is not None else (np.zeros(3), np.array([1.0, 0, 0, 0]))
            t, q = np.asarray(t, np.float64), np.asarray(q, np.float64)
        else:
            t0, q0 = _predict(poses)
            local.commit()
            matcher.set_queries(0, A.KIND_SURF, qry)
            matcher.pose_set(0, t0, q0)
            matcher.iterate(0, n_outer_first if len(poses) == 1 else n_outer, A.MASK_SURF)
            t, q, st = matcher.pose_get(0)      # status != 0: the last accepted iterate (as the fused path)
        poses.append((t, q))
        local.push(qry, t, q)
        rec = dict(stamp=stamp, t=t, q=q, n_surf=int(f["surf"].shape[0]), n_edge=int(f["edge"].shape[0]), n_query=int(qry.shape[0]), q_imu=q_imu)
        out.append(rec)
        if on_frame:
            on_frame(rec)

    for topic, mtype, t_rx, raw in bag.messages(topics=[lidar_topic, imu_topic]):
        if topic == imu_topic:
            m = rosbag.decode_imu(raw)
            imu_t.append(m["header"]["stamp"]); imu_w.append(m["angular_velocity"])
        else:
            if mtype == "livox_ros_driver/CustomMsg":
                m = rosbag.decode_livox_custom(raw)
                cloud = A.livox_custom_to_cloud(ctx, m["points"])[:, [0, 1, 2, 8, 9]]
            else:
                m = rosbag.decode_pointcloud2(raw)
                cloud = rosbag.pointcloud2_xyz_aux(m, "intensity", "curvature")
            queue.append((m["header"]["stamp"], np.ascontiguousarray(cloud, np.float32)))
        # cloudHandler: pop the front scan once two newer ones are queued and the IMU covers time_scan_next
        while len(queue) > 2 and imu_t and imu_t[-1] >= queue[1][0]:
            stamp, pts5 = queue.pop(0)
            process(stamp, pts5, queue[0][0])
            if max_scans and len(out) >= max_scans:
                return out
    return out
