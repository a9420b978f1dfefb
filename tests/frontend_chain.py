"""Test harness (not product code): the control flow of the reference's front-end node, LidarOdometry::run
(L/src/LidarOdometry.cpp:652-700) — poseInitialization (:415-441), buildLocalMap (:274-300), downSampleCloud (:313-323),
updateTransformationWithCeres (:483-585, with one Gauss-Newton step per ceres::Solve: the protocol of
oracle/refshim/ref_lo.cpp and of lili_s2m_iterate), savePoses (:325-350), computeRelative (:443-480), clearCloud (:302-311) —
restated over a small backend interface so that the SAME loop runs on the oracle primitives (CPU) and on the HIP path (GPU)
and can be compared with what the reference's own node did on the same frames (tests/golden/ref_frontend.npz)."""
import numpy as np


def eigen_qrot(q, v):
    """Eigen 3.3 QuaternionBase::_transformVector on rows of v (f64, operation for operation): v + w*(2 u x v) + u x (2 u x v)."""
    w, ux, uy, uz = (float(x) for x in q)
    vx, vy, vz = v[:, 0], v[:, 1], v[:, 2]
    cx, cy, cz = uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx
    cx, cy, cz = cx + cx, cy + cy, cz + cz
    dx, dy, dz = uy * cz - uz * cy, uz * cx - ux * cz, ux * cy - uy * cx
    return np.stack([(vx + cx * w) + dx, (vy + cy * w) + dy, (vz + cz * w) + dz], 1)


def eigen_qmul(a, b):
    aw, ax, ay, az = (float(x) for x in a)
    bw, bx, by, bz = (float(x) for x in b)
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def eigen_qinv(q):
    w, x, y, z = (float(v) for v in q)
    n2 = x * x + y * y + z * z + w * w
    return np.array([w / n2, -x / n2, -y / n2, -z / n2])


class OracleBackend:
    """voxel filter / kd-tree / association / linearisation / GN step from oracle/ (literal PCL mode: std::sort)."""

    def __init__(self, O, stable=False):
        self.O, self.P, self.stable = O, O.params("frontend"), stable
        self.log = []

    def voxel(self, pts4, leaf=0.4):
        return self.O.voxel_grid(pts4, leaf, stable=self.stable)[0] if pts4.shape[0] else pts4

    def set_map(self, clouds_world, leaf=0.4):
        """clouds_world: list of (n,4) clouds already in the odometry frame; the map is their voxel-filtered union."""
        cat = np.concatenate(clouds_world, 0) if clouds_world else np.zeros((0, 4), np.float32)
        self.map = self.voxel(np.ascontiguousarray(cat, np.float32), leaf)
        self.tree = self.O.KdTree(np.ascontiguousarray(self.map[:, :3]))
        return self.map.shape[0]

    def match(self, queries4, q, t, n_iter):
        q, t = np.array(q, np.float64), np.array(t, np.float64)
        xyz = np.ascontiguousarray(queries4[:, :3])
        for _ in range(n_iter):
            rs = self.O.associate_surf(self.tree, None, xyz, None, q, t, self.P)
            G, _, _ = self.O.linearize_surf(rs, t, q, self.P)
            st, t2, q2, _ = self.O.gn_step(G, t, q)
            self.log.append(dict(pose_in=np.r_[q, t], pose_out=np.r_[q2, t2], n_blocks=int(rs["count"]), n_map=self.map.shape[0], n_queries=xyz.shape[0]))
            q, t = (q2, t2) if st == 0 else (q, t)
            if q[0] < 0:
                q = -q                               # unifyQuaternion (L:538-548)
        return q, t


def run_frontend_chain(backend, surf_features, scan_match_cnt=6, leaf=0.4):
    """surf_features: per frame, (n,4) float32 rows x, y, z, curvature of /surf_features.  Returns per-frame abs poses
    (qw qx qy qz x y z), rel poses and the keyframe-independent state the reference node would hold."""
    abs_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
    rel_pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
    poses, surf_frames, recent = [], [], []
    latest_frame_idx = 0
    surf_last_ds = np.zeros((0, 4), np.float32)
    out_abs, out_rel = [], []
    initialized = False
    for surf in surf_features:
        surf = np.ascontiguousarray(surf, np.float32)
        if not initialized:                                    # savePoses + checkInitialization (L:659-663)
            poses.append(abs_pose.copy()); surf_frames.append(surf_last_ds.copy())
            initialized = True
            out_abs.append(abs_pose.copy()); out_rel.append(rel_pose.copy())
            continue
        # poseInitialization
        q0, t0, dq, dt = abs_pose[:4], abs_pose[4:], rel_pose[:4], rel_pose[4:]
        t0 = eigen_qrot(q0, dt[None, :])[0] + t0
        q0 = eigen_qmul(q0, dq)
        abs_pose = np.r_[q0, t0]
        # buildLocalMap
        if len(poses) <= 1:
            world = [surf]
        else:
            def tf(i):
                c = surf_frames[i]
                w = eigen_qrot(poses[i][:4], c[:, :3].astype(np.float64)) + poses[i][4:]
                return np.concatenate([w.astype(np.float32), c[:, 3:4]], 1)
            if len(recent) < 20:
                recent.append(tf(len(poses) - 1))
            elif latest_frame_idx != len(poses) - 1:
                recent.pop(0)
                latest_frame_idx = len(poses) - 1
                recent.append(tf(latest_frame_idx))
            world = recent
        # downSampleCloud
        n_map = backend.set_map(world, leaf)
        surf_last_ds = backend.voxel(surf, leaf)
        # updateTransformationWithCeres
        if n_map >= 10:
            match_cnt = 8 if len(poses) < 2 else scan_match_cnt
            q, t = backend.match(surf_last_ds, abs_pose[:4], abs_pose[4:], match_cnt)
            abs_pose = np.r_[q, t]
        # savePoses
        poses.append(abs_pose.copy()); surf_frames.append(surf_last_ds.copy())
        # computeRelative: previous frame's pose -> this one
        q1, t1 = poses[-2][:4], poses[-2][4:]
        q1i = eigen_qinv(q1)
        rel_pose = np.r_[eigen_qmul(q1i, abs_pose[:4]), eigen_qrot(q1i, (abs_pose[4:] - t1)[None, :])[0]]
        # clearCloud
        if len(surf_frames) > 7:
            surf_frames[len(surf_frames) - 8] = np.zeros((0, 4), np.float32)
        out_abs.append(abs_pose.copy()); out_rel.append(rel_pose.copy())
    return np.array(out_abs), np.array(out_rel)
