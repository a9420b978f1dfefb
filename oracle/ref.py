"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes access to oracle/_ref/: the reference's OWN hot-path sources
(LiLi-OM-ROT/src/Preprocessing.cpp, LiLi-OM/src/Preprocessing.cpp, include/factors/LidarKeyframeFactor.h) compiled
unmodified from /root/reference against the stand-in third-party headers of oracle/refshim/ (recipe:
oracle/refshim/Makefile).  Used to pin the restatement in oracle/*.cpp and to generate tests/golden/ref_*.npz;
never loaded by the product.  /root/reference does not exist on the GPU box: the prebuilt .so files travel, the
sources do not, and nothing here reads /root/reference at run time."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.abspath(os.environ.get("LILI_REF_DIR", os.path.join(HERE, "_ref")))   # LILI_REF_DIR=oracle/_ref_real: the REAL_DEPS build (refshim/README.md)
REFERENCE_ROOT = "/root/reference"
_LIBS = {}


def build():
    """Build oracle/_ref when the reference sources are present (the build container); no-op otherwise."""
    if not os.path.isdir(REFERENCE_ROOT):
        return False
    subprocess.check_call(["make", "-C", os.path.join(HERE, "refshim"), "-s"])
    return True


def available():
    return all(os.path.exists(os.path.join(REF_DIR, f"libref_{n}.so")) for n in ("rot", "livox", "factors", "lo", "backend_L", "backend_R", "format", "marg", "localmap"))


def _lib(name):
    if name not in _LIBS:
        _LIBS[name] = C.CDLL(os.path.join(REF_DIR, f"libref_{name}.so"))
    return _LIBS[name]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Preprocessing:
    """The reference's Preprocessing node (flavour 'rot' or 'livox') driven in-process.  ROS parameters are the
    yaml keys the node reads in its constructor, e.g. {'/preprocessing/line_num': 64, '/preprocessing/ds_rate': 4}."""

    def __init__(self, flavour, params=None, verbose=False):
        self.flavour = flavour
        self.lib = _lib(flavour)
        L = self.lib
        L.ref_pre_create.restype = C.c_void_p
        L.ref_param_num.argtypes = [C.c_char_p, C.c_double]
        L.ref_param_str.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_pre_imu.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.ref_pre_cloud.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]
        L.ref_pre_n_published.argtypes = [C.c_void_p]
        L.ref_pre_msg_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ref_pre_msg_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_pre_destroy.argtypes = [C.c_void_p]
        L.ref_set_verbose(int(bool(verbose)))
        L.ref_param_clear()
        for k, v in (params or {}).items():
            if isinstance(v, str):
                L.ref_param_str(k.encode(), v.encode())
            else:
                L.ref_param_num(k.encode(), float(v))
        self.h = C.c_void_p(L.ref_pre_create())
        self.step = 32 if flavour == "rot" else 48

    def close(self):
        if self.h:
            self.lib.ref_pre_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def imu(self, t, gyr):
        self.lib.ref_pre_imu(self.h, float(t), float(gyr[0]), float(gyr[1]), float(gyr[2]))

    def cloud(self, t, pts):
        """pts: rot — (n,4) float32 x,y,z,intensity;  livox — (n,5) float32 x,y,z,intensity,curvature.
        Packed into the PCL wire layout before it is handed to cloudHandler."""
        pts = np.ascontiguousarray(pts, np.float32)
        n = pts.shape[0]
        buf = np.zeros((n, self.step // 4), np.float32)
        buf[:, 0:3] = pts[:, 0:3]
        buf[:, 3] = 1.0
        if self.flavour == "rot":
            buf[:, 4] = pts[:, 3]
        else:
            buf[:, 8] = pts[:, 3]
            buf[:, 9] = pts[:, 4]
        self.lib.ref_pre_cloud(self.h, float(t), _p(buf), n, self.step)

    def published(self):
        """[(topic, stamp, (n, point_step/4) float32 array)] in publication order."""
        out = []
        for i in range(self.lib.ref_pre_n_published(self.h)):
            topic = C.create_string_buffer(64)
            stamp, step = C.c_double(0), C.c_int(0)
            n = self.lib.ref_pre_msg_info(self.h, i, topic, C.byref(stamp), C.byref(step))
            a = np.zeros((n, max(step.value // 4, 1)), np.float32)
            if n:
                self.lib.ref_pre_msg_data(self.h, i, _p(a))
            out.append((topic.value.decode(), stamp.value, a))
        return out


def run_scans(flavour, params, scans, scan_stamps, imu_stamps, imu_gyr):
    """Replay: IMU samples and clouds are delivered in time order (IMU first on ties, like a 200 Hz stream ahead of a
    10 Hz one).  Returns a list with one dict per PROCESSED scan: {'stamp', 'cutted', 'edge', 'surf'} — cutted/edge/surf
    are the float payloads of /lidar_cloud_cutted, /edge_features, /surf_features."""
    node = Preprocessing(flavour, params)
    ev = [(float(t), 0, i) for i, t in enumerate(imu_stamps)] + [(float(t), 1, i) for i, t in enumerate(scan_stamps)]
    ev.sort()
    for t, kind, i in ev:
        if kind == 0:
            node.imu(t, imu_gyr[i])
        else:
            node.cloud(t, scans[i])
    msgs = node.published()
    node.close()
    res = {}
    for topic, stamp, a in msgs:
        res.setdefault(stamp, {"stamp": stamp})[topic.strip("/").replace("lidar_cloud_", "").replace("_features", "")] = a
    return [res[k] for k in sorted(res)]


def _factor(fn, *args):
    out = np.zeros(8, np.float64)
    a = []
    for x in args:
        if isinstance(x, (float, int)):
            a.append(C.c_double(float(x)))
        else:
            arr = np.ascontiguousarray(x, np.float64)
            a.append(arr)
    cargs = [(_p(x) if isinstance(x, np.ndarray) else x) for x in a]
    rc = fn(*cargs, _p(out), 1)
    if rc != 0:
        raise RuntimeError("Evaluate returned false")
    return out


def edge_factor(cp, a, b, qlb, tlb, s, t, q):
    """LidarEdgeFactor::Create(...)->Evaluate: [r, dr/dt(3), dr/dq(4, wxyz)]."""
    return _factor(_lib("factors").ref_edge_factor, cp, a, b, qlb, tlb, float(s), t, q)


def plane_factor(cp, n, qlb, tlb, d, score, t, q):
    """LidarPlaneNormFactor::Create(...)->Evaluate: [r, dr/dt(3), dr/dq(4)]."""
    return _factor(_lib("factors").ref_plane_factor, cp, n, qlb, tlb, float(d), float(score), t, q)


def plane_incre_factor(cp, n, d, q, t):
    """LidarPlaneNormIncreFactor::Create(...)->Evaluate: [r, dr/dq(4), dr/dt(3)]."""
    return _factor(_lib("factors").ref_plane_incre_factor, cp, n, float(d), q, t)


class LidarOdometry:
    """The reference's front-end node (LiLi-OM/src/LidarOdometry.cpp, compiled unmodified) driven frame by frame.
    ceres::Solve is the documented stand-in: it logs the residual blocks the reference built and takes ONE Gauss-Newton
    step (oracle/refshim/ref_lo.cpp).  frame() takes the three clouds of Preprocessing as (n, 12) float32 rows in the
    48-byte PointXYZINormal layout."""

    def __init__(self, params=None, verbose=False, flavour="livox"):
        """flavour "rot": LiLi-OM-ROT/src/LidarOdometry.cpp (libref_lo_R.so; rows are 32-byte PointXYZI: 8 floats)."""
        L = self.lib = _lib("lo_R" if flavour == "rot" else "lo")
        self.row_floats = int(L.ref_lo_point_floats())
        L.ref_lo_create.restype = C.c_void_p
        L.ref_param_num.argtypes = [C.c_char_p, C.c_double]
        L.ref_param_str.argtypes = [C.c_char_p, C.c_char_p]
        L.ref_lo_frame.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ref_lo_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.ref_lo_destroy.argtypes = [C.c_void_p]
        L.ref_lo_solve_info.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_lo_solve_data.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_msg_info.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ref_msg_data.argtypes = [C.c_int, C.c_void_p]
        L.ref_set_verbose(int(bool(verbose)))
        L.ref_param_clear()
        for k, v in (params or {}).items():
            if isinstance(v, str):
                L.ref_param_str(k.encode(), v.encode())
            else:
                L.ref_param_num(k.encode(), float(v))
        self.h = C.c_void_p(L.ref_lo_create())

    def close(self):
        if self.h:
            self.lib.ref_lo_destroy(self.h)
            self.h = None

    def frame(self, stamp, edge12, surf12, full12):
        a = [np.ascontiguousarray(x, np.float32).reshape(-1, self.row_floats) for x in (edge12, surf12, full12)]
        self.lib.ref_lo_frame(self.h, float(stamp), _p(a[0]), a[0].shape[0], _p(a[1]), a[1].shape[0], _p(a[2]), a[2].shape[0])
        ap, rp, kf = np.zeros(7), np.zeros(7), C.c_int(0)
        self.lib.ref_lo_pose(self.h, _p(ap), _p(rp), C.byref(kf))
        return ap, rp, bool(kf.value)          # abs_pose / rel_pose: qw qx qy qz | x y z

    def solves(self, first=0):
        """Every ceres::Solve call so far: dict(pose_in, pose_out (qw qx qy qz x y z), records (n,7) = cp, weight*n,
        weight*d; rows (n,8) = r, dr/dq(4), dr/dt(3) raw; map (m,4), queries (k,4) = x y z curvature; gn_status)."""
        out = []
        for i in range(first, self.lib.ref_lo_n_solves()):
            sizes = np.zeros(4, np.int32)
            pin, pout = np.zeros(7), np.zeros(7)
            self.lib.ref_lo_solve_info(i, _p(sizes), _p(pin), _p(pout))
            nb, nm, nq, st = [int(v) for v in sizes]
            rec, rows = np.zeros((nb, 7)), np.zeros((nb, 8))
            mp, qs = np.zeros((nm, 4), np.float32), np.zeros((nq, 4), np.float32)
            self.lib.ref_lo_solve_data(i, _p(rec), _p(rows), _p(mp), _p(qs))
            out.append(dict(pose_in=pin, pose_out=pout, records=rec, rows=rows, map=mp, queries=qs, gn_status=st))
        return out

    def published(self):
        out = []
        for i in range(self.lib.ref_n_published()):
            topic = C.create_string_buffer(64)
            stamp, step = C.c_double(0), C.c_int(0)
            n = self.lib.ref_msg_info(i, topic, C.byref(stamp), C.byref(step))
            if step.value:
                a = np.zeros((n, step.value // 4), np.float32)
            else:
                a = np.zeros(n, np.float64)
            if n:
                self.lib.ref_msg_data(i, _p(a))
            out.append((topic.value.decode(), stamp.value, a))
        return out


def backend_associate(flavour, surf_map4, edge_map4, surf_q4, edge_q4, q_assoc, t_assoc, kd_max_radius, surf_dist_thres,
                      lidar_const, reflect_thres=0.0):
    """findCorrespondingSurfFeatures / findCorrespondingCornerFeatures / transformPoint of the reference's BackendFusion.cpp
    (flavour 'livox' = LiLi-OM, 'rot' = LiLi-OM-ROT), member-function text sliced out of the file at build time
    (oracle/refshim/ref_backend.cpp).  Clouds are (n,4) float32 rows x y z aux.  Returns (surf_records (n,8) = cp, weight*n,
    weight*d, score; edge_records (n,10) = cp, A, B, s)."""
    L = _lib("backend_L" if flavour == "livox" else "backend_R")
    a = [np.ascontiguousarray(x, np.float32).reshape(-1, 4) for x in (surf_map4, edge_map4, surf_q4, edge_q4)]
    q, t = np.ascontiguousarray(q_assoc, np.float64), np.ascontiguousarray(t_assoc, np.float64)
    srec, erec = np.zeros((max(a[2].shape[0], 1), 8)), np.zeros((max(a[3].shape[0], 1), 10))
    ns, ne = C.c_int(0), C.c_int(0)
    L.ref_backend_associate(_p(a[0]), a[0].shape[0], _p(a[1]), a[1].shape[0], _p(a[2]), a[2].shape[0], _p(a[3]), a[3].shape[0],
                            _p(q), _p(t), C.c_double(kd_max_radius), C.c_double(surf_dist_thres), C.c_double(lidar_const),
                            C.c_double(reflect_thres), _p(srec), C.byref(ns), _p(erec), C.byref(ne))
    return srec[:ns.value].copy(), erec[:ne.value].copy()


def backend_rows(flavour, surf_rec, edge_rec, qlb, tlb, t, q):
    """The residual blocks the reference's optimisation loop would add for those records (L/src/BackendFusion.cpp:936-972,
    R:836-866, count scaling included for 'rot'), evaluated at (t, q): (surf_rows, edge_rows), each (n,8) = r, dr/dt, dr/dq."""
    L = _lib("backend_L" if flavour == "livox" else "backend_R")
    s, e = np.ascontiguousarray(surf_rec, np.float64), np.ascontiguousarray(edge_rec, np.float64)
    sr, er = np.zeros((max(s.shape[0], 1), 8)), np.zeros((max(e.shape[0], 1), 8))
    a = [np.ascontiguousarray(x, np.float64) for x in (qlb, tlb, t, q)]
    L.ref_backend_rows(_p(s), s.shape[0], _p(e), e.shape[0], _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(sr), _p(er))
    return sr[:s.shape[0]], er[:e.shape[0]]


def format_convert(points19, stamp=0.0):
    """livoxLidarHandler of the reference's FormatConvert.cpp (compiled unmodified): CUSTOM_POINT records (19 bytes each, see
    oracle.CUSTOM_POINT) -> (n, 12) float32 rows of the published pcl::PointXYZINormal cloud."""
    L = _lib("format")
    raw = np.ascontiguousarray(points19).view(np.uint8).reshape(-1, 19)
    out = np.zeros((max(raw.shape[0], 1), 12), np.float32)
    n = L.ref_format_convert(_p(raw), raw.shape[0], C.c_double(stamp), _p(out))
    return out[:n]


def marg_lidar(surf_rec, edge_rec, qlb, tlb, t, q, pos, idx_t, idx_q):
    """ResidualBlockInfo::Evaluate + ThreadsConstructA of the reference's MarginalizationFactor.cpp (text sliced at build time)
    over the lidar blocks of one keyframe with CauchyLoss(1.0): returns (rows (n, 8) = robustified r, J_t, J_q — edge blocks
    first, then surf, like L/src/BackendFusion.cpp adds them —, A (pos, pos), b (pos))."""
    L = _lib("marg")
    s, e = np.ascontiguousarray(surf_rec, np.float64).reshape(-1, 8), np.ascontiguousarray(edge_rec, np.float64).reshape(-1, 10)
    rows = np.zeros((max(s.shape[0] + e.shape[0], 1), 8))
    A, b = np.zeros((pos, pos)), np.zeros(pos)
    a = [np.ascontiguousarray(x, np.float64) for x in (qlb, tlb, t, q)]
    L.ref_marg_lidar(_p(s), s.shape[0], _p(e), e.shape[0], _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), int(pos), int(idx_t), int(idx_q),
                     _p(rows), _p(A), _p(b))
    return rows[:s.shape[0] + e.shape[0]], A, b


class LocalMapSlice:
    """The back-end's local-map assembly from the reference text (transformCloud / buildLocalMapWithLandMark /
    downSampleCloud, L/src/BackendFusion.cpp:730-767, 1387-1484, 1486-1528; oracle/refshim/ref_localmap.cpp).
    keyframe(surf4, edge4) -> the four down-sampled clouds the matcher sees for this keyframe (x y z curvature rows):
    surf map, edge map, surf_last_ds, edge_last_ds; commit(pose_b) records the keyframe with its optimised body pose
    (qw qx qy qz x y z), as saveKeyFramesAndFactors does, before the next one arrives."""

    def __init__(self, local_map_width, surf_map_leaf, edge_map_leaf, surf_leaf, edge_leaf, q_bl, t_bl):
        L = self.lib = _lib("localmap")
        L.ref_lm_create.restype = C.c_void_p
        L.ref_lm_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.ref_lm_destroy.argtypes = [C.c_void_p]
        L.ref_lm_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_lm_get.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.ref_lm_commit.argtypes = [C.c_void_p, C.c_void_p]
        q, t = np.ascontiguousarray(q_bl, np.float64), np.ascontiguousarray(t_bl, np.float64)
        self.h = L.ref_lm_create(int(local_map_width), surf_map_leaf, edge_map_leaf, surf_leaf, edge_leaf, _p(q), _p(t))

    def close(self):
        if self.h:
            self.lib.ref_lm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def keyframe(self, surf4, edge4):
        s, e = np.ascontiguousarray(surf4, np.float32), np.ascontiguousarray(edge4, np.float32)
        sizes = np.zeros(4, np.int32)
        self.lib.ref_lm_keyframe(self.h, _p(s), s.shape[0], _p(e), e.shape[0], _p(sizes))
        outs = [np.zeros((int(n), 4), np.float32) for n in sizes]
        self.lib.ref_lm_get(self.h, *[_p(o) for o in outs])
        return dict(surf_map=outs[0], edge_map=outs[1], surf_ds=outs[2], edge_ds=outs[3])

    def commit(self, pose_b):
        p = np.ascontiguousarray(pose_b, np.float64)
        assert p.shape == (7,)
        self.lib.ref_lm_commit(self.h, _p(p))
