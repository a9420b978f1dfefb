"""ORACLE loader — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of oracle/liblili_oracle.so (the CPU restatement of the reference hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (lili_om_amd) never does.  PARITY: pinned bit for bit against the reference's own sources compiled as-is
(oracle/refshim/README.md, tests/test_reference_cpu.py) on every hot-path row; third-party internals (FLANN, Eigen, PCL, Ceres)
are restated and checked against semantics only (oracle/lo_math.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblili_oracle.so")

VARIANT_LIVOX, VARIANT_ROT, VARIANT_FRONTEND = 0, 1, 2
LOSS_NONE, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2


class Params(C.Structure):
    _fields_ = [("variant", C.c_int), ("loss", C.c_int), ("loss_a", C.c_double), ("lidar_const", C.c_double),
                ("kd_max_radius", C.c_double), ("edge_gate", C.c_double), ("surf_dist_thres", C.c_double),
                ("reflect_thres", C.c_double), ("surf_weight_min", C.c_double), ("edge_dist_max", C.c_double),
                ("q_lb", C.c_double * 4), ("t_lb", C.c_double * 3)]


def params(variant="rot", **kw):
    """Parameter sets of L/config/config_fr_iosb.yaml, R/config/config_fr_iosb.yaml."""
    p = Params()
    if variant == "livox":
        p.variant, p.loss, p.loss_a = VARIANT_LIVOX, LOSS_CAUCHY, 1.0
        p.lidar_const, p.kd_max_radius, p.edge_gate = 20.0, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.12, 15.0, 0.2, 0.0
        p.q_lb[:] = [0.0, 0.0, 0.0, 1.0]
        p.t_lb[:] = [-0.0265, 0.0202, 0.05309]
    elif variant == "rot":
        p.variant, p.loss, p.loss_a = VARIANT_ROT, LOSS_CAUCHY, 1.0
        p.lidar_const, p.kd_max_radius, p.edge_gate = 7.5, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.12, 0.0, 0.3, 0.1
        p.q_lb[:] = [0.7071, 0.0, 0.0, 0.7071]
        p.t_lb[:] = [-0.18, 0.0, -0.095]
    elif variant == "frontend":
        p.variant, p.loss, p.loss_a = VARIANT_FRONTEND, LOSS_HUBER, 0.1
        p.lidar_const, p.kd_max_radius, p.edge_gate = 1.0, 1.0, 1.0
        p.surf_dist_thres, p.reflect_thres, p.surf_weight_min, p.edge_dist_max = 0.06, 0.0, 0.4, 0.0
        p.q_lb[:] = [1.0, 0.0, 0.0, 0.0]
        p.t_lb[:] = [0.0, 0.0, 0.0]
    else:
        raise ValueError(variant)
    for k, v in kw.items():
        if k in ("q_lb", "t_lb"):
            getattr(p, k)[:] = list(v)
        else:
            setattr(p, k, v)
    return p


def build(force=False):
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in os.listdir(_HERE) if f.endswith((".cpp", ".h", "Makefile"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.lo_kdtree_build.restype = C.c_void_p
        _lib.lo_kdtree_build.argtypes = [C.c_void_p, C.c_int]
        _lib.lo_kdtree_free.argtypes = [C.c_void_p]
        for name in ("lo_associate_surf", "lo_associate_edge", "lo_gn_step", "lo_eig3"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _scale(scale):
    """scale: a plain factor, or (numerator, count) for the reference's ROT expressions (score * 1000 / N in double,
    intensity * 200 / N in float — R/src/BackendFusion.cpp:843,861)."""
    if isinstance(scale, tuple):
        return float(scale[0]), int(scale[1])
    return float(scale), 0


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class KdTree:
    """Exact kNN-5 (replaces pcl::KdTreeFLANN; L/src/BackendFusion.cpp:839-840,1541,1611)."""

    def __init__(self, xyz):
        self.xyz = _f32(xyz, 3)
        self.n = self.xyz.shape[0]
        self.h = lib().lo_kdtree_build(_p(self.xyz), self.n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_kdtree_free(C.c_void_p(self.h))
            self.h = None

    def knn5(self, q, nthreads=1):
        q = _f32(q, 3)
        m = q.shape[0]
        idx = np.empty((m, 5), np.int32)
        d2 = np.empty((m, 5), np.float32)
        lib().lo_knn5(C.c_void_p(self.h), _p(q), m, _p(idx), _p(d2), int(nthreads))
        return idx, d2


def knn5_brute(xyz, q):
    xyz, q = _f32(xyz, 3), _f32(q, 3)
    m = q.shape[0]
    idx = np.empty((m, 5), np.int32)
    d2 = np.empty((m, 5), np.float32)
    lib().lo_knn5_brute(_p(xyz), xyz.shape[0], _p(q), m, _p(idx), _p(d2))
    return idx, d2


def associate_surf(tree, map_refl, q_xyz, q_refl, pose_q, pose_t, P, nthreads=1):
    q_xyz = _f32(q_xyz, 3)
    n = q_xyz.shape[0]
    map_refl = None if map_refl is None else _f32(map_refl)
    q_refl = None if q_refl is None else _f32(q_refl)
    out = dict(valid=np.zeros(n, np.uint8), nn_idx=np.full((n, 5), -1, np.int32), nn_d2=np.zeros((n, 5), np.float32),
               cp=np.zeros((n, 3), np.float32), n=np.zeros((n, 3), np.float32), d=np.zeros(n, np.float32),
               score=np.zeros(n, np.float64))
    pq, pt = _f64(pose_q), _f64(pose_t)
    cnt = lib().lo_associate_surf(C.c_void_p(tree.h), _p(tree.xyz), _p(map_refl), tree.n, _p(q_xyz), _p(q_refl), n,
                                  _p(pq), _p(pt), C.byref(P), int(nthreads), _p(out["valid"]), _p(out["nn_idx"]),
                                  _p(out["nn_d2"]), _p(out["cp"]), _p(out["n"]), _p(out["d"]), _p(out["score"]))
    out["count"] = cnt
    return out


def associate_edge(tree, q_xyz, pose_q, pose_t, P, nthreads=1):
    q_xyz = _f32(q_xyz, 3)
    n = q_xyz.shape[0]
    out = dict(valid=np.zeros(n, np.uint8), nn_idx=np.full((n, 5), -1, np.int32), nn_d2=np.zeros((n, 5), np.float32),
               cp=np.zeros((n, 3), np.float32), a=np.zeros((n, 3), np.float32), b=np.zeros((n, 3), np.float32),
               s=np.zeros(n, np.float32))
    pq, pt = _f64(pose_q), _f64(pose_t)
    cnt = lib().lo_associate_edge(C.c_void_p(tree.h), _p(tree.xyz), tree.n, _p(q_xyz), n, _p(pq), _p(pt), C.byref(P),
                                  int(nthreads), _p(out["valid"]), _p(out["nn_idx"]), _p(out["nn_d2"]), _p(out["cp"]),
                                  _p(out["a"]), _p(out["b"]), _p(out["s"]))
    out["count"] = cnt
    return out


def pool_reset():
    """Drops the worker threads of the persistent pool (they are re-created lazily and inherit the affinity mask of the calling thread at that moment)."""
    lib().lo_pool_reset()


def register_surf(tree, q_xyz, t, q, P, scale_num=0.0, n_iters=10, nthreads=1):
    """One scan registration entirely in C (lo_register_surf): n_iters x (association pose, findCorrespondingSurfFeatures, linearisation, GN step) on the
    persistent pool.  Returns (t, q, steps applied, correspondences per iteration)."""
    q_xyz = _f32(q_xyz, 3)
    t, q = _f64(t).copy(), _f64(q).copy()
    counts = np.zeros(n_iters, np.int32)
    lib().lo_register_surf.restype = C.c_int
    applied = lib().lo_register_surf(C.c_void_p(tree.h), _p(tree.xyz), tree.n, _p(q_xyz), q_xyz.shape[0], _p(t), _p(q), C.byref(P), C.c_double(float(scale_num)),
                                     int(n_iters), int(nthreads), _p(counts))
    return t, q, int(applied), counts


def linearize_surf(rec, t, q, P, scale=1.0, nthreads=1):
    gram = np.zeros(64, np.float64)
    cost = C.c_double(0)
    cnt = C.c_int(0)
    t, q = _f64(t), _f64(q)
    if nthreads > 1:
        lib().lo_linearize_surf_mt(_p(rec["valid"]), _p(rec["cp"]), _p(rec["n"]), _p(rec["d"]), _p(rec["score"]),
                                   rec["valid"].shape[0], _p(t), _p(q), C.byref(P), C.c_double(_scale(scale)[0]), _scale(scale)[1], int(nthreads),
                                   _p(gram), C.byref(cost), C.byref(cnt))
        return gram.reshape(8, 8), cost.value, cnt.value
    lib().lo_linearize_surf(_p(rec["valid"]), _p(rec["cp"]), _p(rec["n"]), _p(rec["d"]), _p(rec["score"]),
                            rec["valid"].shape[0], _p(t), _p(q), C.byref(P), C.c_double(_scale(scale)[0]), _scale(scale)[1], _p(gram),
                            C.byref(cost), C.byref(cnt))
    return gram.reshape(8, 8), cost.value, cnt.value


def linearize_edge(rec, t, q, P, scale=1.0):
    gram = np.zeros(64, np.float64)
    cost = C.c_double(0)
    cnt = C.c_int(0)
    t, q = _f64(t), _f64(q)
    lib().lo_linearize_edge(_p(rec["valid"]), _p(rec["cp"]), _p(rec["a"]), _p(rec["b"]), _p(rec["s"]),
                            rec["valid"].shape[0], _p(t), _p(q), C.byref(P), C.c_double(_scale(scale)[0]), _scale(scale)[1], _p(gram),
                            C.byref(cost), C.byref(cnt))
    return gram.reshape(8, 8), cost.value, cnt.value


def gn_step(gram, t, q):
    """Returns (status, t_new, q_new, delta)."""
    g = _f64(gram).reshape(64)
    t = _f64(t).copy()
    q = _f64(q).copy()
    d = np.zeros(6)
    st = lib().lo_gn_step(_p(g), _p(t), _p(q), _p(d))
    return st, t, q, d


def eval_edge(t, q, cp, a, b, s):
    out = np.zeros(8)
    t, q = _f64(t), _f64(q)
    cp, a, b = _f32(cp), _f32(a), _f32(b)
    lib().lo_eval_edge(_p(t), _p(q), _p(cp), _p(a), _p(b), C.c_double(s), _p(out))
    return out


def eval_plane(t, q, cp, n, d, score, P, frontend=False):
    out = np.zeros(8)
    t, q = _f64(t), _f64(q)
    cp, n = _f32(cp), _f32(n)
    qlb = np.array(list(P.q_lb))
    tlb = np.array(list(P.t_lb))
    lib().lo_eval_plane(_p(t), _p(q), _p(cp), _p(n), C.c_float(d), C.c_double(score), _p(qlb), _p(tlb),
                        int(bool(frontend)), _p(out))
    return out


def eig3(A):
    A = _f64(A).reshape(9)
    ev = np.zeros(3)
    V = np.zeros(9)
    st = lib().lo_eig3(_p(A), _p(ev), _p(V))
    return st, ev, V.reshape(3, 3)


def lstsq53(A, b):
    A, b = _f64(A).reshape(15), _f64(b)
    x = np.zeros(3)
    lib().lo_lstsq53(_p(A), _p(b), _p(x))
    return x


def qrot(q, v):
    out = np.zeros(3)
    q, v = _f64(q), _f64(v)
    lib().lo_qrot(_p(q), _p(v), _p(out))
    return out


def loss(kind, a, s):
    rho = np.zeros(3)
    lib().lo_loss(int(kind), C.c_double(a), C.c_double(s), _p(rho))
    return rho


# ---------------------------------------------------------------------------------------------------------
# feature extraction (lo_extract.cpp)
# ---------------------------------------------------------------------------------------------------------
class RotParams(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("ds_rate", C.c_int), ("ds_v", C.c_float), ("near_thres", C.c_float),
                ("atan_mode", C.c_int), ("stable_sort", C.c_int)]


def rot_params(n_scans=64, ds_rate=4, ds_v=0.6, near_thres=3.0, atan_mode=0, stable_sort=0):
    """R/config/config_fr_iosb.yaml:13-14 (line_num 64, ds_rate 4), R/src/Preprocessing.cpp:14,281."""
    return RotParams(n_scans, ds_rate, ds_v, near_thres, atan_mode, stable_sort)


def extract_rot(pts_xyzi, q_imu=(1.0, 0, 0, 0), q_lb=(1.0, 0, 0, 0), P=None):
    """LOAM-style extractor of LiLi-OM-ROT.  pts_xyzi: (n,4) float32 in sensor firing order."""
    P = P or rot_params()
    pts = _f32(pts_xyzi, 4)
    n = pts.shape[0]
    cap = max(n, 1)
    full = np.zeros((cap, 4), np.float32); full_src = np.zeros(cap, np.int32)
    ring_start = np.zeros(P.n_scans, np.int32); ring_end = np.zeros(P.n_scans, np.int32)
    curv = np.zeros(cap, np.float32); label = np.zeros(cap, np.int32)
    edge_idx = np.zeros(cap, np.int32); sharp_idx = np.zeros(cap, np.int32); flat_idx = np.zeros(cap, np.int32)
    surf = np.zeros((cap, 4), np.float32); surf_cnt = np.zeros(cap, np.int32); lf_idx = np.zeros(cap, np.int32)
    ints = [C.c_int(0) for _ in range(7)]   # n_full, n_edge, n_sharp, n_flat, n_surf, n_lessflat, n_ties
    qi, ql = _f64(q_imu), _f64(q_lb)
    lib().lo_extract_rot.restype = C.c_int
    rc = lib().lo_extract_rot(_p(pts), n, _p(qi), _p(ql), C.byref(P), _p(full), _p(full_src), C.byref(ints[0]),
                              _p(ring_start), _p(ring_end), _p(curv), _p(label), _p(edge_idx), C.byref(ints[1]),
                              _p(sharp_idx), C.byref(ints[2]), _p(flat_idx), C.byref(ints[3]), _p(surf), _p(surf_cnt),
                              C.byref(ints[4]), _p(lf_idx), C.byref(ints[5]), C.byref(ints[6]))
    if rc != 0:
        raise RuntimeError(f"lo_extract_rot failed ({rc})")
    nf, ne, ns, nfl, nsu, nlf, nt = [v.value for v in ints]
    return dict(full=full[:nf], full_src=full_src[:nf], ring_start=ring_start, ring_end=ring_end, curvature=curv[:nf],
                label=label[:nf], edge_idx=edge_idx[:ne], sharp_idx=sharp_idx[:ns], flat_idx=flat_idx[:nfl],
                surf=surf[:nsu], surf_cnt=surf_cnt[:nsu], lessflat_idx=lf_idx[:nlf], n_ties=nt)


def voxel_grid(pts_xyzi, leaf, stable=False):
    pts = _f32(pts_xyzi, 4)
    n = pts.shape[0]
    out = np.zeros((max(n, 1), 4), np.float32)
    cnt = np.zeros(max(n, 1), np.int32)
    lib().lo_voxel_grid.restype = C.c_int
    m = lib().lo_voxel_grid(_p(pts), n, C.c_float(leaf), int(bool(stable)), _p(out), _p(cnt))
    return out[:m], cnt[:m]


class LivoxParams(C.Structure):
    _fields_ = [("surf_thres", C.c_double), ("edge_thres", C.c_double), ("near_thres", C.c_float)]


def livox_params(surf_thres=0.28, edge_thres=4.0, near_thres=0.1):
    """L/config/config_fr_iosb.yaml:5-6, L/src/Preprocessing.cpp:226."""
    return LivoxParams(surf_thres, edge_thres, near_thres)


def extract_livox(pts5, q_imu=(1.0, 0, 0, 0), P=None):
    """Livox Horizon extractor.  pts5: (n,5) float32 = x, y, z, intensity (line + 0.1 t), curvature (0.1 reflectivity)."""
    P = P or livox_params()
    pts = _f32(pts5, 5)
    n = pts.shape[0]
    cap = max(n, 1)
    cut = np.zeros((cap, 8), np.float32); cut_src = np.zeros(cap, np.int32)
    edge = np.zeros((24000, 8), np.float32); edge_cell = np.zeros(24000, np.int32)
    surf = np.zeros((24000, 8), np.float32); surf_cell = np.zeros(24000, np.int32)
    cell_src = np.zeros(24000, np.int32)
    nc, ne, ns = C.c_int(0), C.c_int(0), C.c_int(0)
    qi = _f64(q_imu)
    lib().lo_extract_livox.restype = C.c_int
    rc = lib().lo_extract_livox(_p(pts), n, _p(qi), C.byref(P), _p(cut), _p(cut_src), C.byref(nc), _p(edge), _p(edge_cell),
                                C.byref(ne), _p(surf), _p(surf_cell), C.byref(ns), _p(cell_src))
    if rc != 0:
        raise RuntimeError("lo_extract_livox failed")
    return dict(cutted=cut[:nc.value], cut_src=cut_src[:nc.value], edge=edge[:ne.value], edge_cell=edge_cell[:ne.value],
                surf=surf[:ns.value], surf_cell=surf_cell[:ns.value], cell_src=cell_src.reshape(6, 4000))


# ------------------------------------------------------------------------------------------------
# callers / data formats either side of the path — small, so numpy / pure Python
# ------------------------------------------------------------------------------------------------
CUSTOM_POINT = np.dtype([("offset_time", "<u4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
                         ("reflectivity", "u1"), ("tag", "u1"), ("line", "u1")])   # livox_ros_driver/CustomPoint, 19 B


def livox_custom_to_cloud(points):
    """livoxLidarHandler (L/src/FormatConvert.cpp:11-35) on a CUSTOM_POINT array -> (n, 12) float32 rows of
    pcl::PointXYZINormal (x y z 1 | 0 0 0 0 | intensity curvature 0 0).
      float s = float(offset_time / (float)time_end);   pt.intensity = line + s*0.1;   pt.curvature = 0.1 * reflectivity;"""
    pts = np.asarray(points, dtype=CUSTOM_POINT)
    n = pts.shape[0]
    out = np.zeros((n, 12), np.float32)
    if n == 0:
        return out
    time_end = np.float32(pts["offset_time"][-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        s = pts["offset_time"].astype(np.float32) / time_end                       # float / float
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = pts["x"], pts["y"], pts["z"], 1.0
    out[:, 8] = (pts["line"].astype(np.float64) + s.astype(np.float64) * 0.1).astype(np.float32)
    out[:, 9] = (0.1 * pts["reflectivity"].astype(np.float64)).astype(np.float32)
    return out


class ImuIntegrator:
    """Preprocessing's gyro integration (L/src/Preprocessing.cpp:129-171 processIMU/solveRotation, 176-191 imuHandler,
    232-234, 403; deltaQ: L/include/utils/math_tools.h:125-138).  Pure-Python doubles, operation for operation."""

    def __init__(self):
        self.idx = 0
        self.t_cur = -1.0
        self.gyr0 = [0.0, 0.0, 0.0]
        self.first = False

    def _solve_rotation(self, q, dt, w):
        th = [0.5 * (self.gyr0[k] + w[k]) * dt for k in range(3)]
        bw, bx, by, bz = 1.0, th[0] / 2.0, th[1] / 2.0, th[2] / 2.0
        aw, ax, ay, az = q
        q[0] = aw * bw - ax * bx - ay * by - az * bz
        q[1] = aw * bx + ax * bw + ay * bz - az * by
        q[2] = aw * by + ay * bw + az * bx - ax * bz
        q[3] = aw * bz + az * bw + ax * by - ay * bx
        self.gyr0 = [float(w[0]), float(w[1]), float(w[2])]

    def integrate(self, stamps, gyr, t_scan_next):
        stamps = [float(x) for x in stamps]
        gyr = [[float(v) for v in row] for row in gyr]
        n = len(stamps)
        q = [1.0, 0.0, 0.0, 0.0]
        if n > 0:
            if self.t_cur < 0:
                self.t_cur = stamps[0]
            if not self.first:
                self.first = True
                self.gyr0 = list(gyr[0])
            r = [0.0, 0.0, 0.0]
            i = self.idx
            if i >= n:
                i -= 1
            while stamps[i] < t_scan_next:
                t = stamps[i]
                if self.t_cur < 0:
                    self.t_cur = t
                dt = t - self.t_cur
                self.t_cur = stamps[i]
                r = list(gyr[i])
                self._solve_rotation(q, dt, r)
                i += 1
                if i >= n:
                    break
            if i < n:
                dt1 = t_scan_next - self.t_cur
                dt2 = stamps[i] - t_scan_next
                with np.errstate(divide="ignore", invalid="ignore"):      # C semantics for a zero denominator
                    w1 = float(np.float64(dt2) / np.float64(dt1 + dt2))
                    w2 = float(np.float64(dt1) / np.float64(dt1 + dt2))
                r = [w1 * r[k] + w2 * gyr[i][k] for k in range(3)]
                self._solve_rotation(q, dt1, r)
            self.t_cur = t_scan_next
            self.idx = i
        if any(v != v for v in q):
            q = [1.0, 0.0, 0.0, 0.0]
        return np.array(q, np.float64)


def marg_accumulate(J, r, pos, idx_t, idx_q, A=None, b=None):
    """ThreadsConstructA (L/src/MarginalizationFactor.cpp:3-29) for lidar blocks with parameter blocks (t[3], q[4]):
    J (n,7) robustified 1x7 Jacobians (t, then q = w,x,y,z), r (n,) robustified residuals.  size 4 -> rightCols(3)."""
    A = np.zeros((pos, pos)) if A is None else A
    b = np.zeros(pos) if b is None else b
    for k in range(J.shape[0]):
        jt = J[k, 0:3].reshape(1, 3)
        jq = J[k, 4:7].reshape(1, 3)
        blocks = ((idx_t, jt), (idx_q, jq))
        for i in range(2):
            ii, ji = blocks[i]
            for j in range(i, 2):
                ij, jj = blocks[j]
                A[ii:ii + 3, ij:ij + 3] += ji.T @ jj
                if i != j:
                    A[ij:ij + 3, ii:ii + 3] = A[ii:ii + 3, ij:ij + 3].T
            b[ii:ii + 3] += ji.T[:, 0] * r[k]
    return A, b


def linearize_rows(rec, t, q, P, scale=1.0, kind="surf"):
    """Per-residual robustified rows [J(7) r] of the valid records, in record order (what ResidualBlockInfo::Evaluate
    leaves in jacobians / residuals, L/src/MarginalizationFactor.cpp:31-71)."""
    n = rec["valid"].shape[0]
    rows = np.zeros((n, 8), np.float64)
    cnt = C.c_int(0)
    t, q = _f64(t), _f64(q)
    if kind == "surf":
        lib().lo_rows_surf(_p(rec["valid"]), _p(rec["cp"]), _p(rec["n"]), _p(rec["d"]), _p(rec["score"]), n, _p(t), _p(q),
                           C.byref(P), C.c_double(_scale(scale)[0]), _scale(scale)[1], _p(rows), C.byref(cnt))
    else:
        lib().lo_rows_edge(_p(rec["valid"]), _p(rec["cp"]), _p(rec["a"]), _p(rec["b"]), _p(rec["s"]), n, _p(t), _p(q),
                           C.byref(P), C.c_double(_scale(scale)[0]), _scale(scale)[1], _p(rows), C.byref(cnt))
    return rows[:cnt.value]


def _qmul(a, b):
    """Eigen quaternion product, generic path (w, x, y, z), evaluated left to right."""
    aw, ax, ay, az = (float(v) for v in a)
    bw, bx, by, bz = (float(v) for v in b)
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx], np.float64)


def transform_cloud(pts_xyza, q, t):
    """transformCloud (L/src/BackendFusion.cpp:730-767): f32 point -> f64, q * p + t with Eigen's _transformVector
    (uv = 2 u x p; p + w uv + u x uv), -> f32; the aux column (curvature) rides along."""
    pts = _f32(pts_xyza, 4)
    q, t = _f64(q), _f64(t)
    v = pts[:, :3].astype(np.float64)
    u = q[1:4]
    uv = np.cross(u, v)
    uv = uv + uv
    w = (v + uv * q[0]) + np.cross(u, uv)
    out = pts.copy()
    out[:, :3] = (w + t).astype(np.float32)
    return out


class LocalMapAssembly:
    """buildLocalMapWithLandMark + downSampleCloud (L/src/BackendFusion.cpp:1387-1484, 1486-1511) for one feature kind:
    keyframe(features) -> (local map after VoxelGrid(map_leaf), features after VoxelGrid(leaf)); commit(pose_b) stores the
    down-sampled features with the body pose (qw qx qy qz x y z), as saveKeyFramesAndFactors (L:1683-1760) does.
    First keyframe: the map is the keyframe's own RAW features moved by T_bl (L:1390-1404).  While the ring is short it is
    rebuilt from the newest `width` keyframes (L:1407-1443); once full, the oldest leaves and the newest enters (L:1445-1477)."""

    def __init__(self, width, map_leaf, leaf, q_bl, t_bl, stable=False):
        self.width, self.map_leaf, self.leaf, self.stable = int(width), float(map_leaf), float(leaf), stable
        self.q_bl, self.t_bl = _f64(q_bl), _f64(t_bl)
        self.frames, self.poses, self.recent, self.latest, self._last_ds = [], [], [], 0, None

    def lidar_pose(self, pose_b):
        q_po, t_po = _f64(pose_b[:4]), _f64(pose_b[4:7])
        return _qmul(q_po, self.q_bl), qrot(q_po, self.t_bl) + t_po                       # L:1425-1426 / 1462-1463

    def keyframe(self, feats_xyza):
        feats = _f32(feats_xyza, 4)
        if not self.poses:
            raw = transform_cloud(feats, self.q_bl, self.t_bl)
        else:
            if len(self.recent) < self.width:
                self.recent = []
                for i in range(len(self.poses) - 1, -1, -1):
                    q, t = self.lidar_pose(self.poses[i])
                    self.recent.insert(0, transform_cloud(self.frames[i], q, t))
                    if len(self.recent) >= self.width:
                        break
            elif self.latest != len(self.poses) - 1:
                self.recent.pop(0)
                self.latest = len(self.poses) - 1
                q, t = self.lidar_pose(self.poses[self.latest])
                self.recent.append(transform_cloud(self.frames[self.latest], q, t))
            raw = np.concatenate(self.recent, 0) if self.recent else np.zeros((0, 4), np.float32)
        self.raw = raw
        self._last_ds = voxel_grid(feats, self.leaf, stable=self.stable)[0]
        return voxel_grid(raw, self.map_leaf, stable=self.stable)[0], self._last_ds

    def commit(self, pose_b):
        self.frames.append(self._last_ds.copy())
        self.poses.append(_f64(pose_b))
