"""Synthetic scans and local maps for parity tests and bench.py.

No rosbags exist offline, so every workload is generated from a fixed seed:

* ``outdoor_scene`` / ``make_workload``  — BASELINE config 2 "variant A": undulating ground, a 25 m
  lattice of 10x10x8 m box buildings, poles at the lattice corners and a perimeter wall, ray-cast
  analytically from a 64-ring spinning sensor (elevation table implied by
  R/src/Preprocessing.cpp:332-337), range noise N(0, 0.02 m); the map is the same scene sampled at
  the reference's voxel density (0.4 m surf / 0.2 m edge leaves, one centroid per voxel, App. B2).
* ``make_room`` — a small closed room for fast CPU-side parity cases.

Pure numpy; deterministic for a given seed.  This module is data generation only — it contains no
part of the hot path and no oracle code.
"""
import math
import sys

import numpy as np

SEED_SCENE = 0x11110
SEED_POSE = 0x22220


# ------------------------------------------------------------------------------------------------
# sensor model
# ------------------------------------------------------------------------------------------------
def hdl64_elevations_deg():
    """Ring id -> elevation such that R/src/Preprocessing.cpp:333-336 maps it back to the same id."""
    ids = np.arange(64)
    upper = 2.0 - ids / 3.0                      # id = (2 - angle) * 3      for angle >= -8.83
    lower = -8.83 - (ids - 32) / 2.0             # id = 32 + (-8.83-angle)*2 for angle <  -8.83
    return np.where(ids <= 32, upper, lower)


def hdl32_elevations_deg():
    """Ring id -> elevation at the centre of the bin R/src/Preprocessing.cpp:325-326 maps back to the same id: id = int((angle + 92/3) * 3/4)
    (truncation, no + 0.5), i.e. -30.67 .. +10.67 deg in steps of 4/3 deg — an HDL-32E."""
    return -92.0 / 3.0 + (np.arange(32) + 0.5) * 4.0 / 3.0


def spinning_rays(n_az=3125, elev_deg=None, az0=0.0):
    """Azimuth-major firing order (all rings per azimuth step), clockwise like a Velodyne:
    ori = -atan2(y, x) increases with time (R/src/Preprocessing.cpp:285-294,349)."""
    if elev_deg is None:
        elev_deg = hdl64_elevations_deg()
    el = np.deg2rad(np.asarray(elev_deg, np.float64))
    alpha = az0 + 2.0 * np.pi * np.arange(n_az) / n_az      # ori of each azimuth step
    a = np.repeat(alpha, len(el))
    e = np.tile(el, n_az)
    ring = np.tile(np.arange(len(el)), n_az)
    d = np.stack([np.cos(e) * np.cos(-a), np.cos(e) * np.sin(-a), np.sin(e)], axis=1)
    rel_time = np.repeat(np.arange(n_az) / n_az, len(el))
    return d, ring.astype(np.int32), rel_time


# ------------------------------------------------------------------------------------------------
# outdoor scene (variant A)
# ------------------------------------------------------------------------------------------------
class OutdoorScene:
    pitch = 25.0          # lattice pitch
    bsize = 10.0          # building footprint
    bheight = 8.0
    bbase = -1.0          # building base below the lowest ground
    pole_r = 0.15
    pole_h = 6.0
    wall = 140.0          # perimeter wall half-size (all rays return within 200 m)
    wall_h = 12.0

    @staticmethod
    def ground(x, y):
        return 0.5 * np.sin(x / 40.0) * np.cos(y / 55.0)

    # ---------------- ray casting ----------------
    def raycast(self, origin, dirs, t_max=260.0):
        o = np.asarray(origin, np.float64)
        d = np.asarray(dirs, np.float64)
        n = d.shape[0]
        t_hit = np.full(n, np.inf)
        # ground: march 1 m steps to bracket the first crossing, then bisect
        ts = np.arange(0.0, t_max + 1.0, 1.0)
        lo = np.zeros(n)
        hi = np.full(n, np.nan)
        found = np.zeros(n, bool)
        prev = o[2] - self.ground(o[0], o[1])
        prev = np.full(n, prev)
        for t in ts[1:]:
            x, y, z = o[0] + t * d[:, 0], o[1] + t * d[:, 1], o[2] + t * d[:, 2]
            cur = z - self.ground(x, y)
            cross = (~found) & (prev > 0) & (cur <= 0)
            lo[cross] = t - 1.0
            hi[cross] = t
            found |= cross
            prev = cur
        idx = np.nonzero(found)[0]
        a, b = lo[idx], hi[idx]
        for _ in range(40):
            m = 0.5 * (a + b)
            x, y, z = o[0] + m * d[idx, 0], o[1] + m * d[idx, 1], o[2] + m * d[idx, 2]
            above = z - self.ground(x, y) > 0
            a = np.where(above, m, a)
            b = np.where(above, b, m)
        t_hit[idx] = 0.5 * (a + b)
        # perimeter walls (inside faces of a big box)
        W = self.wall
        with np.errstate(divide="ignore", invalid="ignore"):
            for ax in (0, 1):
                other = 1 - ax
                for sgn in (-1.0, 1.0):
                    t = (sgn * W - o[ax]) / d[:, ax]
                    p_o = o[other] + t * d[:, other]
                    z = o[2] + t * d[:, 2]
                    ok = (t > 0) & (np.abs(p_o) <= W) & (z <= self.wall_h)
                    t_hit = np.where(ok & (t < t_hit), t, t_hit)
        # lattice traversal (Amanatides-Woo) for buildings and poles
        P = self.pitch
        ix = np.floor(o[0] / P) * np.ones(n)
        iy = np.floor(o[1] / P) * np.ones(n)
        sx = np.where(d[:, 0] >= 0, 1.0, -1.0)
        sy = np.where(d[:, 1] >= 0, 1.0, -1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            tdx = np.abs(P / d[:, 0])
            tdy = np.abs(P / d[:, 1])
            tmx = np.where(d[:, 0] >= 0, ((ix + 1) * P - o[0]) / d[:, 0], (ix * P - o[0]) / d[:, 0])
            tmy = np.where(d[:, 1] >= 0, ((iy + 1) * P - o[1]) / d[:, 1], (iy * P - o[1]) / d[:, 1])
        tmx = np.where(np.isfinite(tmx), tmx, np.inf)
        tmy = np.where(np.isfinite(tmy), tmy, np.inf)
        tdx = np.where(np.isfinite(tdx), tdx, np.inf)
        tdy = np.where(np.isfinite(tdy), tdy, np.inf)
        nsteps = int(2 * (self.wall * 2 / P) + 4)
        t_enter = np.zeros(n)
        for _ in range(nsteps):
            active = t_enter < t_hit
            if not active.any():
                break
            # building in cell (ix, iy): box centred in the cell
            cx, cy = (ix + 0.5) * P, (iy + 0.5) * P
            h = self.bsize / 2
            t_b = self._ray_box(o, d, cx - h, cx + h, cy - h, cy + h, self.bbase, self.bheight)
            t_hit = np.where(active & (t_b < t_hit), t_b, t_hit)
            # poles at the 4 corners of the cell (none at the origin corner where the sensor stands)
            for ax_, ay_ in ((0, 0), (1, 0), (0, 1), (1, 1)):
                px, py = (ix + ax_) * P, (iy + ay_) * P
                t_p = self._ray_pole(o, d, px, py)
                t_p = np.where((px == 0) & (py == 0), np.inf, t_p)
                t_hit = np.where(active & (t_p < t_hit), t_p, t_hit)
            stepx = tmx < tmy
            t_enter = np.where(stepx, tmx, tmy)
            ix = np.where(stepx, ix + sx, ix)
            iy = np.where(stepx, iy, iy + sy)
            tmx = np.where(stepx, tmx + tdx, tmx)
            tmy = np.where(stepx, tmy, tmy + tdy)
        return t_hit

    @staticmethod
    def _ray_box(o, d, x0, x1, y0, y1, z0, z1):
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            tx0, tx1 = (x0 - o[0]) * inv[:, 0], (x1 - o[0]) * inv[:, 0]
            ty0, ty1 = (y0 - o[1]) * inv[:, 1], (y1 - o[1]) * inv[:, 1]
            tz0, tz1 = (z0 - o[2]) * inv[:, 2], (z1 - o[2]) * inv[:, 2]
        tmin = np.maximum(np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1)), np.minimum(tz0, tz1))
        tmax = np.minimum(np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1)), np.maximum(tz0, tz1))
        hit = (tmax >= tmin) & (tmin > 0)
        return np.where(hit, tmin, np.inf)

    def _ray_pole(self, o, d, px, py):
        ox, oy = o[0] - px, o[1] - py
        a = d[:, 0] ** 2 + d[:, 1] ** 2
        b = 2 * (ox * d[:, 0] + oy * d[:, 1])
        c = ox * ox + oy * oy - self.pole_r ** 2
        disc = b * b - 4 * a * c
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / (2 * a)
        z = o[2] + t * d[:, 2]
        gz = self.ground(px, py)
        ok = (disc >= 0) & (t > 0) & (z <= gz + self.pole_h)
        return np.where(ok, t, np.inf)

    # ---------------- map sampling ----------------
    def sample_surfaces(self, hx, hy, spacing, rng):
        """Jittered-lattice samples of every surface with |x|<=hx, |y|<=hy."""
        out = []

        def lattice(u0, u1, v0, v1):
            nu = max(1, int(round((u1 - u0) / spacing)))
            nv = max(1, int(round((v1 - v0) / spacing)))
            u = u0 + (np.arange(nu) + 0.5) * (u1 - u0) / nu
            v = v0 + (np.arange(nv) + 0.5) * (v1 - v0) / nv
            U, V = np.meshgrid(u, v, indexing="ij")
            U = U.ravel() + rng.uniform(-0.3, 0.3, U.size) * spacing
            V = V.ravel() + rng.uniform(-0.3, 0.3, V.size) * spacing
            return U, V

        # ground (not under buildings)
        X, Y = lattice(-hx, hx, -hy, hy)
        P, h = self.pitch, self.bsize / 2
        fx = np.abs((X / P - np.floor(X / P)) - 0.5) * P
        fy = np.abs((Y / P - np.floor(Y / P)) - 0.5) * P
        keep = ~((fx < h) & (fy < h))
        X, Y = X[keep], Y[keep]
        out.append(np.stack([X, Y, self.ground(X, Y)], 1))
        # buildings
        i0, i1 = int(math.floor(-hx / P)), int(math.ceil(hx / P))
        j0, j1 = int(math.floor(-hy / P)), int(math.ceil(hy / P))
        ci, cj = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
        cx, cy = (ci.ravel() + 0.5) * P, (cj.ravel() + 0.5) * P
        inside = (np.abs(cx) + h <= hx) & (np.abs(cy) + h <= hy)
        cx, cy = cx[inside], cy[inside]
        nb = cx.size
        # walls: template lattice on a (10 x 9) rectangle, replicated per building and face
        U, V = lattice(-h, h, self.bbase + 0.6, self.bheight)     # skip the part buried below ground
        for (ax, sg) in ((0, -1), (0, 1), (1, -1), (1, 1)):
            uu = np.tile(U, nb) + rng.uniform(-0.05, 0.05, U.size * nb)
            vv = np.tile(V, nb)
            bx = np.repeat(cx, U.size)
            by = np.repeat(cy, U.size)
            if ax == 0:
                pts = np.stack([bx + sg * h, by + uu, vv], 1)
            else:
                pts = np.stack([bx + uu, by + sg * h, vv], 1)
            pts = pts[pts[:, 2] > self.ground(pts[:, 0], pts[:, 1]) + 0.05]
            out.append(pts)
        U, V = lattice(-h, h, -h, h)                              # roofs
        out.append(np.stack([np.repeat(cx, U.size) + np.tile(U, nb), np.repeat(cy, U.size) + np.tile(V, nb),
                             np.full(U.size * nb, self.bheight)], 1))
        # perimeter wall
        W = self.wall
        if hx >= W and hy >= W:
            U, V = lattice(-W, W, -0.4, self.wall_h)
            for (ax, sg) in ((0, -1), (0, 1), (1, -1), (1, 1)):
                pts = np.stack([np.full(U.size, sg * W), U, V], 1) if ax == 0 else np.stack([U, np.full(U.size, sg * W), V], 1)
                out.append(pts[pts[:, 2] > self.ground(pts[:, 0], pts[:, 1]) + 0.05])
        # poles (8 points per 0.4 m ring)
        pi_, pj_ = np.meshgrid(np.arange(i0, i1 + 1), np.arange(j0, j1 + 1), indexing="ij")
        px, py = pi_.ravel() * P, pj_.ravel() * P
        ok = (np.abs(px) <= hx) & (np.abs(py) <= hy) & ~((px == 0) & (py == 0))
        px, py = px[ok], py[ok]
        zs = np.arange(0.2, self.pole_h, spacing)
        ang = np.arange(8) * (2 * np.pi / 8)
        PX = np.repeat(px, zs.size * 8)
        PY = np.repeat(py, zs.size * 8)
        ZZ = np.tile(np.repeat(zs, 8), px.size)
        AA = np.tile(ang, px.size * zs.size)
        out.append(np.stack([PX + self.pole_r * np.cos(AA), PY + self.pole_r * np.sin(AA),
                             self.ground(PX, PY) + ZZ], 1))
        return np.concatenate(out, 0)

    def sample_edges(self, hx, hy, spacing, rng):
        """Vertical building corners and pole axes (the 'edge' local map)."""
        P, h = self.pitch, self.bsize / 2
        i0, i1 = int(math.floor(-hx / P)), int(math.ceil(hx / P))
        j0, j1 = int(math.floor(-hy / P)), int(math.ceil(hy / P))
        ci, cj = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
        cx, cy = (ci.ravel() + 0.5) * P, (cj.ravel() + 0.5) * P
        inside = (np.abs(cx) + h <= hx) & (np.abs(cy) + h <= hy)
        cx, cy = cx[inside], cy[inside]
        zs = np.arange(0.5, self.bheight, spacing)
        out = []
        for sx in (-1, 1):
            for sy in (-1, 1):
                X = np.repeat(cx + sx * h, zs.size)
                Y = np.repeat(cy + sy * h, zs.size)
                out.append(np.stack([X, Y, np.tile(zs, cx.size)], 1))
        pi_, pj_ = np.meshgrid(np.arange(i0, i1 + 1), np.arange(j0, j1 + 1), indexing="ij")
        px, py = pi_.ravel() * P, pj_.ravel() * P
        ok = (np.abs(px) <= hx) & (np.abs(py) <= hy) & ~((px == 0) & (py == 0))
        px, py = px[ok], py[ok]
        zp = np.arange(0.2, self.pole_h, spacing)
        X, Y = np.repeat(px, zp.size), np.repeat(py, zp.size)
        out.append(np.stack([X, Y, self.ground(X, Y) + np.tile(zp, px.size)], 1))
        pts = np.concatenate(out, 0)
        pts += rng.normal(0, 0.01, pts.shape)
        return pts


def voxel_centroids(pts, leaf):
    """One centroid per occupied voxel, ascending voxel id — the effect of pcl::VoxelGrid the design
    relies on (<= 1 map point per leaf voxel).  f64 accumulation (generator only)."""
    pts = np.asarray(pts, np.float64)
    ijk = np.floor(pts / leaf).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    uniq, inv = np.unique(key, return_inverse=True)
    cnt = np.bincount(inv)
    out = np.empty((uniq.size, 3))
    for k in range(3):
        out[:, k] = np.bincount(inv, weights=pts[:, k]) / cnt
    return out


def perturbed_pose(t_true, q_true, rng, dt=0.3, dang_deg=2.0):
    """Initial pose error of config 2: 0.3 m / 2 deg."""
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    half = math.radians(dang_deg) / 2
    dq = np.array([math.cos(half), *(math.sin(half) * axis)])
    dirn = rng.normal(size=3)
    dirn /= np.linalg.norm(dirn)
    return np.asarray(t_true, np.float64) + dt * dirn, quat_mul(dq, np.asarray(q_true, np.float64))


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def quat_rot(q, v):
    """Unit-quaternion rotation of row vectors (generator-side only)."""
    w, u = q[0], np.asarray(q[1:4])
    v = np.asarray(v, np.float64)
    uv = 2 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def make_workload(n_map=5_000_000, n_az=3125, seed=SEED_SCENE, half_extent=(460.0, 380.0), verbose=False):
    """BASELINE config 2 (variant A).  Returns float32 arrays:
    map_xyz (n_map,3), edge_map_xyz, scan_xyz (n_az*64,3) in the LiDAR frame (azimuth-major order),
    scan_ring, scan_reltime, and the true LiDAR pose in the map frame."""
    rng = np.random.default_rng(seed)
    sc = OutdoorScene()
    hx, hy = half_extent
    raw = sc.sample_surfaces(hx, hy, 0.4, rng)
    surf = voxel_centroids(raw, 0.4)
    if surf.shape[0] < n_map:
        raise ValueError(f"extent {half_extent} gives only {surf.shape[0]} map points (< {n_map}); enlarge it")
    # 'extent tuned': keep the n_map points nearest to the centre in the (|x|/hx, |y|/hy) max-norm
    r = np.maximum(np.abs(surf[:, 0]) / hx, np.abs(surf[:, 1]) / hy)
    keep = np.sort(np.argpartition(r, n_map - 1)[:n_map])
    extent_used = float(r[keep].max())
    surf = surf[keep]
    edge = voxel_centroids(sc.sample_edges(min(hx, 200.0), min(hy, 200.0), 0.2, rng), 0.2)
    origin = np.array([0.0, 0.0, 1.8])
    dirs, ring, rel = spinning_rays(n_az)
    t = sc.raycast(origin, dirs)
    ok = np.isfinite(t)
    rngn = np.random.default_rng(seed + 1)
    t = t + rngn.normal(0, 0.02, t.shape)
    pts = dirs * t[:, None]
    if verbose:
        print(f"[synth] map {surf.shape[0]} pts (extent frac {extent_used:.3f} of {half_extent}), edge map {edge.shape[0]}, "
              f"scan {int(ok.sum())}/{ok.size} returns, range {np.nanmin(t[ok]):.1f}..{np.nanmax(t[ok]):.1f} m", file=sys.stderr)
    return dict(map_xyz=surf.astype(np.float32), edge_map_xyz=edge.astype(np.float32),
                scan_xyz=pts[ok].astype(np.float32), scan_ring=ring[ok], scan_reltime=rel[ok].astype(np.float32),
                lidar_t=origin, lidar_q=np.array([1.0, 0, 0, 0]), seed=seed, half_extent=half_extent,
                extent_frac=extent_used)


# ------------------------------------------------------------------------------------------------
# small closed room for quick parity cases
# ------------------------------------------------------------------------------------------------
def make_room(seed=1, size=(24.0, 18.0, 6.0), leaf=0.4, n_query=4000, n_edge_query=400, noise=0.01):
    """Room with floor, ceiling, 4 walls and 8 vertical pillars' edges.  Returns map_xyz, map_refl,
    edge_map_xyz, q_xyz (queries in the LiDAR frame), q_refl, eq_xyz (edge queries), true pose."""
    rng = np.random.default_rng(seed)
    sx, sy, sz = size
    faces = []

    def grid(u0, u1, v0, v1):
        nu, nv = int((u1 - u0) / leaf), int((v1 - v0) / leaf)
        U, V = np.meshgrid(u0 + (np.arange(nu) + 0.5) * leaf, v0 + (np.arange(nv) + 0.5) * leaf, indexing="ij")
        j = lambda a: a.ravel() + rng.uniform(-0.12, 0.12, a.size)
        return j(U), j(V)

    U, V = grid(-sx / 2, sx / 2, -sy / 2, sy / 2)
    faces.append(np.stack([U, V, np.zeros_like(U)], 1))
    U, V = grid(-sx / 2, sx / 2, -sy / 2, sy / 2)
    faces.append(np.stack([U, V, np.full_like(U, sz)], 1))
    for sg in (-1, 1):
        U, V = grid(-sy / 2, sy / 2, 0, sz)
        faces.append(np.stack([np.full_like(U, sg * sx / 2), U, V], 1))
        U, V = grid(-sx / 2, sx / 2, 0, sz)
        faces.append(np.stack([U, np.full_like(U, sg * sy / 2), V], 1))
    surf = np.concatenate(faces, 0) + rng.normal(0, 0.004, (sum(f.shape[0] for f in faces), 3))
    surf = voxel_centroids(surf, leaf)
    # edges: the 4 vertical room corners + 4 free-standing pillars
    ex = [(-sx / 2, -sy / 2), (sx / 2, -sy / 2), (-sx / 2, sy / 2), (sx / 2, sy / 2),
          (-sx / 4, -sy / 4), (sx / 4, -sy / 4), (-sx / 4, sy / 4), (sx / 4, sy / 4)]
    zs = np.arange(0.1, sz, 0.2)
    edge = np.concatenate([np.stack([np.full_like(zs, x), np.full_like(zs, y), zs], 1) for x, y in ex], 0)
    edge = edge + rng.normal(0, 0.01, edge.shape)
    # true LiDAR pose inside the room
    t_true = np.array([1.3, -0.7, 1.6])
    ang = math.radians(25.0)
    q_true = np.array([math.cos(ang / 2), 0.0, 0.0, math.sin(ang / 2)])
    # queries: surface points (+noise) expressed in the LiDAR frame, some far-off junk included
    pick = rng.choice(surf.shape[0], n_query, replace=True)
    qw = surf[pick] + rng.normal(0, noise, (n_query, 3)) + rng.uniform(-0.2, 0.2, (n_query, 3))
    qw[: n_query // 20] += rng.uniform(1.0, 3.0, (n_query // 20, 3))            # junk: fails the gates
    q_inv = q_true * np.array([1, -1, -1, -1])
    q_local = quat_rot(q_inv, qw - t_true)
    epick = rng.choice(edge.shape[0], n_edge_query, replace=True)
    ew = edge[epick] + rng.normal(0, 0.03, (n_edge_query, 3))
    e_local = quat_rot(q_inv, ew - t_true)
    map_refl = np.float32(10.0) + rng.integers(0, 30, surf.shape[0]).astype(np.float32) * np.float32(0.1)
    q_refl = np.float32(10.0) + rng.integers(0, 30, n_query).astype(np.float32) * np.float32(0.1) + np.float32(0.05)
    return dict(map_xyz=surf.astype(np.float32), map_refl=map_refl, edge_map_xyz=edge.astype(np.float32),
                q_xyz=q_local.astype(np.float32), q_refl=q_refl, eq_xyz=e_local.astype(np.float32),
                t_true=t_true, q_true=q_true)


# ------------------------------------------------------------------------------------------------
# Livox-Horizon-like scan (6 lines x 4000 time slots, FormatConvert's field layout)
# ------------------------------------------------------------------------------------------------
def make_livox_scan(seed=0, n_slots=4000, dup_frac=0.03, noise=0.02, origin=(0.0, 0.0, 1.8), yaw=0.0, inject_bad=True):
    """Returns (n,5) float32: x, y, z, intensity = line + 0.1 * t, curvature = 0.1 * reflectivity
    (L/src/FormatConvert.cpp:14-23).  6 close scan lines following a Lissajous pattern over an 80 x 20 deg field of
    view of the outdoor scene; a few duplicated time slots (first-writer-wins), out-of-range reflectivities,
    near-range points and a NaN are mixed in to exercise every filter of L/src/Preprocessing.cpp:243-268."""
    rng = np.random.default_rng(seed)
    sc = OutdoorScene()
    s = np.repeat(np.arange(n_slots), 6)
    line = np.tile(np.arange(6), n_slots)
    t = s / float(n_slots - 1)
    az = np.deg2rad(40.0 * np.sin(2 * np.pi * 1.0 * t + 0.4))
    el = np.deg2rad(-6.0 + 9.0 * np.sin(2 * np.pi * 9.3 * t) + 0.25 * line)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)      # sensor frame
    cy, sy = math.cos(yaw), math.sin(yaw)
    dw = np.stack([cy * d[:, 0] - sy * d[:, 1], sy * d[:, 0] + cy * d[:, 1], d[:, 2]], 1)  # world frame
    origin = np.asarray(origin, np.float64)
    rng_t = sc.raycast(origin, dw, t_max=400.0)
    ok = np.isfinite(rng_t) & (rng_t < 190.0)
    rng_t = np.where(ok, rng_t + rng.normal(0, noise, rng_t.shape), 1.0)
    pts = d * rng_t[:, None]
    refl = rng.integers(1, 255, pts.shape[0]).astype(np.float32)
    refl[rng.random(pts.shape[0]) < 0.01] = 0.0          # curvature 0.0 < 0.05 -> filtered, but still in lidar_cloud_cutted
    refl[rng.random(pts.shape[0]) < 0.01] = 255.0        # 25.5 > 25.45 -> filtered
    time_end = np.float32(t[-1] if t[-1] > 0 else 1.0)
    sfrac = (t.astype(np.float32) / time_end).astype(np.float32)
    intensity = (line + sfrac.astype(np.float64) * 0.1).astype(np.float32)       # uint8 + float * double -> float
    out = np.concatenate([pts, intensity[:, None], (0.1 * refl)[:, None]], 1).astype(np.float32)
    out = out[ok]
    if not inject_bad:
        return np.ascontiguousarray(out, np.float32)
    # duplicates of earlier time slots appended later in the stream: they must lose their grid cell
    nd = int(dup_frac * out.shape[0])
    dup = out[rng.choice(out.shape[0], nd, replace=False)].copy()
    dup[:, :3] += rng.normal(0, 0.05, (nd, 3)).astype(np.float32)
    pos = np.sort(rng.choice(out.shape[0], nd, replace=False))
    out = np.insert(out, pos, dup, axis=0)
    out[5, :3] *= 0.001                                   # near point (< 0.1 m)
    out[17, 0] = np.nan
    return np.ascontiguousarray(out, np.float32)
