"""Adversarial inputs for the exact 5-NN of the map index (reach-2 grid with on-demand shell, gate-bounded two-tier
selection, conservative pruning): the neighbour lists of every query the gate can accept must equal the brute-force
restatement — indices in (d², original index) order and f32 distances bit for bit."""
import numpy as np
import pytest

import lili_om_amd as L

pytestmark = pytest.mark.gpu

IDENT_Q, ZERO_T = [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0]


def _check(gpu_ctx, oracle, map_xyz, q_xyz, gate=1.0, min_inside=10, reach=None, cell_pct=None, focus=None, super_rows=None):
    P = L.make_params("frontend", kd_max_radius=gate)          # identity extrinsic: queries are map-frame points
    try:
        if reach is not None:
            gpu_ctx.set_option("grid_reach", reach)
        if cell_pct is not None:
            gpu_ctx.set_option("cell_pct", cell_pct)
        if super_rows is not None:
            gpu_ctx.set_option("super_rows", super_rows)
        m = L.ScanToMapMatcher(gpu_ctx, P)
        m.map_focus(*focus) if focus is not None else m.map_focus(None)
        gpu_ctx.set_debug(True)
        m.set_input_cloud(L.KIND_SURF, map_xyz)
        m.set_queries(0, L.KIND_SURF, q_xyz)
        m.find_corresponding_surf_features(0, IDENT_Q, ZERO_T)
        idx, d2 = m.neighbors(0, L.KIND_SURF, q_xyz.shape[0])
    finally:
        gpu_ctx.set_option("grid_reach", 2)
        gpu_ctx.set_option("cell_pct", 65)
        gpu_ctx.set_option("super_rows", 1)
        L.ScanToMapMatcher(gpu_ctx, P).map_focus(None)
    bi, bd = oracle.knn5_brute(map_xyz, q_xyz)
    inside = bd[:, 4] < gate
    assert inside.sum() >= min_inside
    assert np.array_equal(idx[inside], bi[inside])
    assert np.array_equal(d2[inside].view(np.uint32), bd[inside].view(np.uint32))
    assert np.all(~(d2[~inside][:, 4] < gate))                 # never a false accept
    return inside


@pytest.mark.parametrize("reach,pct", [(2, 65), (2, 50), (2, 100), (1, 65)])
def test_lattice_with_massive_exact_ties(gpu_ctx, oracle, reach, pct):
    """Integer lattice (spacing 0.25 m): queries on lattice points / edge midpoints / cell centres see 6-, 8-, 12-fold
    exact distance ties, many of them ACROSS grid cells — the tie fallback must reproduce the index order."""
    g = np.arange(-12, 13, dtype=np.float32) * np.float32(0.25)
    X, Y, Z = np.meshgrid(g, g, g[:9], indexing="ij")
    lattice = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    rng = np.random.default_rng(3)
    lattice = lattice[rng.permutation(lattice.shape[0])]        # original index order unrelated to position
    base = lattice[rng.integers(0, lattice.shape[0], 3000)]
    off = rng.choice(np.array([0.0, 0.125, 0.25], np.float32), (3000, 3))
    q = (base + off).astype(np.float32)
    _check(gpu_ctx, oracle, lattice, q, gate=1.0, min_inside=2500, reach=reach, cell_pct=pct)


def test_dense_cluster_and_sparse_halo(gpu_ctx, oracle):
    """20 000 points inside one 0.3 m blob (thousands per cell) plus a sparse halo: long runs, shell visits, and
    queries whose 5th neighbour sits just inside / just outside the gate."""
    rng = np.random.default_rng(4)
    blob = rng.normal(0, 0.08, (20000, 3)).astype(np.float32)
    halo = rng.uniform(-6, 6, (4000, 3)).astype(np.float32)
    m = np.concatenate([blob, halo]).astype(np.float32)
    q = np.concatenate([rng.normal(0, 0.5, (2000, 3)), rng.uniform(-7, 7, (4000, 3))]).astype(np.float32)
    inside = _check(gpu_ctx, oracle, m, q, gate=1.0, min_inside=1500)
    assert (~inside).sum() > 500


def test_points_on_cell_boundaries_and_outside_the_grid(gpu_ctx, oracle):
    """Map points and queries exactly on multiples of the cell edge (0.6565 m at 65 %), queries far outside the map's
    bounding box, and non-finite queries."""
    c = np.float32(1.01 * 0.65)
    k = np.arange(-6, 7, dtype=np.float32)
    X, Y, Z = np.meshgrid(k, k, k[4:9], indexing="ij")
    grid_pts = (np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1) * c).astype(np.float32)
    rng = np.random.default_rng(5)
    extra = rng.uniform(-4, 4, (3000, 3)).astype(np.float32)
    m = np.concatenate([grid_pts, extra]).astype(np.float32)
    q = np.concatenate([grid_pts[rng.integers(0, grid_pts.shape[0], 1500)],
                        (grid_pts[rng.integers(0, grid_pts.shape[0], 1500)] + np.float32(0.5) * c).astype(np.float32),
                        rng.uniform(-30, 30, (500, 3)).astype(np.float32),
                        np.array([[np.nan, 0, 0], [np.inf, 1, 1], [1e30, 0, 0]], np.float32)]).astype(np.float32)
    ok = np.all(np.isfinite(q), 1)
    _check(gpu_ctx, oracle, m, q[ok], gate=1.0, min_inside=2000)
    # non-finite queries: rejected, no crash
    P = L.make_params("frontend")
    mm = L.ScanToMapMatcher(gpu_ctx, P)
    mm.set_input_cloud(L.KIND_SURF, m)
    mm.set_queries(0, L.KIND_SURF, q[~ok])
    assert mm.find_corresponding_surf_features(0, IDENT_Q, ZERO_T) == 0


def test_large_gate_radius(gpu_ctx, oracle):
    """kd_max_radius of config_utbm.yaml (1.5, compared with a SQUARED distance): the index is rebuilt for the larger ball."""
    rng = np.random.default_rng(6)
    m = rng.uniform(-10, 10, (30000, 3)).astype(np.float32)
    q = rng.uniform(-11, 11, (5000, 3)).astype(np.float32)
    _check(gpu_ctx, oracle, m, q, gate=1.5, min_inside=1000)
    _check(gpu_ctx, oracle, m, q, gate=0.25, min_inside=5)


@pytest.mark.parametrize("seed", range(8))
def test_randomised_scenes_and_index_settings(gpu_ctx, oracle, seed):
    """Seeded differential sweep: random map sizes / extents / anisotropy / clustering, random gates and index settings."""
    rng = np.random.default_rng(1000 + seed)
    n_map = int(rng.integers(2000, 60000))
    ext = rng.uniform(2.0, 40.0, 3) * np.array([1.0, 1.0, rng.choice([1.0, 0.05])])     # sometimes nearly planar
    pts = rng.uniform(-1, 1, (n_map, 3)) * ext
    k = int(rng.integers(0, 4))
    if k:                                                                                # a few dense clusters
        centres = rng.uniform(-1, 1, (k, 3)) * ext
        extra = np.concatenate([c + rng.normal(0, rng.uniform(0.02, 0.5), (n_map // 4, 3)) for c in centres])
        pts = np.concatenate([pts, extra])
    pts = (pts + rng.uniform(-500, 500, 3)).astype(np.float32)                           # away from the origin: f32 grid effects
    q = (pts[rng.integers(0, pts.shape[0], 4000)] + rng.normal(0, rng.uniform(0.01, 1.0), (4000, 3))).astype(np.float32)
    gate = float(rng.choice([0.25, 0.64, 1.0, 1.5, 2.25]))
    reach = int(rng.choice([1, 2]))
    pct = int(rng.integers(50, 101))
    inside = _check(gpu_ctx, oracle, pts, q, gate=gate, min_inside=0, reach=reach, cell_pct=pct)
    assert inside.sum() + (~inside).sum() == 4000


@pytest.mark.parametrize("seed", range(6))
def test_randomised_focus_boxes(gpu_ctx, oracle, seed):
    """lili_map_focus is a hint: whatever the box — centred inside or outside the map, smaller than a cell or larger than the map, flat
    maps, reach 1 and 2 — queries inside it (one run of the super-row copy) and outside it (nine rows of the base index) get the
    brute-force neighbours; with the copy switched off as well."""
    rng = np.random.default_rng(7000 + seed)
    n_map = int(rng.integers(3000, 40000))
    ext = rng.uniform(3.0, 25.0, 3) * np.array([1.0, 1.0, rng.choice([1.0, 0.05])])
    pts = (rng.uniform(-1, 1, (n_map, 3)) * ext + rng.uniform(-200, 200, 3)).astype(np.float32)
    q = (pts[rng.integers(0, n_map, 3000)] + rng.normal(0, rng.uniform(0.05, 0.8), (3000, 3))).astype(np.float32)
    q[:50] += 60.0                                                                        # some queries far outside the map
    gate = float(rng.choice([0.36, 1.0, 2.25]))
    centre = pts[rng.integers(0, n_map)].astype(np.float64) + rng.normal(0, rng.choice([0.5, 30.0]), 3)
    radius = float(rng.choice([0.05, 2.0, 8.0, 500.0]))
    reach = int(rng.choice([1, 2]))
    _check(gpu_ctx, oracle, pts, q, gate=gate, min_inside=0, reach=reach, focus=(centre, radius))
    if seed % 3 == 0:
        _check(gpu_ctx, oracle, pts, q, gate=gate, min_inside=0, reach=reach, super_rows=0)


def test_capped_cell_count_stays_exact(gpu_ctx, oracle):
    """max_cells smaller than the bounding box needs: the cell edge grows (coarser cells, same exact result)."""
    rng = np.random.default_rng(77)
    pts = rng.uniform(-8, 8, (40000, 3)).astype(np.float32)
    q = (pts[rng.integers(0, 40000, 3000)] + rng.normal(0, 0.3, (3000, 3))).astype(np.float32)
    try:
        gpu_ctx.set_option("max_cells", 2000)
        m = L.ScanToMapMatcher(gpu_ctx, L.make_params("frontend"))
        m.set_input_cloud(L.KIND_SURF, pts)
        n, n_cells, cell = m.map_info(L.KIND_SURF)
        assert n_cells <= 2000 and cell > 1.2          # 0.66 m would need ~15 000 cells
        _check(gpu_ctx, oracle, pts, q, gate=1.0, min_inside=100)
    finally:
        gpu_ctx.set_option("max_cells", 1 << 27)
