"""CPU tests (no GPU): known-answer and property tests that pin the ORACLE itself.

The reference ships no tests or golden vectors and cannot be built offline, so the
oracle is pinned against semantics: brute force for kNN, numpy/LAPACK for the small linear algebra,
complex-step differentiation of the literal functor expressions for the Jacobians, closed forms for
the loss functions, and hand-checkable geometric cases.  (The reference's OWN code is pinned separately: tests/test_reference_cpu.py
compares the oracle with oracle/_ref — the reference's hot-path sources compiled as-is — bit for bit; this file covers the
third-party arithmetic those builds share with the oracle.)
"""
import numpy as np
import pytest

from lili_om_amd import synth


# ---------------------------------------------------------------------------------------------
# small linear algebra
# ---------------------------------------------------------------------------------------------
def test_eig3_matches_lapack(oracle):
    rng = np.random.default_rng(0)
    for k in range(500):
        B = rng.normal(size=(5, 3)) * rng.uniform(0.01, 10)
        if k % 3 == 0:
            B[:, 2] *= 1e-3
        if k % 5 == 0:
            B = B + rng.normal(size=3) * 100
        A = (B - B.mean(0)).T @ (B - B.mean(0))
        st, ev, V = oracle.eig3(A)
        assert st == 0
        ref = np.linalg.eigvalsh(A)
        assert np.abs(ev - ref).max() <= 1e-13 * np.abs(ref).max()
        assert np.all(np.diff(ev) >= 0)
        for i in range(3):
            assert np.linalg.norm(A @ V[i] - ev[i] * V[i]) <= 1e-13 * np.abs(ref).max()
            assert abs(np.linalg.norm(V[i]) - 1) < 1e-14


def test_eig3_degenerate_inputs(oracle):
    st, ev, V = oracle.eig3(np.zeros((3, 3)))
    assert st == 0 and not ev.any()
    st, ev, V = oracle.eig3(np.diag([3.0, 1.0, 2.0]))
    assert np.allclose(ev, [1, 2, 3])
    st, ev, V = oracle.eig3(np.full((3, 3), np.nan))
    assert st != 0 and np.isnan(ev).all()


def test_lstsq53_matches_lapack(oracle):
    rng = np.random.default_rng(1)
    for _ in range(500):
        A = rng.normal(size=(5, 3)) * rng.uniform(0.1, 10) + rng.normal(size=3) * rng.choice([0, 10, 300])
        b = -np.ones(5)
        x = oracle.lstsq53(A, b)
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()


def test_plane_fit_known_answer(oracle):
    """5 points on the plane z = 2: n.p + 1 = 0 -> n = (0,0,-1/2); unit normal (0,0,-1), d = 2."""
    pts = np.array([[0, 0, 2], [1, 0, 2], [0, 1, 2], [1, 1, 2], [0.3, 0.6, 2.0]])
    x = oracle.lstsq53(pts, -np.ones(5))
    assert np.allclose(x, [0, 0, -0.5], atol=1e-14)


def test_quaternion_rotation_is_eigen_formula(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        q = rng.normal(size=4)
        v = rng.normal(size=3)
        u = q[1:]
        uv = 2 * np.cross(u, v)
        ref = v + q[0] * uv + np.cross(u, uv)       # NOT normalised: SURVEY App. A5
        assert np.allclose(oracle.qrot(q, v), ref, rtol=1e-14, atol=1e-14)
    # unit quaternion = proper rotation
    q = np.array([np.cos(0.3), 0, 0, np.sin(0.3)])
    assert np.allclose(oracle.qrot(q, [1, 0, 0]), [np.cos(0.6), np.sin(0.6), 0])


# ---------------------------------------------------------------------------------------------
# loss functions (ceres::CauchyLoss / HuberLoss closed forms)
# ---------------------------------------------------------------------------------------------
def test_loss_known_answers(oracle):
    for s in (0.0, 1.0, 100.0):
        rho = oracle.loss(oracle.LOSS_CAUCHY, 1.0, s)
        assert np.allclose(rho, [np.log1p(s), 1 / (1 + s), -1 / (1 + s) ** 2], rtol=1e-15)
    assert np.allclose(oracle.loss(oracle.LOSS_HUBER, 0.1, 0.005), [0.005, 1, 0])
    s = 1.0
    assert np.allclose(oracle.loss(oracle.LOSS_HUBER, 0.1, s), [2 * 0.1 * 1 - 0.01, 0.1, -0.1 / 2])


# ---------------------------------------------------------------------------------------------
# exact kNN
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_kdtree_equals_brute_force(oracle, seed):
    rng = np.random.default_rng(seed)
    n = 4000
    pts = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    if seed == 1:   # planar + duplicates -> many exact ties
        pts[:, 2] = 0
        pts[n // 2:] = pts[: n // 2]
    if seed == 2:   # lattice -> equal distances everywhere
        pts = np.round(pts)
    q = rng.uniform(-11, 11, (300, 3)).astype(np.float32)
    if seed == 2:
        q = np.round(q * 2) / 2
    tree = oracle.KdTree(pts)
    i1, d1 = tree.knn5(q)
    i2, d2 = oracle.knn5_brute(pts, q)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert np.all(np.diff(d1, axis=1) >= 0)
    i3, d3 = tree.knn5(q, nthreads=4)
    assert np.array_equal(i1, i3)


def test_kdtree_agrees_with_scipy_ckdtree(oracle):
    """Independent implementation (scipy's cKDTree, f64 Euclidean distances) as a pin for the restated nearestKSearch:
    same neighbour SETS wherever the 5th/6th distances are not within f32 rounding of each other, and the f32
    L2_Simple distances within f32 rounding of scipy's squared f64 distances."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.uniform(-20, 20, (30000, 3)), rng.normal(0, 0.3, (5000, 3))]).astype(np.float32)
    q = np.concatenate([rng.uniform(-21, 21, (1500, 3)), rng.normal(0, 0.5, (500, 3))]).astype(np.float32)
    idx, d2 = oracle.KdTree(pts).knn5(q, nthreads=4)
    ds, js = cKDTree(pts.astype(np.float64)).query(q.astype(np.float64), k=6)
    clear = (ds[:, 5] - ds[:, 4]) > 1e-5 * np.maximum(ds[:, 5], 1.0)          # unambiguous 5-sets
    assert clear.mean() > 0.99
    assert np.array_equal(np.sort(idx[clear], 1), np.sort(js[clear, :5], 1))
    np.testing.assert_allclose(d2[clear], ds[clear, :5] ** 2, rtol=2e-6, atol=1e-9)
    order_clear = clear & np.all(np.diff(ds[:, :5], axis=1) > 1e-5 * np.maximum(ds[:, 1:5], 1.0), axis=1)
    assert np.array_equal(idx[order_clear], js[order_clear, :5])             # and the same ascending order


def test_kdtree_small_maps(oracle):
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tree = oracle.KdTree(pts)
    idx, d2 = tree.knn5(np.array([[0.1, 0.1, 0]], np.float32))
    assert list(idx[0][:3]) == [0, 1, 2] and idx[0][3] == -1 and np.isinf(d2[0][3])


# ---------------------------------------------------------------------------------------------
# factor Jacobians: dual numbers (oracle) vs complex-step differentiation of the functor text
# ---------------------------------------------------------------------------------------------
def _qrot_c(q, v):
    u = q[1:]
    uv = 2 * np.cross(u, v)
    return v + q[0] * uv + np.cross(u, uv)


def _edge_expr(x, cp, a, b, s):
    t, q = x[:3], x[3:]
    lp = _qrot_c(q, cp) + t
    nu = np.cross(lp - a, lp - b)
    de = a - b
    return s * np.sqrt((nu * nu).sum()) / np.sqrt((de * de).sum())


def _plane_expr(x, cp, n, d, score, qlb, tlb):
    t, q = x[:3], x[3:]
    n2 = (qlb * qlb).sum()
    qi = np.array([qlb[0], -qlb[1], -qlb[2], -qlb[3]]) / n2
    pw = _qrot_c(qi, cp - tlb)
    pw = _qrot_c(q, pw) + t
    return score * ((n * pw).sum() + d)


def _plane_incre_expr(x, cp, n, d):
    t, q = x[:3], x[3:]
    return (n * (_qrot_c(q, cp) + t)).sum() + d


def _cstep(f, x):
    g = np.zeros(7)
    for i in range(7):
        xc = x.astype(complex)
        xc[i] += 1e-30j
        g[i] = f(xc).imag / 1e-30
    return g, f(x.astype(complex)).real


def test_factor_jacobians_complex_step(oracle):
    rng = np.random.default_rng(3)
    P = oracle.params("rot")
    qlb, tlb = np.array(list(P.q_lb)), np.array(list(P.t_lb))
    for k in range(40):
        t = rng.normal(size=3) * 3
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if k % 2:
            q *= 1.0007           # Ceres keeps q only approximately unit inside an LM step
        x = np.concatenate([t, q])
        cp = (rng.normal(size=3) * 10).astype(np.float32)
        n = rng.normal(size=3).astype(np.float32)
        a = (rng.normal(size=3) * 5).astype(np.float32)
        b = (a + rng.normal(size=3).astype(np.float32) * 0.2).astype(np.float32)
        d, score, s = np.float32(rng.normal()), 3.7, 7.5
        J = oracle.eval_edge(t, q, cp, a, b, s)
        g, r = _cstep(lambda z: _edge_expr(z, cp.astype(float), a.astype(float), b.astype(float), s), x)
        assert np.allclose(J[:7], g, rtol=1e-11, atol=1e-11) and np.isclose(J[7], r, rtol=1e-13)
        J = oracle.eval_plane(t, q, cp, n, d, score, P)
        g, r = _cstep(lambda z: _plane_expr(z, cp.astype(float), n.astype(float), float(d), score, qlb, tlb), x)
        assert np.allclose(J[:7], g, rtol=1e-11, atol=1e-11) and np.isclose(J[7], r, rtol=1e-13)
        J = oracle.eval_plane(t, q, cp, n, d, 1.0, P, frontend=True)
        g, r = _cstep(lambda z: _plane_incre_expr(z, cp.astype(float), n.astype(float), float(d)), x)
        assert np.allclose(J[:7], g, rtol=1e-11, atol=1e-11) and np.isclose(J[7], r, rtol=1e-13)


def test_edge_factor_ignores_extrinsic(oracle):
    """LidarEdgeFactor stores qlb/tlb but never applies them — lp = q*cp + t."""
    t, q = np.array([0.5, -0.2, 0.1]), np.array([1.0, 0, 0, 0])
    cp = np.array([1, 2, 3], np.float32)
    a = np.array([1.5, 1.8, 4.0], np.float32)
    b = np.array([1.5, 1.8, 3.0], np.float32)
    J = oracle.eval_edge(t, q, cp, a, b, 2.0)
    lp = cp + t
    ref = 2.0 * np.linalg.norm(np.cross(lp - a, lp - b)) / np.linalg.norm(a - b)
    assert np.isclose(J[7], ref, rtol=1e-14)


# ---------------------------------------------------------------------------------------------
# association: hand-checkable scenes
# ---------------------------------------------------------------------------------------------
def test_surf_association_plane_known_answer(oracle):
    """Map = lattice on z=0, query 0.05 above it at (3,4): normal (0,0,+-1), pd = 0.05, weight per L:1662."""
    gx, gy = np.meshgrid(np.arange(0, 8, 0.4), np.arange(0, 8, 0.4))
    mp = np.stack([gx.ravel() + 0.013, gy.ravel() + 0.007, np.zeros(gx.size)], 1).astype(np.float32)
    mp[:, 2] += 10.0    # keep the plane away from the origin (n.p + 1 = 0 cannot represent planes through 0)
    tree = oracle.KdTree(mp)
    P = oracle.params("rot")
    q = np.array([[3.0, 4.0, 10.05]], np.float32)
    r = oracle.associate_surf(tree, None, q, None, [1, 0, 0, 0], [0, 0, 0], P)
    assert r["count"] == 1 and r["valid"][0] == 1
    pd = 0.05
    w = np.float32(1 - 0.9 * abs(np.float32(pd)) / np.sqrt(np.sqrt(np.float32(3 * 3 + 4 * 4 + 10.05 ** 2))))
    n = r["n"][0] / w
    assert np.allclose(np.abs(n), [0, 0, 1], atol=1e-6)
    assert np.isclose(abs(r["d"][0]), w * 10.0, rtol=1e-5)
    assert np.isclose(r["score"][0], 7.5 * w, rtol=1e-6)
    # beyond the 1 m gate -> no correspondence
    q2 = np.array([[3.0, 4.0, 11.2]], np.float32)
    assert oracle.associate_surf(tree, None, q2, None, [1, 0, 0, 0], [0, 0, 0], P)["count"] == 0
    # plane check: a bumpy map fails surf_dist_thres
    mp2 = mp.copy()
    ii, jj = np.meshgrid(np.arange(gx.shape[0]), np.arange(gx.shape[1]), indexing="ij")
    mp2[:, 2] += np.where(((ii + jj) % 2).ravel() == 0, 0.3, -0.3).astype(np.float32)   # checkerboard: not coplanar
    r3 = oracle.associate_surf(oracle.KdTree(mp2), None, q, None, [1, 0, 0, 0], [0, 0, 0], P)
    assert r3["count"] == 0


def test_edge_association_line_known_answer(oracle):
    z = np.arange(0, 4, 0.2)
    line = np.stack([np.full_like(z, 2.0), np.full_like(z, -1.0), z], 1).astype(np.float32)
    tree = oracle.KdTree(line)
    for variant, expect in (("livox", 1), ("rot", 1)):
        P = oracle.params(variant)
        q = np.array([[2.03, -1.02, 1.5]], np.float32)
        r = oracle.associate_edge(tree, q, [1, 0, 0, 0], [0, 0, 0], P)
        assert r["count"] == expect
        a, b = r["a"][0], r["b"][0]
        assert np.allclose((a - b) / np.linalg.norm(a - b), [0, 0, 1], atol=1e-6)   # canonical sign: +z
        assert np.isclose(np.linalg.norm(a - b), 0.2, rtol=1e-5)
        assert r["s"][0] == np.float32(P.lidar_const)
    # ROT only: more than 0.1 m from the line is rejected (R/src/BackendFusion.cpp:1443)
    q = np.array([[2.2, -1.0, 1.5]], np.float32)
    assert oracle.associate_edge(tree, q, [1, 0, 0, 0], [0, 0, 0], oracle.params("rot"))["count"] == 0
    assert oracle.associate_edge(tree, q, [1, 0, 0, 0], [0, 0, 0], oracle.params("livox"))["count"] == 1
    # an isotropic planar patch is not a line: centre + 4 lattice neighbours -> lambda2 == lambda1
    gx, gy = np.meshgrid(np.arange(-2, 3) * 0.3, np.arange(-2, 3) * 0.3)
    patch = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1).astype(np.float32)
    r = oracle.associate_edge(oracle.KdTree(patch), np.array([[0.01, 0.02, 0.0]], np.float32), [1, 0, 0, 0], [0, 0, 0], oracle.params("livox"))
    assert r["count"] == 0


def test_livox_reflectivity_gate(oracle):
    room = synth.make_room(seed=5, n_query=800, n_edge_query=10)
    tree = oracle.KdTree(room["map_xyz"])
    P = oracle.params("livox")
    Q2 = room["q_true"]; T2 = room["t_true"]
    r = oracle.associate_surf(tree, room["map_refl"], room["q_xyz"], room["q_refl"], Q2, T2, P)
    assert r["count"] > 100
    # identical reflectivity -> 1/0 = inf weights -> NaN normal -> silently dropped
    same = np.full_like(room["q_refl"], 7.0)
    r2 = oracle.associate_surf(tree, np.full_like(room["map_refl"], 7.0), room["q_xyz"], same, Q2, T2, P)
    assert r2["count"] == 0
    # large reflectivity differences -> rejected by reflect_thres (L:1628)
    r3 = oracle.associate_surf(tree, room["map_refl"], room["q_xyz"], room["q_refl"] + 100.0, Q2, T2, P)
    assert r3["count"] == 0


# ---------------------------------------------------------------------------------------------
# Gram / Gauss-Newton
# ---------------------------------------------------------------------------------------------
def test_gram_is_sum_of_robustified_outer_products(oracle):
    room = synth.make_room(seed=6, n_query=600, n_edge_query=60)
    P = oracle.params("rot")
    tree = oracle.KdTree(room["map_xyz"])
    t, q = room["t_true"], room["q_true"]
    rs = oracle.associate_surf(tree, None, room["q_xyz"], None, q, t, P)
    G, cost, cnt = oracle.linearize_surf(rs, t, q, P, 1.0)
    Gref = np.zeros((8, 8)); cref = 0.0
    for i in np.nonzero(rs["valid"])[0]:
        Jr = oracle.eval_plane(t, q, rs["cp"][i], rs["n"][i], rs["d"][i], rs["score"][i], P)
        s = Jr[7] ** 2
        w = np.sqrt(1 / (1 + s))            # Cauchy: rho'' < 0 always -> pure sqrt(rho') scaling
        Jr = Jr * w
        Gref += np.outer(Jr, Jr); cref += 0.5 * np.log1p(s)
    assert cnt == rs["count"]
    assert np.allclose(G, Gref, rtol=1e-12, atol=1e-12 * np.abs(Gref).max())
    assert np.isclose(cost, cref, rtol=1e-12)
    assert np.allclose(G, G.T) and np.linalg.eigvalsh(G).min() > -1e-9 * np.abs(G).max()


def test_gauss_newton_converges_on_room(oracle):
    room = synth.make_room(seed=7, n_query=3000, n_edge_query=300, noise=0.003)
    P = oracle.params("frontend")
    tree = oracle.KdTree(room["map_xyz"])
    rng = np.random.default_rng(synth.SEED_POSE)
    t, q = synth.perturbed_pose(room["t_true"], room["q_true"], rng, 0.15, 1.0)
    e0 = np.linalg.norm(t - room["t_true"])
    for _ in range(10):
        rs = oracle.associate_surf(tree, None, room["q_xyz"], None, q, t, P)
        G, _, _ = oracle.linearize_surf(rs, t, q, P)
        st, t, q, _ = oracle.gn_step(G, t, q)
        assert st == 0
    assert np.linalg.norm(t - room["t_true"]) < 0.2 * e0   # queries carry +-0.2 m synthetic offsets
    assert abs(np.linalg.norm(q) - 1) < 1e-9
    # singular normal matrix -> status 1, pose untouched
    st, t2, q2, _ = oracle.gn_step(np.zeros((8, 8)), t, q)
    assert st == 1 and np.array_equal(t2, t) and np.array_equal(q2, q)


def test_synth_is_deterministic():
    a = synth.make_room(seed=9, n_query=100, n_edge_query=10)
    b = synth.make_room(seed=9, n_query=100, n_edge_query=10)
    for k in a:
        assert np.array_equal(a[k], b[k])
    el = synth.hdl64_elevations_deg()
    ids = np.where(el >= -8.83, ((2 - el) * 3.0 + 0.5).astype(int), 32 + ((-8.83 - el) * 2.0 + 0.5).astype(int))
    assert np.array_equal(ids, np.arange(64))     # R/src/Preprocessing.cpp:333-336 maps the table back to 0..63


def test_fdlibm_atan_restatement_is_this_libm(oracle):
    """oracle/lo_math.h fd_atanf / fd_atan2f (glibc's float routines restated; the HIP extractor carries the same statements) against
    the libm of this image on 4 M pseudo-random inputs — bit for bit.  tools/check_fdlibm_atan.cpp is the exhaustive run (all 2^32
    atanf arguments, 3e8 atan2f pairs: 0 mismatches on glibc 2.35)."""
    import ctypes as C
    lib = oracle.lib()
    lib.lo_fd_atan_mismatches.restype = C.c_longlong
    lib.lo_fd_atan_mismatches.argtypes = [C.c_longlong, C.c_ulonglong]
    assert lib.lo_fd_atan_mismatches(2_000_000, 7) == 0
