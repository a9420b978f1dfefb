// Device-side f64 helpers of the gfx950 hot path.  Written for register residency: every loop is
// fully unrolled and every array index is a compile-time constant after unrolling, so nothing is
// placed in scratch memory.  Compiled with -ffp-contract=off: the reference's CPU build has no FMA
// contraction and the f32 decisions (distances, weights) must not depend on it.
#pragma once
#include <hip/hip_runtime.h>

namespace lili {

struct d3 { double x, y, z; };
struct dq { double w, x, y, z; };

__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross3(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// q * v as Eigen 3.3 evaluates it for a quaternion that is NOT assumed unit:
// v + w*(2 u x v) + u x (2 u x v)
__device__ __forceinline__ d3 qrot(dq q, d3 v) {
    d3 u{q.x, q.y, q.z};
    d3 uv = cross3(u, v);
    uv = uv + uv;
    return (v + q.w * uv) + cross3(u, uv);
}
__device__ __forceinline__ dq qmul(dq a, dq b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen inverse(): conjugate / squared norm
__device__ __forceinline__ dq qinv(dq q) {
    double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (n2 > 0) return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    return {0, 0, 0, 0};
}
// d(q*v)/dq for the expression above, as a 3x4 matrix [d/dw | d/dx d/dy d/dz]:
//   d/dw = 2 (u x v);   d/du = -2 w [v]x + 2 ((u.v) I + u v^T - 2 v u^T)
// rowK(g) returns g^T * D (1x4) for a 3-vector g, which is all the factors need.
__device__ __forceinline__ void qrot_jac_row(dq q, d3 v, d3 g, double out[4]) {
    d3 u{q.x, q.y, q.z};
    d3 uxv = cross3(u, v);
    out[0] = 2.0 * dot3(g, uxv);
    // g^T(-2w [v]x) = -2w (g x v)^T ... (g^T [v]x = (g x v)^T ... sign: [v]x a = v x a, g^T [v]x = -(v x g)^T... )
    // [v]x is skew: g^T [v]x = -([v]x g)^T = -(v x g)^T = (g x v)^T
    d3 gxv = cross3(g, v);
    double uv = dot3(u, v), gu = dot3(g, u), gv = dot3(g, v);
    // g^T ((u.v) I + u v^T - 2 v u^T) = (u.v) g^T + (g.u) v^T - 2 (g.v) u^T
    out[1] = -2.0 * q.w * gxv.x + 2.0 * (uv * g.x + gu * v.x - 2.0 * gv * u.x);
    out[2] = -2.0 * q.w * gxv.y + 2.0 * (uv * g.y + gu * v.y - 2.0 * gv * u.y);
    out[3] = -2.0 * q.w * gxv.z + 2.0 * (uv * g.z + gu * v.z - 2.0 * gv * u.z);
}

// ------------------------------------------------------------------------------------------------
// 3x3 symmetric eigen-decomposition by cyclic Jacobi rotations (f64).  Chosen over the tridiagonal
// QL iteration Eigen uses because it is branch-light and index-static on a GPU; both converge to
// the same eigen-pairs to ~1e-16 relative, and only threshold decisions / +-v enter the results.
// Output: ascending eigenvalues l0<=l1<=l2 and the unit eigenvector of l2 (vmax) and of l0 (vmin).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& arp, double& arq,
                                           double& vp0, double& vp1, double& vp2, double& vq0, double& vq1, double& vq2) {
    if (apq != 0.0) {
        double theta = (aqq - app) / (2.0 * apq);
        double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0) t = -t;
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        double tau = s / (1.0 + c);
        double h = t * apq;
        app -= h; aqq += h; apq = 0.0;
        double g = arp, hh = arq;
        arp = g - s * (hh + g * tau);
        arq = hh + s * (g - hh * tau);
        g = vp0; hh = vq0; vp0 = g - s * (hh + g * tau); vq0 = hh + s * (g - hh * tau);
        g = vp1; hh = vq1; vp1 = g - s * (hh + g * tau); vq1 = hh + s * (g - hh * tau);
        g = vp2; hh = vq2; vp2 = g - s * (hh + g * tau); vq2 = hh + s * (g - hh * tau);
    }
}
__device__ __forceinline__ void eig3_sym(double a00, double a01, double a02, double a11, double a12, double a22,
                                         double ev[3], d3& vmin, d3& vmax) {
    // scale like Eigen does (max |coeff|) so the zero tests below are scale free
    double scale = fmax(fmax(fabs(a00), fabs(a01)), fmax(fmax(fabs(a02), fabs(a11)), fmax(fabs(a12), fabs(a22))));
    if (!(scale > 0)) scale = 1.0;   // also catches NaN: results then propagate NaN
    a00 /= scale; a01 /= scale; a02 /= scale; a11 /= scale; a12 /= scale; a22 /= scale;
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;  // vIJ: component J of eigenvector I
#pragma unroll 1
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (!(off > 1e-300)) break;
        if (off <= 1e-18 * (fabs(a00) + fabs(a11) + fabs(a22))) break;
        jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v02, v10, v11, v12);   // (p,q)=(0,1), r=2
        jacobi_rot(a00, a22, a02, a01, a12, v00, v01, v02, v20, v21, v22);   // (0,2), r=1
        jacobi_rot(a11, a22, a12, a01, a02, v10, v11, v12, v20, v21, v22);   // (1,2), r=0
    }
    double l0 = a00, l1 = a11, l2 = a22;
    d3 e0{v00, v01, v02}, e1{v10, v11, v12}, e2{v20, v21, v22};
    if (l1 < l0) { double t = l0; l0 = l1; l1 = t; d3 tv = e0; e0 = e1; e1 = tv; }
    if (l2 < l1) { double t = l1; l1 = l2; l2 = t; d3 tv = e1; e1 = e2; e2 = tv; }
    if (l1 < l0) { double t = l0; l0 = l1; l1 = t; d3 tv = e0; e0 = e1; e1 = tv; }
    ev[0] = l0 * scale; ev[1] = l1 * scale; ev[2] = l2 * scale;
    vmin = e0; vmax = e2;
}
// canonical sign: the largest-magnitude component is positive (first one on ties)
__device__ __forceinline__ d3 canon_sign(d3 v) {
    double ax = fabs(v.x), ay = fabs(v.y), az = fabs(v.z);
    double lead = v.x; double m = ax;
    if (ay > m) { m = ay; lead = v.y; }
    if (az > m) { lead = v.z; }
    if (lead < 0) return {-v.x, -v.y, -v.z};
    return v;
}

// ------------------------------------------------------------------------------------------------
// min ||A x - b|| for a 5x3 A by Householder QR with column pivoting (same procedure as
// Eigen::ColPivHouseholderQR: pivot = largest remaining column norm, solve on the non-zero pivots).
// Columns are held as three 5-vectors so that pivoting is a conditional swap of whole columns.
// ------------------------------------------------------------------------------------------------
struct col5 { double v[5]; };
__device__ __forceinline__ void swap_col(col5& a, col5& b, bool doit) {
#pragma unroll
    for (int i = 0; i < 5; i++) { double t = a.v[i]; a.v[i] = doit ? b.v[i] : a.v[i]; b.v[i] = doit ? t : b.v[i]; }
}
template <int K> __device__ __forceinline__ double tail_sq(const col5& c) {
    double s = 0;
#pragma unroll
    for (int i = K; i < 5; i++) s += c.v[i] * c.v[i];
    return s;
}
// Householder on column c (rows K..4), applied to columns o1,o2 (if used) and b.  Returns the pivot (beta).
template <int K> __device__ __forceinline__ void house_apply(col5& c, col5* o1, col5* o2, col5& b) {
    double tailSq = tail_sq<K + 1>(c);
    double c0 = c.v[K];
    double beta, tau;
    if (tailSq <= 2.2250738585072014e-308) {
        tau = 0; beta = c0;
#pragma unroll
        for (int i = K + 1; i < 5; i++) c.v[i] = 0;
    } else {
        beta = sqrt(c0 * c0 + tailSq);
        if (c0 >= 0) beta = -beta;
        double inv_den = 1.0 / (c0 - beta);
#pragma unroll
        for (int i = K + 1; i < 5; i++) c.v[i] *= inv_den;
        tau = (beta - c0) / beta;
    }
    c.v[K] = beta;
    col5* cols[3] = {o1, o2, &b};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (cols[j] == nullptr) continue;
        col5& x = *cols[j];
        double s = x.v[K];
#pragma unroll
        for (int i = K + 1; i < 5; i++) s += c.v[i] * x.v[i];
        s *= tau;
        x.v[K] -= s;
#pragma unroll
        for (int i = K + 1; i < 5; i++) x.v[i] -= s * c.v[i];
    }
}
__device__ __forceinline__ void lstsq53(col5 c0, col5 c1, col5 c2, col5 b, double x[3]) {
    // pivoting compares SQUARED column norms (same order as the norms, no sqrt needed)
    const double eps = 2.220446049250313e-16;
    double n0 = tail_sq<0>(c0), n1 = tail_sq<0>(c1), n2 = tail_sq<0>(c2);
    double maxn = fmax(n0, fmax(n1, n2));
    double thr = maxn * (eps * eps) / 5.0;   // (max_norm * eps)^2 / rows
    int p0 = 0, p1 = 1, p2 = 2;   // original column index held in position 0,1,2
    int nz = 3;
    // ---- k = 0 : pivot among {0,1,2}
    {
        bool s1 = n1 > n0 && n1 >= n2;           // first maximum wins, like maxCoeff
        bool s2 = n2 > n0 && n2 > n1;
        if (maxn < thr * 5.0) nz = 0;
        swap_col(c0, c1, s1); if (s1) { int ti = p0; p0 = p1; p1 = ti; }
        swap_col(c0, c2, s2); if (s2) { int ti = p0; p0 = p2; p2 = ti; }
        house_apply<0>(c0, &c1, &c2, b);
    }
    // remaining column norms are recomputed exactly on the trailing rows (Eigen down-dates them and
    // recomputes when cancellation is detected; the pivot choice can differ only on near-ties)
    n1 = tail_sq<1>(c1); n2 = tail_sq<1>(c2);
    {
        bool s2 = n2 > n1;
        if (nz == 3 && fmax(n1, n2) < thr * 4.0) nz = 1;
        swap_col(c1, c2, s2); if (s2) { int ti = p1; p1 = p2; p2 = ti; }
        house_apply<1>(c1, &c2, nullptr, b);
    }
    n2 = tail_sq<2>(c2);
    {
        if (nz == 3 && n2 < thr * 3.0) nz = 2;
        house_apply<2>(c2, nullptr, nullptr, b);
    }
    // back substitution on the leading nz pivots; R = [[c0[0], c1[0], c2[0]], [0, c1[1], c2[1]], [0, 0, c2[2]]]
    double y0 = 0, y1 = 0, y2 = 0;
    if (nz >= 3) y2 = b.v[2] / c2.v[2];
    if (nz >= 2) y1 = (b.v[1] - c2.v[1] * y2) / c1.v[1];
    if (nz >= 1) y0 = (b.v[0] - c1.v[0] * y1 - c2.v[0] * y2) / c0.v[0];
    // x[perm[i]] = y[i]
    x[0] = (p0 == 0) ? y0 : ((p1 == 0) ? y1 : y2);
    x[1] = (p0 == 1) ? y0 : ((p1 == 1) ? y1 : y2);
    x[2] = (p0 == 2) ? y0 : ((p1 == 2) ? y1 : y2);
}

// ------------------------------------------------------------------------------------------------
// Fast path of the 5-point plane fit  min sum_k w_k^2 (p_k . n + 1)^2  (the least-squares problem the reference hands to
// colPivHouseholderQr): centred normal equations.  With c = weighted centroid, e_k = p_k - c, S = sum w_k^2 e_k e_k^T
// (3x3, entries O(neighbourhood size^2) — no 500 m offsets left in it) the cross term vanishes and
//     (S + W c c^T) n = -W c     =>     n = -W adj(S) c / (det S + W c^T adj(S) c)          (Sherman-Morrison / determinant lemma)
// ~110 f64 instructions, one division, no square root, no pivot bookkeeping — against ~450 for the pivoted Householder QR.
// The denominator equals det(A^T A); when it is small against the size of its terms (nearly collinear neighbours, or a
// plane through the origin) the caller falls back to lstsq53, which follows Eigen's rank-revealing procedure, so
// ill-conditioned and rank-deficient fits keep the reference's behaviour.  Elsewhere both paths agree to ~1e-9 relative
// (error <= eps / 1e-7), far below the f32 rounding of the stored record.
// ------------------------------------------------------------------------------------------------
template <bool WEIGHTED>
__device__ __forceinline__ bool plane_fit_centered(const double x[5], const double y[5], const double z[5], const double w2[5], double n[3], double tol = 1e-7) {
    double W, cx, cy, cz;
    if (WEIGHTED) {
        W = ((w2[0] + w2[1]) + (w2[2] + w2[3])) + w2[4];
        cx = fma(w2[4], x[4], fma(w2[3], x[3], fma(w2[2], x[2], fma(w2[1], x[1], w2[0] * x[0]))));
        cy = fma(w2[4], y[4], fma(w2[3], y[3], fma(w2[2], y[2], fma(w2[1], y[1], w2[0] * y[0]))));
        cz = fma(w2[4], z[4], fma(w2[3], z[3], fma(w2[2], z[2], fma(w2[1], z[1], w2[0] * z[0]))));
        const double rW = 1.0 / W;
        cx *= rW; cy *= rW; cz *= rW;
    } else {
        W = 5.0;
        cx = (((x[0] + x[1]) + (x[2] + x[3])) + x[4]) * 0.2;
        cy = (((y[0] + y[1]) + (y[2] + y[3])) + y[4]) * 0.2;
        cz = (((z[0] + z[1]) + (z[2] + z[3])) + z[4]) * 0.2;
    }
    double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double ex = x[k] - cx, ey = y[k] - cy, ez = z[k] - cz;
        const double wx = WEIGHTED ? w2[k] * ex : ex, wy = WEIGHTED ? w2[k] * ey : ey, wz = WEIGHTED ? w2[k] * ez : ez;
        sxx = fma(wx, ex, sxx); sxy = fma(wx, ey, sxy); sxz = fma(wx, ez, sxz);
        syy = fma(wy, ey, syy); syz = fma(wy, ez, syz); szz = fma(wz, ez, szz);
    }
    const double a00 = fma(syy, szz, -(syz * syz)), a01 = fma(sxz, syz, -(sxy * szz)), a02 = fma(sxy, syz, -(sxz * syy));
    const double a11 = fma(sxx, szz, -(sxz * sxz)), a12 = fma(sxy, sxz, -(sxx * syz)), a22 = fma(sxx, syy, -(sxy * sxy));
    const double det = fma(sxz, a02, fma(sxy, a01, sxx * a00));
    const double ux = fma(a02, cz, fma(a01, cy, a00 * cx));
    const double uy = fma(a12, cz, fma(a11, cy, a01 * cx));
    const double uz = fma(a22, cz, fma(a12, cy, a02 * cx));
    const double cu = fma(cz, uz, fma(cy, uy, cx * ux));
    const double c2 = fma(cz, cz, fma(cy, cy, cx * cx));
    const double tr = sxx + syy + szz;
    const double denom = fma(W, cu, det);
    const double scale = tr * tr * fma(W, c2, tr);
    if (!(denom > tol * scale)) return false;          // also catches NaN / inf inputs
    const double f = -W / denom;
    n[0] = f * ux; n[1] = f * uy; n[2] = f * uz;
    return true;
}

// ceres loss functions: rho[0..2] = rho(s), rho'(s), rho''(s)
__device__ __forceinline__ void loss_eval(int loss, double a, double s, double rho[3], bool want_rho0 = true) {
    if (loss == 1) {  // Cauchy
        double b = a * a, c = 1.0 / b;
        double sum = 1.0 + s * c, inv = 1.0 / sum;
        rho[0] = want_rho0 ? b * log(sum) : 0.0;   // the cost value is not needed by a Gauss-Newton step (f64 log: ~60 instructions)
        rho[1] = fmax(2.2250738585072014e-308, inv); rho[2] = -c * (inv * inv);
    } else if (loss == 2) {  // Huber
        double b = a * a;
        if (s > b) { double r = sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = fmax(2.2250738585072014e-308, a / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}
// Corrector of L/src/MarginalizationFactor.cpp:44-70 for a 1-residual block: scales J (7) and r in place.
__device__ __forceinline__ double robustify(int loss, double a, double J[7], double& r, bool want_cost = true) {
    double sq = r * r;
    double rho[3];
    loss_eval(loss, a, sq, rho, want_cost);
    double cost = 0.5 * rho[0];
    if (loss == 0) return cost;
    double sqrt_rho1 = sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else {
        double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
        double alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
    }
#pragma unroll
    for (int k = 0; k < 7; k++) J[k] = sqrt_rho1 * (J[k] - alpha_sq_norm * r * (r * J[k]));
    r *= residual_scaling;
    return cost;
}

}  // namespace lili
