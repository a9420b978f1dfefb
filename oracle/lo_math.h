// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped or measured as the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// Small dependency-free f64 linear algebra that restates, on the CPU, the third-party arithmetic
// the reference calls on its hot path (Eigen 3.3 / Ceres 2.0 — NOT present under /root/reference,
// unpinned: see SURVEY.md §8c, App. B).  PARITY: the extractor and factor-functor restatements built
// on this file are pinned bit for bit against the reference's own sources compiled as-is
// (oracle/refshim/README.md, tests/test_reference_cpu.py); the third-party arithmetic restated HERE
// (Eigen / Ceres / PCL rules) and the matcher's association remain checked against semantics only
// (brute force, LAPACK, complex-step differences, hand-checkable cases).
//
// Compile with -ffp-contract=off and without -ffast-math (reference build: L/CMakeLists.txt:5,62).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace lo {

struct V3 { double x, y, z; };
struct Q4 { double w, x, y, z; };  // Eigen::Quaterniond coefficient meaning; storage order here is w,x,y,z

static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// Eigen 3.3 QuaternionBase::_transformVector: uv = 2 (u x v); v + w*uv + u x uv.  The quaternion
// is NOT normalised inside — the reference relies on that with its non-unit q_lb.
static inline V3 qrot(Q4 q, V3 v) {
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return (v + q.w * uv) + cross(u, uv);
}
// Eigen quat_product (generic path), a*b
static inline Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen QuaternionBase::inverse(): conjugate / squaredNorm (zero quaternion -> zeros)
static inline Q4 qinv(Q4 q) {
    double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (n2 > 0) return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    return {0, 0, 0, 0};
}
// Eigen 3.3 QuaternionBase::slerp(t, other) called on *this = a
static inline Q4 qslerp(Q4 a, double t, Q4 b) {
    const double one = 1.0 - 2.220446049250313e-16;
    double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    double absD = std::fabs(d);
    double s0, s1;
    if (absD >= one) { s0 = 1.0 - t; s1 = t; }
    else {
        double theta = std::acos(absD);
        double sinTheta = std::sin(theta);
        s0 = std::sin((1.0 - t) * theta) / sinTheta;
        s1 = std::sin(t * theta) / sinTheta;
    }
    if (d < 0) s1 = -s1;
    return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// ---------------------------------------------------------------------------------------------
// 3x3 symmetric eigen-decomposition, following the structure of Eigen 3.3
// SelfAdjointEigenSolver<Matrix3d>::compute(): scale by max |coeff|, Householder tridiagonalise,
// implicit symmetric QR (Wilkinson shift) iterations, sort ascending.  
// evals ascending; evecs[k] = eigenvector of evals[k] (unit norm, sign arbitrary).
// Returns false when the iteration does not converge or the input holds non-finite values
// (Eigen then reports NoConvergence and leaves values the reference compares anyway; callers here
// treat it as "all comparisons false", which is what NaN input does in the reference).
// ---------------------------------------------------------------------------------------------
static inline bool eig3_sym(const double A_in[3][3], double evals[3], double evecs[3][3]) {
    double scale = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) {  // lower triangle, like Eigen
        double a = std::fabs(A_in[i][j]);
        if (!(a == a)) { evals[0] = evals[1] = evals[2] = NAN; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) evecs[r][c] = NAN; return false; }
        if (a > scale) scale = a;
    }
    if (scale == 0) scale = 1;
    double a00 = A_in[0][0] / scale, a10 = A_in[1][0] / scale, a20 = A_in[2][0] / scale;
    double a11 = A_in[1][1] / scale, a21 = A_in[2][1] / scale, a22 = A_in[2][2] / scale;
    // Tridiagonalisation (Eigen tridiagonalization_inplace_selector<MatrixType,3,false>)
    double diag[3], sub[2];
    double Q[3][3];  // columns = eigenvector basis
    const double tol = 2.2250738585072014e-308;
    diag[0] = a00;
    double v1norm2 = a20 * a20;
    if (v1norm2 <= tol) {
        diag[1] = a11; diag[2] = a22; sub[0] = a10; sub[1] = a21;
        Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0; Q[1][0] = 0; Q[1][1] = 1; Q[1][2] = 0; Q[2][0] = 0; Q[2][1] = 0; Q[2][2] = 1;
    } else {
        double beta = std::sqrt(a10 * a10 + v1norm2);
        double invBeta = 1.0 / beta;
        double m01 = a10 * invBeta, m02 = a20 * invBeta;
        double q = 2.0 * m01 * a21 + m02 * (a22 - a11);
        diag[1] = a11 + m02 * q; diag[2] = a22 - m02 * q;
        sub[0] = beta; sub[1] = a21 - m01 * q;
        Q[0][0] = 1; Q[0][1] = 0; Q[0][2] = 0;
        Q[1][0] = 0; Q[1][1] = m01; Q[1][2] = m02;
        Q[2][0] = 0; Q[2][1] = m02; Q[2][2] = -m01;
    }
    // Implicit symmetric QR (Eigen computeFromTridiagonal_impl + tridiagonal_qr_step)
    int end = 2, start = 0, iter = 0;
    const int maxit = 30 * 3;
    const double considerAsZero = 2.2250738585072014e-308;
    const double precision = 2.0 * 2.220446049250313e-16;
    while (end > 0) {
        for (int i = start; i < end; ++i) {
            // Eigen 3.3.7: isMuchSmallerThan(|sub[i]|, |diag[i]|+|diag[i+1]|, precision) || |sub[i]| <= considerAsZero
            double s = std::fabs(sub[i]);
            if (s <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision || s <= considerAsZero) sub[i] = 0;
        }
        while (end > 0 && sub[end - 1] == 0) end--;
        if (end <= 0) break;
        if (++iter > maxit) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0) start--;
        // one QR step on [start, end]
        double td = (diag[end - 1] - diag[end]) * 0.5;
        double e = sub[end - 1];
        double mu = diag[end];
        if (td == 0) mu -= std::fabs(e);
        else {
            double e2 = e * e;
            double h = std::hypot(td, e);
            if (e2 == 0) mu -= (e / (td + (td > 0 ? 1 : -1))) * (e / h);
            else mu -= e2 / (td + (td > 0 ? h : -h));
        }
        double x = diag[start] - mu;
        double z = sub[start];
        for (int k = start; k < end; ++k) {
            // JacobiRotation::makeGivens(x, z)
            double c, s;
            if (z == 0) { c = x < 0 ? -1 : 1; s = 0; }
            else if (x == 0) { c = 0; s = z < 0 ? 1 : -1; }
            else if (std::fabs(x) > std::fabs(z)) { double t = z / x; double u = std::sqrt(1 + t * t); if (x < 0) u = -u; c = 1 / u; s = -t * c; }
            else { double t = x / z; double u = std::sqrt(1 + t * t); if (z < 0) u = -u; s = -1 / u; c = -t * s; }
            double sdk = s * diag[k] + c * sub[k];
            double dkp1 = s * sub[k] + c * diag[k + 1];
            diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
            diag[k + 1] = s * sdk + c * dkp1;
            sub[k] = c * sdk - s * dkp1;
            if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
            x = sub[k];
            if (k < end - 1) { z = -s * sub[k + 1]; sub[k + 1] = c * sub[k + 1]; }
            // Q = Q * rot  (applyOnTheRight(k, k+1, rot)): col_k' = c col_k - s col_k+1 ; col_k+1' = s col_k + c col_k+1
            for (int r = 0; r < 3; r++) {
                double qk = Q[r][k], qk1 = Q[r][k + 1];
                Q[r][k] = c * qk - s * qk1;
                Q[r][k + 1] = s * qk + c * qk1;
            }
        }
    }
    bool ok = iter <= maxit;
    // sort ascending (selection sort, swapping columns) like Eigen
    for (int i = 0; i < 2; i++) {
        int k = i;
        for (int j = i + 1; j < 3; j++) if (diag[j] < diag[k]) k = j;
        if (k != i) {
            double t = diag[i]; diag[i] = diag[k]; diag[k] = t;
            for (int r = 0; r < 3; r++) { double tt = Q[r][i]; Q[r][i] = Q[r][k]; Q[r][k] = tt; }
        }
    }
    for (int k = 0; k < 3; k++) { evals[k] = diag[k] * scale; for (int r = 0; r < 3; r++) evecs[k][r] = Q[r][k]; }
    return ok;
}

// ---------------------------------------------------------------------------------------------
// x = argmin ||A x - b||, A 5x3, via column-pivoted Householder QR — the procedure of
// Eigen 3.3 ColPivHouseholderQR::compute + _solve_impl (App. B4): pivot on the largest remaining
// (down-dated) column norm, Householder reflectors, solve on the leading nonzero pivots.
// ---------------------------------------------------------------------------------------------
static inline void lstsq_5x3_colpiv(const double A_in[5][3], const double b_in[5], double x[3]) {
    const int R = 5, C = 3;
    double A[5][3], b[5];
    for (int i = 0; i < R; i++) { b[i] = b_in[i]; for (int j = 0; j < C; j++) A[i][j] = A_in[i][j]; }
    int perm[3] = {0, 1, 2};
    double colNormsUpdated[3], colNormsDirect[3];
    double maxnorm = 0;
    for (int j = 0; j < C; j++) {
        double s = 0; for (int i = 0; i < R; i++) s += A[i][j] * A[i][j];
        colNormsDirect[j] = colNormsUpdated[j] = std::sqrt(s);
        if (colNormsUpdated[j] > maxnorm) maxnorm = colNormsUpdated[j];
    }
    const double eps = 2.220446049250313e-16;
    double threshold_helper = (maxnorm * eps) * (maxnorm * eps) / double(R);
    double norm_downdate_threshold = std::sqrt(eps);
    int nonzero_pivots = C;
    double hcoef[3] = {0, 0, 0};
    for (int k = 0; k < C; k++) {
        int big = k; double bigv = colNormsUpdated[k];
        for (int j = k + 1; j < C; j++) if (colNormsUpdated[j] > bigv) { bigv = colNormsUpdated[j]; big = j; }
        double biggest_sq = bigv * bigv;
        if (nonzero_pivots == C && biggest_sq < threshold_helper * double(R - k)) nonzero_pivots = k;
        if (big != k) {
            for (int i = 0; i < R; i++) { double t = A[i][k]; A[i][k] = A[i][big]; A[i][big] = t; }
            double t = colNormsUpdated[k]; colNormsUpdated[k] = colNormsUpdated[big]; colNormsUpdated[big] = t;
            t = colNormsDirect[k]; colNormsDirect[k] = colNormsDirect[big]; colNormsDirect[big] = t;
            int ti = perm[k]; perm[k] = perm[big]; perm[big] = ti;
        }
        // makeHouseholderInPlace on A[k..R-1][k]
        double tailSq = 0; for (int i = k + 1; i < R; i++) tailSq += A[i][k] * A[i][k];
        double c0 = A[k][k], beta, tau;
        if (tailSq <= 2.2250738585072014e-308) { tau = 0; beta = c0; for (int i = k + 1; i < R; i++) A[i][k] = 0; }
        else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < R; i++) A[i][k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta; hcoef[k] = tau;
        // apply H = I - tau v v^T (v = [1; essential]) to the trailing columns and to b
        for (int j = k + 1; j < C; j++) {
            double s = A[k][j]; for (int i = k + 1; i < R; i++) s += A[i][k] * A[i][j];
            s *= tau;
            A[k][j] -= s; for (int i = k + 1; i < R; i++) A[i][j] -= s * A[i][k];
        }
        {
            double s = b[k]; for (int i = k + 1; i < R; i++) s += A[i][k] * b[i];
            s *= tau;
            b[k] -= s; for (int i = k + 1; i < R; i++) b[i] -= s * A[i][k];
        }
        // column-norm down-dating (LAPACK-style, as in Eigen)
        for (int j = k + 1; j < C; j++) {
            if (colNormsUpdated[j] != 0) {
                double temp = std::fabs(A[k][j]) / colNormsUpdated[j];
                temp = (1 + temp) * (1 - temp);
                temp = temp < 0 ? 0 : temp;
                double r = colNormsUpdated[j] / colNormsDirect[j];
                double temp2 = temp * r * r;
                if (temp2 <= norm_downdate_threshold) {
                    double s = 0; for (int i = k + 1; i < R; i++) s += A[i][j] * A[i][j];
                    colNormsDirect[j] = std::sqrt(s);
                    colNormsUpdated[j] = colNormsDirect[j];
                } else colNormsUpdated[j] *= std::sqrt(temp);
            }
        }
    }
    (void)hcoef;
    double y[3] = {0, 0, 0};
    for (int i = nonzero_pivots - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < nonzero_pivots; j++) s -= A[i][j] * y[j];
        y[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0;
    for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = y[i];
}

// ---------------------------------------------------------------------------------------------
// Forward-mode dual numbers with 7 partials: what ceres::AutoDiffCostFunction<F,1,3,4> feeds the
// reference's templated functors (ceres/jet.h arithmetic).
// ---------------------------------------------------------------------------------------------
struct Jet7 {
    double a; double v[7];
    Jet7() : a(0) { for (int i = 0; i < 7; i++) v[i] = 0; }
    Jet7(double a_) : a(a_) { for (int i = 0; i < 7; i++) v[i] = 0; }
    Jet7(double a_, int k) : a(a_) { for (int i = 0; i < 7; i++) v[i] = 0; v[k] = 1; }
};
static inline Jet7 operator+(const Jet7& f, const Jet7& g) { Jet7 r; r.a = f.a + g.a; for (int i = 0; i < 7; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
static inline Jet7 operator-(const Jet7& f, const Jet7& g) { Jet7 r; r.a = f.a - g.a; for (int i = 0; i < 7; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
static inline Jet7 operator*(const Jet7& f, const Jet7& g) { Jet7 r; r.a = f.a * g.a; for (int i = 0; i < 7; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
static inline Jet7 operator/(const Jet7& f, const Jet7& g) {
    Jet7 r; double gi = 1.0 / g.a; double fg = f.a * gi; r.a = fg;
    for (int i = 0; i < 7; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi;
    return r;
}
static inline Jet7 jsqrt(const Jet7& f) { Jet7 r; r.a = std::sqrt(f.a); double t = 1.0 / (2.0 * r.a); for (int i = 0; i < 7; i++) r.v[i] = f.v[i] * t; return r; }
struct JV3 { Jet7 x, y, z; };
static inline JV3 operator+(const JV3& a, const JV3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline JV3 operator-(const JV3& a, const JV3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline JV3 jcross(const JV3& a, const JV3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline Jet7 jdot(const JV3& a, const JV3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Jet7 jnorm(const JV3& a) { return jsqrt(jdot(a, a)); }
struct JQ4 { Jet7 w, x, y, z; };
static inline JV3 jqrot(const JQ4& q, const JV3& v) {
    JV3 u{q.x, q.y, q.z};
    JV3 uv = jcross(u, v);
    uv = uv + uv;
    JV3 wuv{q.w * uv.x, q.w * uv.y, q.w * uv.z};
    return (v + wuv) + jcross(u, uv);
}

// Cholesky solve of an n x n SPD system (n <= 8), in place; returns false if not positive definite.
static inline bool chol_solve(int n, double* H /*row-major n*n*/, double* rhs) {
    for (int j = 0; j < n; j++) {
        double d = H[j * n + j];
        for (int k = 0; k < j; k++) d -= H[j * n + k] * H[j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d); H[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = H[i * n + j];
            for (int k = 0; k < j; k++) s -= H[i * n + k] * H[j * n + k];
            H[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= H[i * n + k] * rhs[k]; rhs[i] = s / H[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = rhs[i]; for (int k = i + 1; k < n; k++) s -= H[k * n + i] * rhs[k]; rhs[i] = s / H[i * n + i]; }
    return true;
}


// ---- glibc's float atan / atan2 (third-party, NOT under /root/reference) ------------------------------------------------
// R/src/Preprocessing.cpp:285-288,315,349 call atan / atan2 on float arguments = libm's atanf / atan2f.  The reference pins no
// libm; every glibc up to 2.40 (Ubuntu 18.04 / 20.04 / 22.04 of the ROS releases the README names, and this image: 2.35) ships
// the fdlibm float routines sysdeps/ieee754/flt-32/s_atanf.c and e_atan2f.c (Sun Microsystems' algorithm: argument reduction to
// |x| < 7/16 by one float division, an 11-term odd/even polynomial, hi/lo table constants).  They are float-only arithmetic and
// therefore reproducible instruction by instruction without FMA contraction; restated here from the published algorithm and
// PINNED against this image's libm on all 2^32 atanf arguments and 3e8 atan2f pairs (tools/check_fdlibm_atan.cpp: 0 mismatches;
// tests/test_oracle_cpu.py samples it).  The HIP extractor carries the same statements (lili_extract_rot.hip), so its ring /
// relTime decisions are the reference build's bit for bit.  glibc >= 2.41 switched atanf to a correctly rounded routine.
static inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float bitsf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float fd_atanf(float x) {
    static const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                               6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const float one = 1.0f, huge = 1.0e30f;
    float w, s1, s2, z;
    int32_t ix, hx, id;
    hx = (int32_t)fbits(x);
    ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {          /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        else return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {           /* |x| < 0.4375 */
        if (ix < 0x31000000) {       /* |x| < 2^-29 */
            if (huge + x > one) return x;
        }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {       /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}
static inline float fd_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    int32_t k, m, hx, hy, ix, iy;
    hx = (int32_t)fbits(x); ix = hx & 0x7fffffff;
    hy = (int32_t)fbits(y); iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return fd_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            case 3: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                case 3: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                case 3: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fd_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return bitsf(fbits(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace lo
