// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md).
//
// Stand-in for the handful of Eigen 3.3 types the reference's hot-path sources use, so that those
// sources can be compiled UNMODIFIED from /root/reference into oracle/_ref/ (Eigen itself is not in this
// image).  Fixed-size dense matrices and quaternions only, written from Eigen 3.3's documented
// semantics: element-wise arithmetic in index order, 3-vector reductions as
// (x*x + y*y) + z*z, QuaternionBase::_transformVector as v + w*(2 u x v) + u x (2 u x v), generic
// quaternion product, inverse = conjugate / squaredNorm, slerp with the 1-eps threshold.  The iterative
// 3x3 SelfAdjointEigenSolver delegates to oracle/lo_math.h (the oracle's restatement of the same
// third-party algorithm) — the shim adds no second opinion on third-party arithmetic; what it buys
// is that the REFERENCE'S OWN statements run as written.
#pragma once
#include <cmath>
#include <algorithm>
#include <cstddef>
#include <vector>
#include "../../../lo_math.h"

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_ALIGN16 __attribute__((aligned(16)))

namespace Eigen {

template <class Derived> struct MatrixBase {
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    Derived& derived() { return *static_cast<Derived*>(this); }
};

enum { Dynamic = -1, ColMajor = 0, RowMajor = 1 };
template <class T, int R, int C, int O = 0> struct Matrix;
template <class T, int R, int C> struct Matrix<T, R, C, 0> : MatrixBase<Matrix<T, R, C, 0>> {
    typedef T Scalar;
    enum { Rows = R, Cols = C, Size = R * C };
    T d[R * C];   // column-major like Eigen's default
    Matrix() : d() {}
    Matrix(const T& a, const T& b, const T& c) : d{a, b, c} { static_assert(R * C == 3, "3-vector ctor"); }
    template <class D> Matrix(const MatrixBase<D>& o) { *this = o.derived(); }
    T& operator()(int i) { return d[i]; }
    const T& operator()(int i) const { return d[i]; }
    T& operator[](int i) { return d[i]; }
    const T& operator[](int i) const { return d[i]; }
    T& operator()(int r, int c) { return d[c * R + r]; }
    const T& operator()(int r, int c) const { return d[c * R + r]; }
    T& x() { return d[0]; } const T& x() const { return d[0]; }
    T& y() { return d[1]; } const T& y() const { return d[1]; }
    T& z() { return d[2]; } const T& z() const { return d[2]; }
    static Matrix Zero() { Matrix m; for (int i = 0; i < R * C; i++) m.d[i] = T(0); return m; }
    static Matrix Identity() { Matrix m = Zero(); for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = T(1); return m; }
    static Matrix Ones() { Matrix m; for (int i = 0; i < R * C; i++) m.d[i] = T(1); return m; }
    T sum() const { T s = d[0]; for (int i = 1; i < R * C; i++) s = s + d[i]; return s; }   // association order unspecified in Eigen; only used for values the reference never reads
    void normalize() { T n = norm(); for (int i = 0; i < R * C; i++) d[i] = d[i] / n; }        // *this /= norm()
    struct ColPivQR53;   // colPivHouseholderQr() of the 5x3 plane-fit matrix
    ColPivQR53 colPivHouseholderQr() const;
    Matrix operator+(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] + o.d[i]; return r; }
    Matrix operator-(const Matrix& o) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] - o.d[i]; return r; }
    Matrix operator-() const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = -d[i]; return r; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; i++) d[i] = d[i] + o.d[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; i++) d[i] = d[i] - o.d[i]; return *this; }
    Matrix operator*(const T& s) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] * s; return r; }
    Matrix operator/(const T& s) const { Matrix r; for (int i = 0; i < R * C; i++) r.d[i] = d[i] / s; return r; }
    Matrix& operator*=(const T& s) { for (int i = 0; i < R * C; i++) d[i] = d[i] * s; return *this; }
    Matrix& operator/=(const T& s) { for (int i = 0; i < R * C; i++) d[i] = d[i] / s; return *this; }   // scalar_quotient_op: true division
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> r; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) r(j, i) = (*this)(i, j); return r; }
    Matrix<T, R, 1> col(int j) const { Matrix<T, R, 1> r; for (int i = 0; i < R; i++) r.d[i] = (*this)(i, j); return r; }
    // 3-vector helpers (Eigen: redux over x, y, z in order; cross as the textbook determinant)
    T dot(const Matrix& o) const { static_assert(R * C == 3, ""); return d[0] * o.d[0] + d[1] * o.d[1] + d[2] * o.d[2]; }
    T squaredNorm() const { static_assert(R * C == 3, ""); return d[0] * d[0] + d[1] * d[1] + d[2] * d[2]; }
    T norm() const { using std::sqrt; return sqrt(squaredNorm()); }
    Matrix cross(const Matrix& o) const {
        static_assert(R * C == 3, "");
        return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]);
    }
};
template <class T, int R, int C> Matrix<T, R, C> operator*(const T& s, const Matrix<T, R, C>& m) { return m * s; }
template <class T, int R, int K, int C> Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b) {
    Matrix<T, R, C> r;
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) {
        T s = a(i, 0) * b(0, j);
        for (int k = 1; k < K; k++) s = s + a(i, k) * b(k, j);
        r(i, j) = s;
    }
    return r;
}
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;

template <class T> struct AngleAxis;   // named by utils/math_tools.h templates that are never instantiated here

template <class Derived> struct QuaternionBase {};

template <class T> struct Quaternion : QuaternionBase<Quaternion<T>> {
    typedef T Scalar;
    T qw, qx, qy, qz;
    Quaternion() : qw(), qx(), qy(), qz() {}
    Quaternion(const T& w_, const T& x_, const T& y_, const T& z_) : qw(w_), qx(x_), qy(y_), qz(z_) {}   // (w, x, y, z) like Eigen
    static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
    T& w() { return qw; } const T& w() const { return qw; }
    T& x() { return qx; } const T& x() const { return qx; }
    T& y() { return qy; } const T& y() const { return qy; }
    T& z() { return qz; } const T& z() const { return qz; }
    Matrix<T, 3, 1> vec() const { return Matrix<T, 3, 1>(qx, qy, qz); }
    // internal::quat_product, generic path
    Quaternion operator*(const Quaternion& b) const {
        const Quaternion& a = *this;
        return Quaternion(a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz,
                          a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy,
                          a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz,
                          a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx);
    }
    Quaternion& operator*=(const Quaternion& b) { *this = *this * b; return *this; }
    // QuaternionBase::_transformVector — no normalisation inside
    Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1>& v) const {
        Matrix<T, 3, 1> u = vec();
        Matrix<T, 3, 1> uv = u.cross(v);
        uv += uv;
        return v + uv * qw + u.cross(uv);
    }
    T squaredNorm() const { return qx * qx + qy * qy + qz * qz + qw * qw; }   // coeffs() order x, y, z, w
    Quaternion conjugate() const { return Quaternion(qw, -qx, -qy, -qz); }
    Quaternion inverse() const {
        T n2 = squaredNorm();
        if (n2 > T(0)) return Quaternion(qw / n2, -qx / n2, -qy / n2, -qz / n2);
        return Quaternion(T(0), T(0), T(0), T(0));
    }
    T dot(const Quaternion& o) const { return qx * o.qx + qy * o.qy + qz * o.qz + qw * o.qw; }
    Quaternion slerp(const T& t, const Quaternion& other) const {
        using std::acos; using std::sin; using std::abs;
        const T one = T(1) - T(2.220446049250313e-16);
        T d = this->dot(other);
        T absD = abs(d);
        T scale0, scale1;
        if (absD >= one) { scale0 = T(1) - t; scale1 = t; }
        else {
            T theta = acos(absD);
            T sinTheta = sin(theta);
            scale0 = sin((T(1) - t) * theta) / sinTheta;
            scale1 = sin((t * theta)) / sinTheta;
        }
        if (d < T(0)) scale1 = -scale1;
        return Quaternion(scale0 * qw + scale1 * other.qw, scale0 * qx + scale1 * other.qx,
                          scale0 * qy + scale1 * other.qy, scale0 * qz + scale1 * other.qz);
    }
};
typedef Quaternion<double> Quaterniond;

// Matrix<double,5,3>::colPivHouseholderQr().solve(b) — delegated to the oracle's restatement of Eigen 3.3's
// ColPivHouseholderQR (lo::lstsq_5x3_colpiv); any other shape is a compile error.
template <class T, int R, int C> struct Matrix<T, R, C>::ColPivQR53 {
    static_assert(R == 5 && C == 3, "only the 5x3 plane fit is stood in for");
    double A[5][3];
    Matrix<double, 3, 1> solve(const Matrix<double, 5, 1>& b) const {
        double bb[5], x[3];
        for (int i = 0; i < 5; i++) bb[i] = b.d[i];
        lo::lstsq_5x3_colpiv(A, bb, x);
        return Matrix<double, 3, 1>(x[0], x[1], x[2]);
    }
};
template <class T, int R, int C> typename Matrix<T, R, C>::ColPivQR53 Matrix<T, R, C>::colPivHouseholderQr() const {
    ColPivQR53 q;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 3; j++) q.A[i][j] = (*this)(i, j);
    return q;
}

// SelfAdjointEigenSolver<Matrix3d>(A): iterative compute() — delegated to the oracle's restatement.
template <class M> struct SelfAdjointEigenSolver;
template <> struct SelfAdjointEigenSolver<Matrix3d> {
    Vector3d vals; Matrix3d vecs; bool ok;
    explicit SelfAdjointEigenSolver(const Matrix3d& A) {
        double a[3][3], ev[3], evec[3][3];
        // Eigen reads the lower triangle only
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = (i >= j) ? A(i, j) : A(j, i);
        ok = lo::eig3_sym(a, ev, evec);
        for (int k = 0; k < 3; k++) { vals[k] = ev[k]; for (int i = 0; i < 3; i++) vecs(i, k) = evec[k][i]; }
        if (!ok) { double n = std::nan(""); for (int k = 0; k < 3; k++) vals[k] = n; }
    }
    const Vector3d& eigenvalues() const { return vals; }
    const Matrix3d& eigenvectors() const { return vecs; }
};


// ---- run-time sized matrices: just enough for MarginalizationFactor.cpp:3-71 (ThreadsConstructA, ResidualBlockInfo::Evaluate).
// Eager evaluation, row-major storage; every product / sum is the textbook element-wise definition evaluated left to right,
// which is what Eigen does for these 1-residual blocks (outer products and 1x1 scalings: no reduction order to choose).
struct Dyn {
    int r = 0, c = 0;
    std::vector<double> v;
    Dyn() {}
    Dyn(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return v[(size_t)i * c + j]; }
    const double& operator()(int i, int j) const { return v[(size_t)i * c + j]; }
    double& operator()(int i) { return v[i]; }
    const double& operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
    int rows() const { return r; }
    int cols() const { return c; }
    int size() const { return r * c; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
    void resize(int r_, int c_) { r = r_; c = c_; v.assign((size_t)r_ * c_, 0.0); }
    void resize(int n) { resize(n, 1); }
    void setZero() { std::fill(v.begin(), v.end(), 0.0); }
    Dyn transpose() const { Dyn t(c, r); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) t(j, i) = (*this)(i, j); return t; }
    Dyn rightCols(int n) const { Dyn t(r, n); for (int i = 0; i < r; i++) for (int j = 0; j < n; j++) t(i, j) = (*this)(i, c - n + j); return t; }
    double squaredNorm() const { double s = 0; for (double x : v) s += x * x; return s; }
    Dyn& operator*=(double s) { for (double& x : v) x *= s; return *this; }
    Dyn& operator+=(const Dyn& o) { for (size_t k = 0; k < v.size(); k++) v[k] += o.v[k]; return *this; }
    struct Block {
        Dyn& m; int i0, j0, nr, nc;
        Dyn eval() const { Dyn t(nr, nc); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) t(i, j) = m(i0 + i, j0 + j); return t; }
        Dyn transpose() const { return eval().transpose(); }
        Block& operator+=(const Dyn& o) { for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m(i0 + i, j0 + j) += o(i, j); return *this; }
        Block& operator=(const Dyn& o) { for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m(i0 + i, j0 + j) = o(i, j); return *this; }
    };
    Block block(int i0, int j0, int nr, int nc) { return Block{*this, i0, j0, nr, nc}; }
    Block segment(int i0, int n) { return Block{*this, i0, 0, n, 1}; }
};
inline Dyn operator*(const Dyn& a, const Dyn& b) {
    Dyn t(a.r, b.c);
    for (int i = 0; i < a.r; i++) for (int j = 0; j < b.c; j++) { double s = a(i, 0) * b(0, j); for (int k = 1; k < a.c; k++) s += a(i, k) * b(k, j); t(i, j) = s; }
    return t;
}
inline Dyn operator*(double s, const Dyn& a) { Dyn t = a; for (double& x : t.v) x = s * x; return t; }
inline Dyn operator*(const Dyn& a, double s) { Dyn t = a; for (double& x : t.v) x = x * s; return t; }
inline Dyn operator-(const Dyn& a, const Dyn& b) { Dyn t = a; for (size_t k = 0; k < t.v.size(); k++) t.v[k] = a.v[k] - b.v[k]; return t; }
inline Dyn operator+(const Dyn& a, const Dyn& b) { Dyn t = a; for (size_t k = 0; k < t.v.size(); k++) t.v[k] = a.v[k] + b.v[k]; return t; }
#define REFSHIM_DYN_MATRIX(C_, O_)                                                         \
    template <> struct Matrix<double, Dynamic, C_, O_> : Dyn {                             \
        Matrix() {}                                                                        \
        Matrix(int r_, int c_) : Dyn(r_, c_) {}                                            \
        explicit Matrix(int n) : Dyn(n, 1) {}                                              \
        Matrix(const Dyn& d) : Dyn(d) {}                                                   \
        Matrix& operator=(const Dyn& d) { Dyn::operator=(d); return *this; }               \
    };
REFSHIM_DYN_MATRIX(Dynamic, 0)
REFSHIM_DYN_MATRIX(Dynamic, 1)
REFSHIM_DYN_MATRIX(1, 0)
#undef REFSHIM_DYN_MATRIX
typedef Matrix<double, Dynamic, Dynamic, 0> MatrixXd;
typedef Matrix<double, Dynamic, 1, 0> VectorXd;

}  // namespace Eigen
