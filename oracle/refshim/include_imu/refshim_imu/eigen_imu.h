// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md).
//
// A second, richer Eigen stand-in used by ONE translation unit (ref_imu.cpp), so that the reference's IMU factor
// (L/include/factors/Preintegration.h, ImuFactor.h, utils/math_tools.h) compiles UNMODIFIED: run-time sized value-semantic
// matrices with eager, textbook evaluation (every product is the triple loop, every sum element-wise, left to right), fixed-size
// names as thin wrappers, block / corner views that can be assigned to, the comma initialiser, Map over caller memory (row- or
// column-major), Quaternion with Eigen 3.3's documented formulas, inverse() by Gauss-Jordan with partial pivoting and LLT as the
// plain Cholesky recurrence.  Like refshim/eigen_min.h it adds NO second opinion on Eigen's own arithmetic (Eigen's blocked
// products / LU / LLT associate differently; results agree to rounding, not bit for bit) — what it pins is the reference's own
// statements: the mid-point integration, the F and V matrices of the covariance propagation, the residual and the six Jacobian
// blocks of ImuFactor::Evaluate.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

enum { Dynamic = -1, ColMajor = 0, RowMajor = 1 };

template <class Derived> struct MatrixBase {
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    Derived& derived() { return *static_cast<Derived*>(this); }
    double operator()(int i) const { return derived().at(i); }
    double operator()(int i, int j) const { return derived().at(i, j); }
};

struct Mat : MatrixBase<Mat> {
    typedef double Scalar;
    int r = 0, c = 0;
    std::vector<double> v;     // row-major
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
    int rows() const { return r; }
    int cols() const { return c; }
    double at(int i) const { return v[i]; }
    double at(int i, int j) const { return v[(size_t)i * c + j]; }
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator()(int i, int j) { return v[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return v[(size_t)i * c + j]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double& x() { return v[0]; } double x() const { return v[0]; }
    double& y() { return v[1]; } double y() const { return v[1]; }
    double& z() { return v[2]; } double z() const { return v[2]; }
    void setZero() { for (double& t : v) t = 0.0; }
    void setIdentity() { setZero(); for (int i = 0; i < (r < c ? r : c); i++) (*this)(i, i) = 1.0; }
    Mat transpose() const { Mat t(c, r); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) t(j, i) = (*this)(i, j); return t; }
    double maxCoeff() const { double m = v[0]; for (double t : v) if (t > m) m = t; return m; }
    double minCoeff() const { double m = v[0]; for (double t : v) if (t < m) m = t; return m; }
    double dot(const Mat& o) const { double s = v[0] * o.v[0]; for (size_t k = 1; k < v.size(); k++) s += v[k] * o.v[k]; return s; }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    Mat cross(const Mat& o) const {
        Mat t(3, 1);
        t.v = {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
        return t;
    }
    Mat& operator+=(const Mat& o) { for (size_t k = 0; k < v.size(); k++) v[k] += o.v[k]; return *this; }
    Mat& operator-=(const Mat& o) { for (size_t k = 0; k < v.size(); k++) v[k] -= o.v[k]; return *this; }
    Mat& operator*=(double s) { for (double& t : v) t *= s; return *this; }
    Mat& operator/=(double s) { for (double& t : v) t /= s; return *this; }
    Mat operator-() const { Mat t = *this; for (double& q : t.v) q = -q; return t; }
    // inverse: Gauss-Jordan elimination with partial pivoting (Eigen: PartialPivLU)
    Mat inverse() const {
        assert(r == c);
        const int n = r;
        Mat a = *this, inv(n, n);
        inv.setIdentity();
        for (int k = 0; k < n; k++) {
            int p = k;
            for (int i = k + 1; i < n; i++) if (std::fabs(a(i, k)) > std::fabs(a(p, k))) p = i;
            if (p != k) for (int j = 0; j < n; j++) { std::swap(a(k, j), a(p, j)); std::swap(inv(k, j), inv(p, j)); }
            const double d = a(k, k);
            for (int j = 0; j < n; j++) { a(k, j) /= d; inv(k, j) /= d; }
            for (int i = 0; i < n; i++) if (i != k) {
                const double f = a(i, k);
                if (f != 0.0) for (int j = 0; j < n; j++) { a(i, j) -= f * a(k, j); inv(i, j) -= f * inv(k, j); }
            }
        }
        return inv;
    }
    // ---- views
    struct BlockRef {
        Mat& m; int i0, j0, nr, nc;
        operator Mat() const { Mat t(nr, nc); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) t(i, j) = m(i0 + i, j0 + j); return t; }
        BlockRef& operator=(const Mat& o) { assert(o.r == nr && o.c == nc); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m(i0 + i, j0 + j) = o(i, j); return *this; }
        BlockRef& operator=(const BlockRef& o) { return *this = (Mat)o; }
        BlockRef& operator<<(const Mat& o) { return *this = o; }
        Mat transpose() const { return ((Mat) * this).transpose(); }
    };
    template <int BR, int BC> BlockRef block(int i0, int j0) { return BlockRef{*this, i0, j0, BR, BC}; }
    template <int BR, int BC> Mat block(int i0, int j0) const { return (Mat)BlockRef{const_cast<Mat&>(*this), i0, j0, BR, BC}; }
    BlockRef block(int i0, int j0, int nr, int nc) { return BlockRef{*this, i0, j0, nr, nc}; }
    template <int BR, int BC> Mat topLeftCorner() const { return block<BR, BC>(0, 0); }
    template <int BR, int BC> Mat bottomRightCorner() const { return block<BR, BC>(r - BR, c - BC); }
    // ---- comma initialiser (row-major fill)
    struct Comma { Mat& m; int k; Comma& operator,(double t) { m.v[k++] = t; return *this; } };
    Comma operator<<(double first) { v[0] = first; return Comma{*this, 1}; }
};
inline Mat operator+(const Mat& a, const Mat& b) { assert(a.r == b.r && a.c == b.c); Mat t = a; for (size_t k = 0; k < t.v.size(); k++) t.v[k] = a.v[k] + b.v[k]; return t; }
inline Mat operator-(const Mat& a, const Mat& b) { assert(a.r == b.r && a.c == b.c); Mat t = a; for (size_t k = 0; k < t.v.size(); k++) t.v[k] = a.v[k] - b.v[k]; return t; }
inline Mat operator*(const Mat& a, const Mat& b) {
    assert(a.c == b.r);
    Mat t(a.r, b.c);
    for (int i = 0; i < a.r; i++) for (int j = 0; j < b.c; j++) { double s = a(i, 0) * b(0, j); for (int k = 1; k < a.c; k++) s += a(i, k) * b(k, j); t(i, j) = s; }
    return t;
}
inline Mat operator*(double s, const Mat& a) { Mat t = a; for (double& q : t.v) q = s * q; return t; }
inline Mat operator*(const Mat& a, double s) { Mat t = a; for (double& q : t.v) q = q * s; return t; }
inline Mat operator/(const Mat& a, double s) { Mat t = a; for (double& q : t.v) q = q / s; return t; }

template <class T, int R, int C, int O = 0> struct Matrix : Mat {
    typedef T Scalar;
    Matrix() : Mat(R == Dynamic ? 0 : R, C == Dynamic ? 0 : C) {}
    Matrix(int r_, int c_) : Mat(r_, c_) {}
    Matrix(double a, double b, double cc) : Mat(3, 1) { v = {a, b, cc}; }
    Matrix(const Mat& m) : Mat(m) {}
    Matrix(const Mat::BlockRef& b) : Mat((Mat)b) {}
    template <class D> Matrix(const MatrixBase<D>& o) : Mat(o.derived()) {}
    Matrix& operator=(const Mat& m) { Mat::operator=(m); return *this; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Zero(int r_, int c_) { return Matrix(r_, c_); }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(int r_, int c_) { Matrix m(r_, c_); m.setIdentity(); return m; }
};
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;

template <class M> struct Map;
template <class T, int R, int C, int O> struct Map<Matrix<T, R, C, O>> {
    double* p;
    explicit Map(double* p_) : p(p_) {}
    double& ref(int i, int j) const { return O == RowMajor ? p[(size_t)i * C + j] : p[(size_t)j * R + i]; }
    operator Mat() const { Mat t(R, C); for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t(i, j) = ref(i, j); return t; }
    Map& operator=(const Mat& m) { assert(m.r == R && m.c == C); for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) ref(i, j) = m(i, j); return *this; }
    void setZero() { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) ref(i, j) = 0.0; }
    double maxCoeff() const { return ((Mat) * this).maxCoeff(); }
    double minCoeff() const { return ((Mat) * this).minCoeff(); }
    struct BlockRef {
        const Map& m; int i0, j0, nr, nc;
        BlockRef& operator=(const Mat& o) { assert(o.r == nr && o.c == nc); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m.ref(i0 + i, j0 + j) = o(i, j); return *this; }
    };
    template <int BR, int BC> BlockRef block(int i0, int j0) { return BlockRef{*this, i0, j0, BR, BC}; }
};

template <class M> struct LLT {
    Mat L;
    explicit LLT(const Mat& a) : L(a.r, a.c) {
        const int n = a.r;
        for (int j = 0; j < n; j++) {
            double d = a(j, j);
            for (int k = 0; k < j; k++) d -= L(j, k) * L(j, k);
            const double ljj = std::sqrt(d);
            L(j, j) = ljj;
            for (int i = j + 1; i < n; i++) { double s = a(i, j); for (int k = 0; k < j; k++) s -= L(i, k) * L(j, k); L(i, j) = s / ljj; }
        }
    }
    Mat matrixL() const { return L; }
};

template <class T> struct AngleAxis;   // named by math_tools.h templates that are never instantiated here

template <class Derived> struct QuaternionBase {
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    double w() const { return derived().qw; }
    Mat vec() const { Mat t(3, 1); t.v = {derived().qx, derived().qy, derived().qz}; return t; }
};
template <class T> struct Quaternion : QuaternionBase<Quaternion<T>> {
    typedef T Scalar;
    T qw, qx, qy, qz;
    Quaternion() : qw(), qx(), qy(), qz() {}
    Quaternion(double w_, double x_, double y_, double z_) : qw(w_), qx(x_), qy(y_), qz(z_) {}   // (w, x, y, z) like Eigen
    static Quaternion Identity() { return Quaternion(1, 0, 0, 0); }
    void setIdentity() { *this = Identity(); }
    T& w() { return qw; } T w() const { return qw; }
    T& x() { return qx; } T x() const { return qx; }
    T& y() { return qy; } T y() const { return qy; }
    T& z() { return qz; } T z() const { return qz; }
    Mat vec() const { Mat t(3, 1); t.v = {qx, qy, qz}; return t; }
    Quaternion operator*(const Quaternion& b) const {      // internal::quat_product, generic path
        const Quaternion& a = *this;
        return Quaternion(a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz, a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy,
                          a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz, a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx);
    }
    Mat operator*(const Mat& vv) const {                   // QuaternionBase::_transformVector — no normalisation inside
        Mat u = vec();
        Mat uv = u.cross(vv);
        uv += uv;
        return vv + uv * qw + u.cross(uv);
    }
    T squaredNorm() const { return qx * qx + qy * qy + qz * qz + qw * qw; }
    T norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { const T n = norm(); qx /= n; qy /= n; qz /= n; qw /= n; }
    Quaternion normalized() const { Quaternion q = *this; q.normalize(); return q; }
    Quaternion inverse() const {
        const T n2 = squaredNorm();
        if (n2 > T(0)) return Quaternion(qw / n2, -qx / n2, -qy / n2, -qz / n2);
        return Quaternion(0, 0, 0, 0);
    }
    Matrix<T, 3, 3> toRotationMatrix() const {             // QuaternionBase::toRotationMatrix
        const T tx = T(2) * qx, ty = T(2) * qy, tz = T(2) * qz;
        const T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        Matrix<T, 3, 3> R;
        R(0, 0) = T(1) - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
        R(1, 0) = txy + twz; R(1, 1) = T(1) - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = T(1) - (txx + tyy);
        return R;
    }
};
typedef Quaternion<double> Quaterniond;

}  // namespace Eigen
