// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's include/factors/LidarKeyframeFactor.h UNMODIFIED
// and evaluates its three functors through the reference's own seam — XxxFactor::Create(...) followed by
// ceres::CostFunction::Evaluate — with the Jet / AutoDiffCostFunction stand-ins of
// refshim/ceres_min.h.  Output layout of every call: out[0] = residual, out[1..] = the two Jacobian blocks in
// the functor's parameter order, row-major.
#include "refshim_deps.h"
#include "factors/LidarKeyframeFactor.h"
#include <memory>

static Eigen::Vector3d v3(const double* p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
static Eigen::Quaterniond q4(const double* q) { return Eigen::Quaterniond(q[0], q[1], q[2], q[3]); }   // (w, x, y, z)

extern "C" {

// LidarEdgeFactor(cp, A, B, qlb, tlb, s); parameters (t[3], q[4] wxyz).  out[8] = r, dr/dt(3), dr/dq(4)
int ref_edge_factor(const double cp[3], const double a[3], const double b[3], const double qlb[4], const double tlb[3], double s,
                    const double t[3], const double q[4], double out[8], int want_jac) {
    std::unique_ptr<ceres::CostFunction> f(LidarEdgeFactor::Create(v3(cp), v3(a), v3(b), q4(qlb), v3(tlb), s));
    const double* params[2] = {t, q};
    double* jac[2] = {out + 1, out + 4};
    return f->Evaluate(params, out, want_jac ? jac : nullptr) ? 0 : 1;
}
// LidarPlaneNormFactor(cp, n, qlb, tlb, d, score); parameters (t[3], q[4]).  out[8]
int ref_plane_factor(const double cp[3], const double n[3], const double qlb[4], const double tlb[3], double d, double score,
                     const double t[3], const double q[4], double out[8], int want_jac) {
    std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormFactor::Create(v3(cp), v3(n), q4(qlb), v3(tlb), d, score));
    const double* params[2] = {t, q};
    double* jac[2] = {out + 1, out + 4};
    return f->Evaluate(params, out, want_jac ? jac : nullptr) ? 0 : 1;
}
// LidarPlaneNormIncreFactor(cp, n, d); parameters (q[4], t[3]).  out[8] = r, dr/dq(4), dr/dt(3)
int ref_plane_incre_factor(const double cp[3], const double n[3], double d,
                           const double q[4], const double t[3], double out[8], int want_jac) {
    std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormIncreFactor::Create(v3(cp), v3(n), d));
    const double* params[2] = {q, t};
    double* jac[2] = {out + 1, out + 5};
    return f->Evaluate(params, out, want_jac ? jac : nullptr) ? 0 : 1;
}

}  // extern "C"
