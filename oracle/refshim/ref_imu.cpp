// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's IMU pre-integration and IMU factor UNMODIFIED
// (L/include/factors/Preintegration.h:23-211, L/include/factors/ImuFactor.h:11-147, L/include/utils/math_tools.h) with the Eigen stand-in
// of include_imu/refshim_imu/eigen_imu.h and evaluates them through the reference's own seam: `new Preintegration(acc0, gyr0, ba, bg)`,
// push_back(dt, acc, gyr) per IMU sample (L/src/BackendFusion.cpp:823), `ImuFactor(pre).Evaluate(parameters, residuals, jacobians)` with
// the six parameter blocks of L/src/BackendFusion.cpp:911-917 (t_i[3], q_i[4] wxyz, speed-bias_i[9], t_j, q_j, speed-bias_j).
// Pins the numpy restatement oracle/lo_window.py (BASELINE configs[4]).
#include <cstdio>
#include <string>
#include <vector>
using std::vector;
namespace ros { struct NodeHandle {}; }          // Preintegration carries an (unused) ros::NodeHandle member
#define ROS_WARN(...) do {} while (0)
#if defined(LILI_WITH_REFERENCE_DEPS)
#include <Eigen/Dense>                            // REAL_DEPS build: the machine's Eigen (and Ceres, through ImuFactor.h)
#else
#include "refshim_imu/eigen_imu.h"
#endif
#include "factors/ImuFactor.h"
#include <memory>

static Eigen::Vector3d v3(const double* p) { return Eigen::Vector3d(p[0], p[1], p[2]); }

static std::unique_ptr<Preintegration> run_pre(int n, const double* dt, const double* acc, const double* gyr, const double acc0[3], const double gyr0[3],
                                               const double ba[3], const double bg[3]) {
    std::unique_ptr<Preintegration> p(new Preintegration(v3(acc0), v3(gyr0), v3(ba), v3(bg)));
    for (int k = 0; k < n; k++) p->push_back(dt[k], v3(acc + 3 * k), v3(gyr + 3 * k));
    return p;
}

extern "C" {

// state out: delta_p[3], delta_q[4] (w,x,y,z), delta_v[3], sum_dt; jacobian / covariance 15x15 row-major
int ref_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3],
                     double state[11], double jac[225], double cov[225]) {
    auto p = run_pre(n, dt, acc, gyr, acc0, gyr0, ba, bg);
    for (int i = 0; i < 3; i++) { state[i] = p->delta_p_(i); state[7 + i] = p->delta_v_(i); }
    state[3] = p->delta_q_.w(); state[4] = p->delta_q_.x(); state[5] = p->delta_q_.y(); state[6] = p->delta_q_.z();
    state[10] = p->sum_dt_;
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) { jac[i * 15 + j] = p->jacobian_(i, j); cov[i * 15 + j] = p->covariance_(i, j); }
    return 0;
}

// params = [t_i(3) q_i(4) sb_i(9) t_j(3) q_j(4) sb_j(9)] (32 doubles); residual[15]; jac = the six row-major blocks one after the other
// (15x3, 15x4, 15x9, 15x3, 15x4, 15x9 = 480 doubles)
int ref_imu_factor(int n, const double* dt, const double* acc, const double* gyr, const double acc0[3], const double gyr0[3], const double ba[3], const double bg[3],
                   const double params[32], double residual[15], double jac[480]) {
    auto p = run_pre(n, dt, acc, gyr, acc0, gyr0, ba, bg);
    ImuFactor f(p.get());
    const double* blocks[6] = {params, params + 3, params + 7, params + 16, params + 19, params + 23};
    double* jb[6] = {jac, jac + 45, jac + 105, jac + 240, jac + 285, jac + 345};
    return f.Evaluate(blocks, residual, jac ? jb : nullptr) ? 0 : 1;
}

}  // extern "C"
