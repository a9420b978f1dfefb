// ORACLE — TEST INFRASTRUCTURE ONLY.  ThreadsConstructA and ResidualBlockInfo::Evaluate of the reference's marginalisation
// (LiLi-OM/src/MarginalizationFactor.cpp:3-71: the loss corrector every lidar block goes through, SURVEY §8 a-17, and the
// A += J_i^T J_j / b += J_i^T r assembly with rightCols(3) of the quaternion block, a-18 / f-3) compiled from the reference
// text: the two functions are sliced out of the file at build time (oracle/refshim/Makefile, lines 3-71; the rest of the
// file — Schur complement, eigen-decomposition — needs more of dynamic Eigen than is stood in for) and compiled against the
// reference's own include/factors/MarginalizationFactor.h and LidarKeyframeFactor.h.
#include <cstring>
#include <memory>
#include <vector>
#include "refshim_deps.h"
#include "factors/MarginalizationFactor.h"
#include "factors/LidarKeyframeFactor.h"
#include "gen/marg_L.inc"

extern "C" {

// The lidar blocks of ONE keyframe, as L/src/BackendFusion.cpp:1112-1160 adds them to the marginalisation: parameter blocks
// (t[3], q[4]) at local indices idx_t / idx_q of a pos x pos system, CauchyLoss(1.0), records as in ref_backend.cpp
// (surf: cp, n, d, score; edge: cp, A, B, s).  Outputs: rows (n, 8) = r, J_t(3), J_q(4) AFTER Evaluate() (robustified), and the
// dense A (pos x pos, row-major), b (pos) that ThreadsConstructA accumulates from zero.
void ref_marg_lidar(const double* surf_rec, int n_surf, const double* edge_rec, int n_edge, const double qlb[4], const double tlb[3],
                    const double t[3], const double q[4], int pos, int idx_t, int idx_q, double* rows, double* A, double* b) {
    Eigen::Quaterniond q_lb(qlb[0], qlb[1], qlb[2], qlb[3]);
    Eigen::Vector3d t_lb(tlb[0], tlb[1], tlb[2]);
    double tt[3] = {t[0], t[1], t[2]}, qq[4] = {q[0], q[1], q[2], q[3]};
    ceres::LossFunction* loss = new ceres::CauchyLoss(1.0);
    ThreadsStruct ts;
    ts.A = Eigen::MatrixXd(pos, pos); ts.b = Eigen::VectorXd(pos);
    ts.A.setZero(); ts.b.setZero();
    ts.parameter_block_size[reinterpret_cast<long>(tt)] = 3; ts.parameter_block_idx[reinterpret_cast<long>(tt)] = idx_t;
    ts.parameter_block_size[reinterpret_cast<long>(qq)] = 4; ts.parameter_block_idx[reinterpret_cast<long>(qq)] = idx_q;
    std::vector<std::unique_ptr<ceres::CostFunction>> owned;
    std::vector<std::unique_ptr<ResidualBlockInfo>> infos;
    auto add = [&](ceres::CostFunction* f) {
        owned.emplace_back(f);
        infos.emplace_back(new ResidualBlockInfo(f, loss, std::vector<double*>{tt, qq}, std::vector<int>{}));
        infos.back()->Evaluate();
        ts.sub_factors.push_back(infos.back().get());
    };
    for (int i = 0; i < n_edge; i++) {
        const double* e = edge_rec + 10 * i;
        add(LidarEdgeFactor::Create(Eigen::Vector3d(e[0], e[1], e[2]), Eigen::Vector3d(e[3], e[4], e[5]), Eigen::Vector3d(e[6], e[7], e[8]), q_lb, t_lb, e[9]));
    }
    for (int i = 0; i < n_surf; i++) {
        const double* s = surf_rec + 8 * i;
        add(LidarPlaneNormFactor::Create(Eigen::Vector3d(s[0], s[1], s[2]), Eigen::Vector3d(s[3], s[4], s[5]), q_lb, t_lb, s[6], s[7]));
    }
    for (size_t k = 0; k < infos.size(); k++) {
        const ResidualBlockInfo& I = *infos[k];
        double* o = rows + 8 * k;
        o[0] = I.residuals[0];
        for (int c = 0; c < 3; c++) o[1 + c] = I.jacobians[0](0, c);
        for (int c = 0; c < 4; c++) o[4 + c] = I.jacobians[1](0, c);
        delete[] I.raw_jacobians;
    }
    ThreadsConstructA(&ts);
    std::memcpy(A, ts.A.data(), sizeof(double) * pos * pos);
    std::memcpy(b, ts.b.data(), sizeof(double) * pos);
    delete loss;
}

}  // extern "C"
