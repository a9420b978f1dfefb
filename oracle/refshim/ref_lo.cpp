// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's LiLi-OM/src/LidarOdometry.cpp UNMODIFIED (the whole
// front-end node: poseInitialization, buildLocalMap, downSampleCloud, updateTransformationWithCeres,
// findCorrespondingSurfFeatures, savePoses, computeRelative, keyframe logic) into oracle/_ref/libref_lo.so and drives it
// in-process.  Access control is lifted with a macro (no source edit) so that the driver can read the node's private
// clouds.  ceres::Solve is NOT Ceres here (oracle/refshim/README.md): the installed hook
//   1. records, for every residual block the reference added, the record it was built from (cp, weight*n, weight*d)
//      and the raw residual + Jacobians returned by the reference's own CostFunction::Evaluate at the current pose,
//   2. applies the loss the reference attached (HuberLoss(0.1)) with the Triggs corrector (closed form of
//      ceres/corrector.cc, the same algebra as L/src/MarginalizationFactor.cpp:44-70), sums the 8x8 Gram and takes ONE
//      Gauss-Newton step on ceres::QuaternionParameterization (oracle/lo_s2m.cpp: lo_gn_step) — the protocol the
//      product's lili_s2m_iterate implements — and writes the pose back into the reference's parameter blocks.
// So everything up to and including "what is handed to the solver" is the reference's code; the solver step is the
// documented stand-in shared with the oracle and the GPU path.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>
#include "refshim_deps.h"
// Built twice (oracle/refshim/Makefile): libref_lo.so from LiLi-OM/src/LidarOdometry.cpp (PointType = pcl::PointXYZINormal, 48-byte rows) and, with
// -DREF_FLAVOUR_ROT, libref_lo_R.so from LiLi-OM-ROT/src/LidarOdometry.cpp (PointType = pcl::PointXYZI, 32-byte rows) — the include path decides which file this is.
#define private public
#define main ref_lo_node_main
#include "src/LidarOdometry.cpp"
#undef main
#undef private

extern "C" int lo_gn_step(const double gram[64], double t[3], double q[4], double delta_out[6]);   // oracle/lo_s2m.cpp

namespace {
struct SolveLog {
    double pose_in[7];                 // qw qx qy qz | tx ty tz   (the reference's transformInc layout)
    double pose_out[7];
    std::vector<double> records;       // per block: cp(3), n(3), d
    std::vector<double> rows;          // per block: r, dr/dq(4), dr/dt(3)  — raw, before the loss
    std::vector<float> map_xyzc;       // the cloud the node's kd-tree was built on for this solve (x y z curvature)
    std::vector<float> queries;        // surf_last_ds (x y z curvature)
    int gn_status;
};
std::vector<SolveLog> g_log;
LidarOdometry* g_node = nullptr;

void hook(const ceres::Solver::Options&, ceres::Problem* p, ceres::Solver::Summary* s) {
    SolveLog L;
    double* q = nullptr; double* t = nullptr;
    double gram[64] = {0};
    for (const ceres::ResidualBlock& b : p->blocks) {
        typedef ceres::AutoDiffCostFunction<LidarPlaneNormIncreFactor, 1, 4, 3> CF;
        const CF* cf = dynamic_cast<const CF*>(b.cost);
        if (!cf) abort();
        q = b.params[0]; t = b.params[1];
        const LidarPlaneNormIncreFactor* f = cf->functor();
        const double rec[7] = {f->curr_point.x(), f->curr_point.y(), f->curr_point.z(), f->plane_unit_norm.x(), f->plane_unit_norm.y(),
                               f->plane_unit_norm.z(), f->negative_OA_dot_norm};
        L.records.insert(L.records.end(), rec, rec + 7);
        double r, jq[4], jt[3];
        double* jac[2] = {jq, jt};
        const double* params[2] = {q, t};
        b.cost->Evaluate(params, &r, jac);
        const double row[8] = {r, jq[0], jq[1], jq[2], jq[3], jt[0], jt[1], jt[2]};
        L.rows.insert(L.rows.end(), row, row + 8);
        // loss + corrector (one residual): ceres/corrector.cc
        double Jr[8] = {jt[0], jt[1], jt[2], jq[0], jq[1], jq[2], jq[3], r};   // oracle layout: t(3), q(4), r
        if (b.loss) {
            double sq = r * r, rho[3];
            b.loss->Evaluate(sq, rho);
            const double sqrt_rho1 = std::sqrt(rho[1]);
            double residual_scaling, alpha_sq_norm;
            if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
            else {
                const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
                const double alpha = 1.0 - std::sqrt(D);
                residual_scaling = sqrt_rho1 / (1 - alpha);
                alpha_sq_norm = alpha / sq;
            }
            for (int k = 0; k < 7; k++) Jr[k] = sqrt_rho1 * (Jr[k] - alpha_sq_norm * Jr[7] * (Jr[7] * Jr[k]));
            Jr[7] *= residual_scaling;
        }
        for (int a = 0; a < 8; a++) for (int c = 0; c < 8; c++) gram[a * 8 + c] += Jr[a] * Jr[c];
    }
    L.map_xyzc = refshim::last_tree_input();
#ifdef REF_FLAVOUR_ROT
    if (g_node) for (const auto& pt : g_node->surf_last_ds->points) { const float v[4] = {pt.x, pt.y, pt.z, pt.intensity}; L.queries.insert(L.queries.end(), v, v + 4); }
#else
    if (g_node) for (const auto& pt : g_node->surf_last_ds->points) { const float v[4] = {pt.x, pt.y, pt.z, pt.curvature}; L.queries.insert(L.queries.end(), v, v + 4); }
#endif
    L.gn_status = -1;
    if (q && t) {
        for (int k = 0; k < 4; k++) L.pose_in[k] = q[k];
        for (int k = 0; k < 3; k++) L.pose_in[4 + k] = t[k];
        L.gn_status = lo_gn_step(gram, t, q, nullptr);
        for (int k = 0; k < 4; k++) L.pose_out[k] = q[k];
        for (int k = 0; k < 3; k++) L.pose_out[4 + k] = t[k];
    }
    if (s) s->num_blocks = (int)p->blocks.size();
    g_log.push_back(std::move(L));
}
}  // namespace

extern "C" {

void ref_set_verbose(int v) { refshim::verbose() = v != 0; }
void ref_param_clear() { refshim::params().clear(); }
void ref_param_num(const char* key, double v) { refshim::params()[key] = refshim::ParamVal{0, v, ""}; }
void ref_param_str(const char* key, const char* v) { refshim::params()[key] = refshim::ParamVal{1, 0.0, v}; }

void* ref_lo_create() {
    refshim::sink().clear(); g_log.clear();
    ceres::solve_hook() = hook;
    g_node = new LidarOdometry();
    return g_node;
}
void ref_lo_destroy(void* h) { delete (LidarOdometry*)h; g_node = nullptr; }

#ifdef REF_FLAVOUR_ROT
constexpr uint32_t kStep = 32;      // pcl::PointXYZI
#else
constexpr uint32_t kStep = 48;      // pcl::PointXYZINormal
#endif
int ref_lo_point_floats() { return (int)(kStep / 4); }
static std::shared_ptr<sensor_msgs::PointCloud2> msg48(double stamp, const float* rows12, int n) {
    auto m = std::make_shared<sensor_msgs::PointCloud2>();
    m->header.stamp.t = stamp; m->point_step = kStep; m->width = (uint32_t)n; m->row_step = kStep * (uint32_t)n;
    m->data.assign((const uint8_t*)rows12, (const uint8_t*)rows12 + (size_t)n * kStep);
    return m;
}
// One frame = the three clouds Preprocessing publishes (rows in the node's PointType layout: 48-byte PointXYZINormal, ROT flavour 32-byte PointXYZI), then the node's run().
void ref_lo_frame(void* h, double stamp, const float* edge, int n_edge, const float* surf, int n_surf, const float* full, int n_full) {
    LidarOdometry* lo = (LidarOdometry*)h;
    lo->laserCloudLessSharpHandler(msg48(stamp, edge, n_edge));
    lo->laserCloudLessFlatHandler(msg48(stamp, surf, n_surf));
    lo->FullPointCloudHandler(msg48(stamp, full, n_full));
    lo->run();
}
void ref_lo_pose(void* h, double abs_pose[7], double rel_pose[7], int* is_kf) {
    LidarOdometry* lo = (LidarOdometry*)h;
    for (int k = 0; k < 7; k++) { abs_pose[k] = lo->abs_pose[k]; rel_pose[k] = lo->rel_pose[k]; }
    *is_kf = lo->kf ? 1 : 0;
}

int ref_lo_n_solves() { return (int)g_log.size(); }
// sizes[4] = n_blocks, n_map, n_queries, gn_status
void ref_lo_solve_info(int i, int sizes[4], double pose_in[7], double pose_out[7]) {
    const SolveLog& L = g_log.at(i);
    sizes[0] = (int)(L.rows.size() / 8); sizes[1] = (int)(L.map_xyzc.size() / 4); sizes[2] = (int)(L.queries.size() / 4); sizes[3] = L.gn_status;
    std::memcpy(pose_in, L.pose_in, sizeof(L.pose_in)); std::memcpy(pose_out, L.pose_out, sizeof(L.pose_out));
}
void ref_lo_solve_data(int i, double* records7, double* rows8, float* map4, float* queries4) {
    const SolveLog& L = g_log.at(i);
    if (records7 && !L.records.empty()) std::memcpy(records7, L.records.data(), L.records.size() * sizeof(double));
    if (rows8 && !L.rows.empty()) std::memcpy(rows8, L.rows.data(), L.rows.size() * sizeof(double));
    if (map4 && !L.map_xyzc.empty()) std::memcpy(map4, L.map_xyzc.data(), L.map_xyzc.size() * sizeof(float));
    if (queries4 && !L.queries.empty()) std::memcpy(queries4, L.queries.data(), L.queries.size() * sizeof(float));
}

int ref_n_published() { return (int)refshim::sink().size(); }
int ref_msg_info(int i, char* topic, double* stamp, int* point_step) {
    const refshim::PubMsg& m = refshim::sink().at(i);
    std::strncpy(topic, m.topic.c_str(), 63); topic[63] = 0;
    *stamp = m.stamp; *point_step = (int)m.point_step;
    return m.point_step ? (int)(m.data.size() / m.point_step) : (int)(m.data.size() / 8);
}
void ref_msg_data(int i, void* out) {
    const refshim::PubMsg& m = refshim::sink().at(i);
    if (!m.data.empty()) std::memcpy(out, m.data.data(), m.data.size());
}

}  // extern "C"
