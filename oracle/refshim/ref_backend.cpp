// ORACLE — TEST INFRASTRUCTURE ONLY.  The back-end matcher's association functions of the reference, compiled from the
// reference's own text: BackendFusion.cpp as a whole drags in GTSAM, ceres::Problem with IMU / marginalisation factors and
// dynamic Eigen and is not stood in for; instead oracle/refshim/Makefile SLICES, at build time, the three member functions
// of the hot path out of the file where it lies —
//     transformPoint(pi, po, quaternion, transition)      L/src/BackendFusion.cpp:695-711   R/src/BackendFusion.cpp:624-632
//     findCorrespondingCornerFeatures(idx, q, t)          L:1531-1599                        R:1394-1462
//     findCorrespondingSurfFeatures(idx, q, t)            L:1601-1681                        R:1464-1520
// — into a temporary oracle/_ref/gen/backend_{L,R}.inc that exists only while this file is compiled (the Makefile removes it
// again; no reference text stays in the tree) and this file #includes that text, unmodified,
// inside a harness class whose data members carry the names and types the reference's class declares for them
// (L:62-76,93,105-110,119-120,154,157,216-223).  The Makefile checks that each slice starts with the expected signature.
// The residual blocks are then created exactly as the reference's optimisation loop does (L:936-972, R:836-866: the
// expressions are restated below, with their C++ promotion rules) through the reference's own factor header, and evaluated
// through ceres::CostFunction::Evaluate.
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#include "refshim_deps.h"
#include "utils/common.h"
#include "utils/math_tools.h"
#include "factors/LidarKeyframeFactor.h"

class BackendSlice {
public:
    vector<pcl::PointCloud<PointType>::Ptr> edge_lasts_ds, surf_lasts_ds;
    pcl::PointCloud<PointType>::Ptr edge_local_map_ds, surf_local_map_ds;
    vector<pcl::PointCloud<PointType>::Ptr> vec_edge_cur_pts, vec_edge_match_j, vec_edge_match_l;
    vector<pcl::PointCloud<PointType>::Ptr> vec_surf_cur_pts, vec_surf_normal;
    vector<vector<double>> vec_surf_scores;
    PointType pt_in_local, pt_in_map;
    pcl::KdTreeFLANN<PointType>::Ptr kd_tree_edge_local_map, kd_tree_surf_local_map;
    vector<int> pt_search_idx;
    vector<float> pt_search_sq_dists;
    vector<int> vec_edge_res_cnt, vec_surf_res_cnt;
    int slide_window_width = 1;
    vector<int> keyframe_idx;
    double kd_max_radius = 1.0, surf_dist_thres = 0.1, lidar_const = 1.0, reflect_thres = 0;

#if defined(REF_FLAVOUR_ROT)
#include "gen/backend_R.inc"
#else
#include "gen/backend_L.inc"
#endif
};

namespace {
template <class P> void set_aux(P& p, float aux);
template <> void set_aux(pcl::PointXYZI& p, float aux) { p.intensity = aux; }
template <> void set_aux(pcl::PointXYZINormal& p, float aux) { p.curvature = aux; }   // Livox: reflectivity lives in curvature

pcl::PointCloud<PointType>::Ptr cloud_of(const float* xyza, int n) {
    pcl::PointCloud<PointType>::Ptr c(new pcl::PointCloud<PointType>());
    c->points.resize(n);
    for (int i = 0; i < n; i++) { PointType p; p.x = xyza[4 * i]; p.y = xyza[4 * i + 1]; p.z = xyza[4 * i + 2]; set_aux(p, xyza[4 * i + 3]); c->points[i] = p; }
    c->width = (uint32_t)n; c->height = 1;
    return c;
}
}  // namespace

extern "C" {

// One keyframe of a window of width 1 (idVec 0): maps and queries as x y z aux rows (aux = reflectivity/curvature for the
// Livox flavour, unused for ROT).  (q_assoc, t_assoc) is the pose the reference passes to the find* functions
// (Q2 * q_lb^-1, T2 - Q2 t_lb — computed by the caller).  Outputs: surf records (n, 8) = cp(3), weight*n(3), weight*d, score;
// edge records (n, 10) = cp(3), A(3), B(3), s (pt_in_local.intensity = lidar_const).  Returns 0.
int ref_backend_associate(const float* surf_map, int n_surf_map, const float* edge_map, int n_edge_map,
                          const float* surf_q, int n_surf_q, const float* edge_q, int n_edge_q,
                          const double q_assoc[4], const double t_assoc[3],
                          double kd_max_radius, double surf_dist_thres, double lidar_const, double reflect_thres,
                          double* surf_rec, int* n_surf_rec, double* edge_rec, int* n_edge_rec) {
    BackendSlice B;
    B.kd_max_radius = kd_max_radius; B.surf_dist_thres = surf_dist_thres; B.lidar_const = lidar_const; B.reflect_thres = reflect_thres;
    B.slide_window_width = 1;
    B.keyframe_idx = {1};                       // the reference calls find*(idx - 1, ...) with idx = keyframe_idx[...] (L:934-935); idVec = idx - keyframe_idx[..] + 1
    const int idx = 0;                          // => idVec = 0 - 1 + 1 = 0
    B.surf_lasts_ds = {cloud_of(surf_q, n_surf_q)};
    B.edge_lasts_ds = {cloud_of(edge_q, n_edge_q)};
    B.surf_local_map_ds = cloud_of(surf_map, n_surf_map);
    B.edge_local_map_ds = cloud_of(edge_map, n_edge_map);
    B.kd_tree_surf_local_map.reset(new pcl::KdTreeFLANN<PointType>());
    B.kd_tree_edge_local_map.reset(new pcl::KdTreeFLANN<PointType>());
    B.kd_tree_surf_local_map->setInputCloud(B.surf_local_map_ds);      // L:839-840
    B.kd_tree_edge_local_map->setInputCloud(B.edge_local_map_ds);
    for (auto* v : {&B.vec_edge_cur_pts, &B.vec_edge_match_j, &B.vec_edge_match_l, &B.vec_surf_cur_pts, &B.vec_surf_normal})
        v->push_back(pcl::PointCloud<PointType>::Ptr(new pcl::PointCloud<PointType>()));
    B.vec_surf_scores.resize(1); B.vec_edge_res_cnt = {0}; B.vec_surf_res_cnt = {0};
    Eigen::Quaterniond Q(q_assoc[0], q_assoc[1], q_assoc[2], q_assoc[3]);
    Eigen::Vector3d T(t_assoc[0], t_assoc[1], t_assoc[2]);
    B.findCorrespondingSurfFeatures(idx, Q, T);
    B.findCorrespondingCornerFeatures(idx, Q, T);
    *n_surf_rec = B.vec_surf_res_cnt[0]; *n_edge_rec = B.vec_edge_res_cnt[0];
    for (int i = 0; i < B.vec_surf_res_cnt[0]; i++) {
        const PointType& c = B.vec_surf_cur_pts[0]->points[i]; const PointType& n = B.vec_surf_normal[0]->points[i];
        const double r[8] = {c.x, c.y, c.z, n.x, n.y, n.z, n.intensity, B.vec_surf_scores[0][i]};
        std::memcpy(surf_rec + 8 * i, r, sizeof(r));
    }
    for (int i = 0; i < B.vec_edge_res_cnt[0]; i++) {
        const PointType& c = B.vec_edge_cur_pts[0]->points[i]; const PointType& a = B.vec_edge_match_j[0]->points[i]; const PointType& b = B.vec_edge_match_l[0]->points[i];
        const double r[10] = {c.x, c.y, c.z, a.x, a.y, a.z, b.x, b.y, b.z, c.intensity};
        std::memcpy(edge_rec + 10 * i, r, sizeof(r));
    }
    return 0;
}

// The residual blocks of one keyframe as the reference's optimisation loop creates them (L:936-972 / R:836-866), evaluated at
// (t, q): rows (n, 8) = r, dr/dt(3), dr/dq(4) — raw, before CauchyLoss.  float_/int_ arguments keep the C++ types the
// reference's expressions have, so the ROT count scaling promotes exactly as there:
//   ROT edge:  points[i].intensity * 200 / vec_edge_res_cnt[idVec]        float * int / int  -> float arithmetic
//   ROT surf:  vec_surf_scores[idVec][i] * 1000 / vec_surf_res_cnt[idVec]  double * int / int -> double arithmetic
void ref_backend_rows(const double* surf_rec, int n_surf, const double* edge_rec, int n_edge, const double qlb[4], const double tlb[3],
                      const double t[3], const double q[4], double* surf_rows, double* edge_rows) {
    Eigen::Quaterniond q_lb(qlb[0], qlb[1], qlb[2], qlb[3]);
    Eigen::Vector3d t_lb(tlb[0], tlb[1], tlb[2]);
    const double* params[2] = {t, q};
    for (int i = 0; i < n_edge; i++) {
        const double* e = edge_rec + 10 * i;
        Eigen::Vector3d currentPt(e[0], e[1], e[2]), lastPtJ(e[3], e[4], e[5]), lastPtL(e[6], e[7], e[8]);
        const float intensity = (float)e[9];
#if defined(REF_FLAVOUR_ROT)
        std::unique_ptr<ceres::CostFunction> f(LidarEdgeFactor::Create(currentPt, lastPtJ, lastPtL, q_lb, t_lb, intensity * 200 / n_edge));
#else
        std::unique_ptr<ceres::CostFunction> f(LidarEdgeFactor::Create(currentPt, lastPtJ, lastPtL, q_lb, t_lb, intensity));
#endif
        double* jac[2] = {edge_rows + 8 * i + 1, edge_rows + 8 * i + 4};
        f->Evaluate(params, edge_rows + 8 * i, jac);
    }
    for (int i = 0; i < n_surf; i++) {
        const double* s = surf_rec + 8 * i;
        Eigen::Vector3d currentPt(s[0], s[1], s[2]), norm(s[3], s[4], s[5]);
        const double normInverse = s[6], score = s[7];
#if defined(REF_FLAVOUR_ROT)
        std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormFactor::Create(currentPt, norm, q_lb, t_lb, normInverse, score * 1000 / n_surf));
#else
        std::unique_ptr<ceres::CostFunction> f(LidarPlaneNormFactor::Create(currentPt, norm, q_lb, t_lb, normInverse, score));
#endif
        double* jac[2] = {surf_rows + 8 * i + 1, surf_rows + 8 * i + 4};
        f->Evaluate(params, surf_rows + 8 * i, jac);
    }
}

}  // extern "C"
