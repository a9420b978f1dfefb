// ORACLE — TEST INFRASTRUCTURE ONLY.  The back-end's local-map assembly compiled from the reference text:
//     transformCloud(cloudIn, PointPoseInfo*)      L/src/BackendFusion.cpp:730-767
//     buildLocalMapWithLandMark()                  L:1387-1484
//     downSampleCloud()                            L:1486-1528
// sliced out of the file at build time (oracle/refshim/Makefile; the temporary .inc is removed after compilation) and compiled
// inside a harness class whose data members carry the reference's names and types (L:41-42,54-70,87-122,153,192-193,228).
// The driver plays the part of the node's main loop around these functions: per keyframe it sets edge_last / surf_last /
// full_cloud, calls buildLocalMapWithLandMark() + downSampleCloud(), hands the down-sampled maps back, then records the
// keyframe (pose_cloud_frame / pose_info_cloud_frame / edge_frames / surf_frames, as saveKeyFramesAndFactors L:1683-1760 does)
// and clears the maps (clearCloud L:2389-2394).
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>
#include "refshim_deps.h"
#include "utils/common.h"

class LocalMapSlice {
public:
    ros::Publisher pub_local_surfs, pub_local_edges;
    double time_new_odom = 0;
    pcl::PointCloud<PointType>::Ptr edge_last, surf_last, full_cloud;
    vector<pcl::PointCloud<PointType>::Ptr> full_clouds_ds, full_clouds;
    pcl::PointCloud<PointType>::Ptr edge_last_ds, surf_last_ds;
    vector<pcl::PointCloud<PointType>::Ptr> edge_lasts_ds, surf_lasts_ds;
    pcl::PointCloud<PointType>::Ptr edge_local_map, surf_local_map, edge_local_map_ds, surf_local_map_ds;
    pcl::PointCloud<PointXYZI>::Ptr pose_cloud_frame;
    pcl::PointCloud<PointPoseInfo>::Ptr pose_info_cloud_frame;
    vector<pcl::PointCloud<PointType>::Ptr> edge_frames, surf_frames;
    deque<pcl::PointCloud<PointType>::Ptr> recent_edge_keyframes, recent_surf_keyframes;
    int latest_frame_idx = 0;
    pcl::VoxelGrid<PointType> ds_filter_edge, ds_filter_surf, ds_filter_edge_map, ds_filter_surf_map;
    int local_map_width = 5;
    Eigen::Quaterniond q_bl;
    Eigen::Vector3d t_bl;
    string frame_id = "lili_om";

    LocalMapSlice() {
        for (auto* p : {&edge_last, &surf_last, &full_cloud, &edge_last_ds, &surf_last_ds, &edge_local_map, &surf_local_map, &edge_local_map_ds, &surf_local_map_ds})
            p->reset(new pcl::PointCloud<PointType>());
        pose_cloud_frame.reset(new pcl::PointCloud<PointXYZI>());
        pose_info_cloud_frame.reset(new pcl::PointCloud<PointPoseInfo>());
        pub_local_surfs.topic = "/lm_planes"; pub_local_edges.topic = "/lm_edges";
    }
#include "gen/localmap_L.inc"
};

namespace {
pcl::PointCloud<PointType>::Ptr cloud_of(const float* xyza, int n) {
    pcl::PointCloud<PointType>::Ptr c(new pcl::PointCloud<PointType>());
    c->points.resize(n);
    for (int i = 0; i < n; i++) { PointType p; p.x = xyza[4 * i]; p.y = xyza[4 * i + 1]; p.z = xyza[4 * i + 2]; p.curvature = xyza[4 * i + 3]; c->points[i] = p; }
    c->width = (uint32_t)n; c->height = 1;
    return c;
}
void rows_of(const pcl::PointCloud<PointType>& c, float* out) {
    for (size_t i = 0; i < c.points.size(); i++) { out[4 * i] = c.points[i].x; out[4 * i + 1] = c.points[i].y; out[4 * i + 2] = c.points[i].z; out[4 * i + 3] = c.points[i].curvature; }
}
}  // namespace

extern "C" {

void* ref_lm_create(int local_map_width, double surf_map_leaf, double edge_map_leaf, double surf_leaf, double edge_leaf, const double q_bl[4], const double t_bl[3]) {
    LocalMapSlice* s = new LocalMapSlice();
    s->local_map_width = local_map_width;
    s->ds_filter_surf_map.setLeafSize(surf_map_leaf, surf_map_leaf, surf_map_leaf); s->ds_filter_edge_map.setLeafSize(edge_map_leaf, edge_map_leaf, edge_map_leaf);
    s->ds_filter_surf.setLeafSize(surf_leaf, surf_leaf, surf_leaf); s->ds_filter_edge.setLeafSize(edge_leaf, edge_leaf, edge_leaf);
    s->q_bl = Eigen::Quaterniond(q_bl[0], q_bl[1], q_bl[2], q_bl[3]); s->t_bl = Eigen::Vector3d(t_bl[0], t_bl[1], t_bl[2]);
    return s;
}
void ref_lm_destroy(void* h) { delete (LocalMapSlice*)h; }

// One keyframe: features in the LiDAR frame (x y z curvature rows), then its body pose pose_b (qw qx qy qz | x y z) once it is known.
// Returns the sizes of the four down-sampled clouds: surf map, edge map, surf_last_ds, edge_last_ds.
void ref_lm_keyframe(void* h, const float* surf, int n_surf, const float* edge, int n_edge, int sizes[4]) {
    LocalMapSlice* s = (LocalMapSlice*)h;
    s->surf_last = cloud_of(surf, n_surf); s->edge_last = cloud_of(edge, n_edge); s->full_cloud = cloud_of(surf, 0);
    refshim::sink().clear();
    s->buildLocalMapWithLandMark();
    s->downSampleCloud();
    sizes[0] = (int)s->surf_local_map_ds->size(); sizes[1] = (int)s->edge_local_map_ds->size();
    sizes[2] = (int)s->surf_last_ds->size(); sizes[3] = (int)s->edge_last_ds->size();
}
void ref_lm_get(void* h, float* surf_map, float* edge_map, float* surf_ds, float* edge_ds) {
    LocalMapSlice* s = (LocalMapSlice*)h;
    rows_of(*s->surf_local_map_ds, surf_map); rows_of(*s->edge_local_map_ds, edge_map); rows_of(*s->surf_last_ds, surf_ds); rows_of(*s->edge_last_ds, edge_ds);
}
// what saveKeyFramesAndFactors (L:1683-1760) and clearCloud (L:2389-2394) leave behind for the next keyframe
void ref_lm_commit(void* h, const double pose_b[7]) {
    LocalMapSlice* s = (LocalMapSlice*)h;
    PointXYZI p; p.x = pose_b[4]; p.y = pose_b[5]; p.z = pose_b[6]; p.intensity = s->pose_cloud_frame->points.size();
    s->pose_cloud_frame->push_back(p);
    PointPoseInfo pi; pi.x = pose_b[4]; pi.y = pose_b[5]; pi.z = pose_b[6]; pi.qw = pose_b[0]; pi.qx = pose_b[1]; pi.qy = pose_b[2]; pi.qz = pose_b[3];
    pi.idx = s->pose_info_cloud_frame->points.size(); pi.time = 0;
    s->pose_info_cloud_frame->push_back(pi);
    pcl::PointCloud<PointType>::Ptr e(new pcl::PointCloud<PointType>()), f(new pcl::PointCloud<PointType>());
    pcl::copyPointCloud(*s->edge_last_ds, *e); pcl::copyPointCloud(*s->surf_last_ds, *f);
    s->edge_frames.push_back(e); s->surf_frames.push_back(f);
    s->edge_local_map->clear(); s->edge_local_map_ds->clear(); s->surf_local_map->clear(); s->surf_local_map_ds->clear();
}

}  // extern "C"
