// ORACLE — TEST INFRASTRUCTURE ONLY.  REAL_DEPS twin of include/refshim_deps.h (SURVEY §8c mitigation 3, -DLILI_WITH_REFERENCE_DEPS):
// the reference's sources and the drivers see the REAL Eigen, PCL (point types, KdTreeFLANN, VoxelGrid, removeNaNFromPointCloud) and Ceres
// of the machine; only the ROS plumbing (NodeHandle, publishers, messages, params — no arithmetic) stays the in-process stand-in of
// refshim/ros_min.h, and fromROSMsg / toROSMsg are the memcpy they are for PCL's own point layouts.  NOT buildable in the graft image
// (none of the three libraries is installed): written for a maintainer's ROS machine; see README.md "Closing the third-party pin".
#pragma once
#include <Eigen/Dense>
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/common/common.h>
#include <pcl/filters/filter.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/kdtree/kdtree_flann.h>
#include <ceres/ceres.h>
#include "../include/refshim/ros_min.h"
namespace pcl {
template <class P> void fromROSMsg(const sensor_msgs::PointCloud2& m, pcl::PointCloud<P>& c) {
    const size_t n = m.data.size() / sizeof(P);
    c.points.resize(n);
    if (n) std::memcpy(reinterpret_cast<void*>(c.points.data()), m.data.data(), n * sizeof(P));
    c.width = (uint32_t)n; c.height = 1; c.is_dense = false;
}
template <class P> void toROSMsg(const pcl::PointCloud<P>& c, sensor_msgs::PointCloud2& m) {
    m.point_step = (uint32_t)sizeof(P);
    m.data.resize(c.points.size() * sizeof(P));
    if (!c.points.empty()) std::memcpy(m.data.data(), reinterpret_cast<const void*>(c.points.data()), m.data.size());
    m.width = (uint32_t)c.points.size(); m.height = 1;
}
}  // namespace pcl
