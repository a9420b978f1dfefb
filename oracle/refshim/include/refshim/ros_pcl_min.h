// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md).
//
// Stand-ins for the ROS / PCL types named by the reference's Preprocessing.cpp (both flavours), so that
// the reference's node classes compile UNMODIFIED and can be driven in-process:
//   * ros::NodeHandle::subscribe keeps the member-function callbacks; the driver calls them directly,
//   * ros::Publisher::publish appends the message to a process-wide sink (topic, stamp, bytes),
//   * ros::param::* read a process-wide string->double/int/string map filled by the driver,
//   * pcl::PointXYZI / PointXYZINormal have PCL's memory layout (32 B / 48 B) and PCL's
//     zero-initialising constructors; fromROSMsg / toROSMsg are the memcpy they are for matching layouts,
//   * pcl::removeNaNFromPointCloud follows PCL's filter.hpp (drop points with a non-finite x, y or z,
//     order preserved), pcl::VoxelGrid<PointXYZI>::filter delegates to the oracle's restatement of PCL's
//     applyFilter in its literal (std::sort) mode.
// None of this is the reference's arithmetic; it is the scaffolding that lets the reference's own run.
#pragma once
#include "eigen_min.h"
#include "ros_min.h"
#include "pcl_min.h"
