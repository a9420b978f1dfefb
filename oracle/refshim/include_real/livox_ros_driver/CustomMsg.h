// stand-in header (oracle/refshim/README.md): livox_ros_driver/CustomMsg and CustomPoint as published by livox_ros_driver v2.5.0
// (msg/CustomMsg.msg, msg/CustomPoint.msg): header, timebase, point_num, lidar_id, rsvd[3], points[]; a point is
// offset_time (uint32), x, y, z (float32), reflectivity, tag, line (uint8).
#pragma once
#include "refshim_deps.h"
namespace livox_ros_driver {
struct CustomPoint { uint32_t offset_time = 0; float x = 0, y = 0, z = 0; uint8_t reflectivity = 0, tag = 0, line = 0; };
struct CustomMsg {
    std_msgs::Header header;
    uint64_t timebase = 0;
    uint32_t point_num = 0;
    uint8_t lidar_id = 0;
    uint8_t rsvd[3] = {0, 0, 0};
    std::vector<CustomPoint> points;
};
typedef std::shared_ptr<const CustomMsg> CustomMsgConstPtr;
}  // namespace livox_ros_driver
