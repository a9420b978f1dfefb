// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's LiLi-OM/src/FormatConvert.cpp UNMODIFIED (livoxLidarHandler:
// livox_ros_driver/CustomMsg -> pcl::PointXYZINormal cloud) into oracle/_ref/libref_format.so.
#include "livox_ros_driver/CustomMsg.h"
#define main ref_format_node_main
#include "src/FormatConvert.cpp"
#undef main

extern "C" {
// points: n records of 19 bytes as serialised on the wire (offset_time u32 LE, x, y, z f32, reflectivity, tag, line u8).
// out: n x 12 floats (the 48-byte PointXYZINormal rows the node publishes).  Returns the number of published points.
int ref_format_convert(const unsigned char* points, int n, double stamp, float* out) {
    // the node's main() (renamed, never called: it falls off its end without a return, which is only legal for a real main)
    // advertises this topic; the driver does the same by hand
    pub_ros_points.topic = "/livox_ros_points";
    auto m = std::make_shared<livox_ros_driver::CustomMsg>();
    m->header.stamp.t = stamp;
    m->point_num = (uint32_t)n;
    m->points.resize(n);
    for (int i = 0; i < n; i++) {
        const unsigned char* p = points + (size_t)19 * i;
        livox_ros_driver::CustomPoint& c = m->points[i];
        std::memcpy(&c.offset_time, p, 4); std::memcpy(&c.x, p + 4, 4); std::memcpy(&c.y, p + 8, 4); std::memcpy(&c.z, p + 12, 4);
        c.reflectivity = p[16]; c.tag = p[17]; c.line = p[18];
    }
    refshim::sink().clear();
    livoxLidarHandler(m);
    const refshim::PubMsg& o = refshim::sink().back();
    if (!o.data.empty()) std::memcpy(out, o.data.data(), o.data.size());
    return o.point_step ? (int)(o.data.size() / o.point_step) : 0;
}
}
