// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md).
//
// ROS half of the stand-ins (split out so that oracle/refshim/Makefile REAL_DEPS=1 can pair it with the REAL Eigen / PCL / Ceres): stand-ins for the ROS / PCL types named by the reference's Preprocessing.cpp (both flavours), so that
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
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace std::placeholders;   // boost/bind.hpp (pulled in by ros.h) puts _1, _2 in the global namespace

// ---------------------------------------------------------------------------------------------- ROS
#define ROS_WARN(...)  do { if (::refshim::verbose()) { fprintf(stderr, "[ref WARN] " __VA_ARGS__); fputc('\n', stderr); } } while (0)
#define ROS_ERROR(...) do { if (::refshim::verbose()) { fprintf(stderr, "[ref ERROR] " __VA_ARGS__); fputc('\n', stderr); } } while (0)
#define ROS_INFO(...)  do { if (::refshim::verbose()) { fprintf(stderr, "[ref INFO] " __VA_ARGS__); fputc('\n', stderr); } } while (0)
#define ROS_BREAK()    do { fprintf(stderr, "[ref] ROS_BREAK\n"); abort(); } while (0)

namespace refshim {
struct PubMsg { std::string topic; double stamp; uint32_t point_step; std::vector<uint8_t> data; };
struct ParamVal { int kind; double d; std::string s; };   // kind 0 = number, 1 = string
inline bool& verbose() { static bool v = false; return v; }
inline std::vector<PubMsg>& sink() { static std::vector<PubMsg> s; return s; }
inline std::map<std::string, ParamVal>& params() { static std::map<std::string, ParamVal> p; return p; }
}  // namespace refshim

namespace ros {
struct Time {
    double t = 0;
    Time() {}
    explicit Time(double sec) : t(sec) {}
    double toSec() const { return t; }
    Time& fromSec(double sec) { t = sec; return *this; }
};
struct Subscriber {};
struct Publisher {
    std::string topic;
    template <class M> void publish(const M& m) const { refshim_publish(topic, m); }   // overloads below, found by ADL at instantiation
};
struct NodeHandle {
    explicit NodeHandle(const std::string& = "") {}
    template <class M, class C> Subscriber subscribe(const std::string&, int, void (C::*)(const std::shared_ptr<const M>&), C*) { return Subscriber(); }
    template <class M> Subscriber subscribe(const std::string&, int, void (*)(const std::shared_ptr<const M>&)) { return Subscriber(); }
    template <class M> Publisher advertise(const std::string& topic, int) { Publisher p; p.topic = topic; return p; }
};
namespace this_node { inline std::string getName() { return "refshim"; } }
namespace param {
inline bool search(const std::string& name, std::string& key) { key = name; return refshim::params().count(name) != 0; }
inline bool has(const std::string& key) { return refshim::params().count(key) != 0; }
inline bool get(const std::string& key, double& v) { auto& p = refshim::params().at(key); if (p.kind) return false; v = p.d; return true; }
inline bool get(const std::string& key, int& v) { auto& p = refshim::params().at(key); if (p.kind) return false; v = (int)p.d; return true; }
inline bool get(const std::string& key, bool& v) { auto& p = refshim::params().at(key); if (p.kind) return false; v = p.d != 0; return true; }
inline bool get(const std::string& key, std::string& v) { auto& p = refshim::params().at(key); if (!p.kind) return false; v = p.s; return true; }
}  // namespace param
inline void init(int, char**, const std::string&) {}
inline void spin() {}
inline void spinOnce() {}
inline bool ok() { return false; }
struct Rate { explicit Rate(double) {} void sleep() {} };
}  // namespace ros
namespace google { inline void InitGoogleLogging(const char*) {} }

namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
}
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
}
namespace sensor_msgs {
struct PointCloud2 {
    std_msgs::Header header;
    uint32_t height = 1, width = 0, point_step = 0, row_step = 0;
    bool is_dense = true;
    std::vector<uint8_t> data;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
struct Imu {
    std_msgs::Header header;
    geometry_msgs::Quaternion orientation;
    geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs

namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PoseWithCovariance { Pose pose; };
}
namespace nav_msgs {
struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; };
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
}
namespace sensor_msgs {
inline void refshim_publish(const std::string& topic, const PointCloud2& m) {
    refshim::sink().push_back(refshim::PubMsg{topic, m.header.stamp.toSec(), m.point_step, m.data});
}
}
namespace nav_msgs {
// pose messages land in the sink as 7 doubles (qw qx qy qz | x y z), point_step 0
inline void refshim_publish(const std::string& topic, const Odometry& m) {
    double v[7] = {m.pose.pose.orientation.w, m.pose.pose.orientation.x, m.pose.pose.orientation.y, m.pose.pose.orientation.z,
                   m.pose.pose.position.x, m.pose.pose.position.y, m.pose.pose.position.z};
    refshim::PubMsg p{topic, m.header.stamp.toSec(), 0, {}};
    p.data.assign((const uint8_t*)v, (const uint8_t*)v + sizeof(v));
    refshim::sink().push_back(p);
}
inline void refshim_publish(const std::string&, const Path&) {}   // the path repeats the odometry poses
}

