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
#include "eigen_min.h"

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

// ---------------------------------------------------------------------------------------------- PCL
#define POINT_CLOUD_REGISTER_POINT_STRUCT(...)

extern "C" int lo_voxel_grid(const float* pts, int n, float leaf, int stable, float* out, int* counts);   // oracle/lo_extract.cpp

namespace pcl {
struct alignas(16) PointXYZI {
    float x, y, z, pad0;
    float intensity, pad1[3];
    PointXYZI() : x(0), y(0), z(0), pad0(1.0f), intensity(0), pad1{0, 0, 0} {}
};
struct alignas(16) PointXYZINormal {
    float x, y, z, pad0;
    float normal_x, normal_y, normal_z, pad1;
    float intensity, curvature, pad2[2];
    PointXYZINormal() : x(0), y(0), z(0), pad0(1.0f), normal_x(0), normal_y(0), normal_z(0), pad1(0), intensity(0), curvature(0), pad2{0, 0} {}
};
static_assert(sizeof(PointXYZI) == 32 && sizeof(PointXYZINormal) == 48, "PCL layouts");

struct PCLHeader { uint32_t seq = 0; uint64_t stamp = 0; std::string frame_id; };

template <class P> struct PointCloud {
    typedef std::shared_ptr<PointCloud<P>> Ptr;
    typedef std::shared_ptr<const PointCloud<P>> ConstPtr;
    PCLHeader header;
    std::vector<P> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
    void clear() { points.clear(); width = 0; height = 0; }
    void push_back(const P& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    PointCloud& operator+=(const PointCloud& o) {
        points.insert(points.end(), o.points.begin(), o.points.end());
        width = (uint32_t)points.size(); height = 1;
        return *this;
    }
};

template <class P> void fromROSMsg(const sensor_msgs::PointCloud2& m, PointCloud<P>& c) {
    size_t n = m.point_step ? m.data.size() / m.point_step : 0;
    c.points.resize(n);
    for (size_t i = 0; i < n; i++) std::memcpy((void*)&c.points[i], m.data.data() + i * m.point_step, std::min<size_t>(sizeof(P), m.point_step));
    c.width = (uint32_t)n; c.height = 1; c.is_dense = m.is_dense;
}
template <class P> void toROSMsg(const PointCloud<P>& c, sensor_msgs::PointCloud2& m) {
    m.point_step = sizeof(P); m.width = (uint32_t)c.points.size(); m.height = 1; m.row_step = m.point_step * m.width;
    m.data.resize(c.points.size() * sizeof(P));
    if (!c.points.empty()) std::memcpy(m.data.data(), (const void*)c.points.data(), m.data.size());
}
// pcl/filters/impl/filter.hpp: keep points whose x, y, z are all finite; order preserved; index = source position.
template <class P> void removeNaNFromPointCloud(const PointCloud<P>& in, PointCloud<P>& out, std::vector<int>& index) {
    if (&in != &out) { out.header = in.header; out.points.resize(in.points.size()); }
    index.resize(in.points.size());
    size_t j = 0;
    for (size_t i = 0; i < in.points.size(); ++i) {
        if (!std::isfinite(in.points[i].x) || !std::isfinite(in.points[i].y) || !std::isfinite(in.points[i].z)) continue;
        out.points[j] = in.points[i];
        index[j] = (int)i;
        j++;
    }
    if (j != in.points.size()) { out.points.resize(j); index.resize(j); }
    out.height = 1; out.width = (uint32_t)j; out.is_dense = true;
}

template <class P> struct VoxelGrid;
template <> struct VoxelGrid<PointXYZI> {
    PointCloud<PointXYZI>::Ptr in;
    float leaf = 0;
    void setInputCloud(const PointCloud<PointXYZI>::Ptr& c) { in = c; }
    void setLeafSize(float lx, float, float) { leaf = lx; }
    void filter(PointCloud<PointXYZI>& out) {
        size_t n = in->points.size();
        std::vector<float> p(4 * n + 4), o(4 * n + 4);
        std::vector<int> cnt(n + 1);
        for (size_t i = 0; i < n; i++) { p[4 * i] = in->points[i].x; p[4 * i + 1] = in->points[i].y; p[4 * i + 2] = in->points[i].z; p[4 * i + 3] = in->points[i].intensity; }
        int m = lo_voxel_grid(p.data(), (int)n, leaf, /*stable=*/0, o.data(), cnt.data());
        out.points.resize(m);
        for (int k = 0; k < m; k++) { PointXYZI q; q.x = o[4 * k]; q.y = o[4 * k + 1]; q.z = o[4 * k + 2]; q.intensity = o[4 * k + 3]; out.points[k] = q; }
        out.width = (uint32_t)m; out.height = 1; out.is_dense = true;
    }
};
template <class A, class B> void copyPointCloud(const PointCloud<A>& in, PointCloud<B>& out) {
    out.header = in.header; out.width = in.width; out.height = in.height; out.is_dense = in.is_dense;
    out.points.resize(in.points.size());
    for (size_t i = 0; i < in.points.size(); i++) out.points[i] = in.points[i];   // same type at every call site of the reference
}

// pcl::VoxelGrid<PointXYZINormal>: x, y, z, intensity, curvature are PCL's float-accumulated centroids (the oracle's
// applyFilter restatement, one pass per extra field over identical voxel membership and order); the normal is the
// normalised sum (PCL's AccumulatorNormal).  Nothing downstream of the filter in the reference's matchers reads the
// normal or intensity of a filtered point, only x, y, z and (back-end, Livox) curvature.
template <> struct VoxelGrid<PointXYZINormal> {
    PointCloud<PointXYZINormal>::Ptr in;
    float leaf = 0;
    void setInputCloud(const PointCloud<PointXYZINormal>::Ptr& c) { in = c; }
    void setLeafSize(float lx, float, float) { leaf = lx; }
    void filter(PointCloud<PointXYZINormal>& out) {
        size_t n = in->points.size();
        std::vector<float> p(4 * n + 4), o[5];
        std::vector<int> cnt(n + 1);
        int m = 0;
        for (int f = 0; f < 5; f++) {
            o[f].resize(4 * n + 4);
            for (size_t i = 0; i < n; i++) {
                const PointXYZINormal& q = in->points[i];
                const float aux[5] = {q.intensity, q.curvature, q.normal_x, q.normal_y, q.normal_z};
                p[4 * i] = q.x; p[4 * i + 1] = q.y; p[4 * i + 2] = q.z; p[4 * i + 3] = aux[f];
            }
            m = lo_voxel_grid(p.data(), (int)n, leaf, /*stable=*/0, o[f].data(), cnt.data());
        }
        out.points.resize(m);
        for (int k = 0; k < m; k++) {
            PointXYZINormal q;
            q.x = o[0][4 * k]; q.y = o[0][4 * k + 1]; q.z = o[0][4 * k + 2];
            q.intensity = o[0][4 * k + 3]; q.curvature = o[1][4 * k + 3];
            float nx = o[2][4 * k + 3], ny = o[3][4 * k + 3], nz = o[4][4 * k + 3];
            float nn = std::sqrt(nx * nx + ny * ny + nz * nz);
            if (nn > 0) { q.normal_x = nx / nn; q.normal_y = ny / nn; q.normal_z = nz / nn; }
            out.points[k] = q;
        }
        out.width = (uint32_t)m; out.height = 1; out.is_dense = true;
    }
};

// pcl::KdTreeFLANN<P>::nearestKSearch(p, 5, idx, d2): exact 5-NN, FLANN's L2_Simple f32 distance, ascending —
// the oracle's kd-tree (ties by (d2, index); FLANN's own tie order is traversal-dependent).
extern "C" void* lo_kdtree_build(const float* xyz, int n);
extern "C" void lo_kdtree_free(void* t);
extern "C" void lo_knn5(void* tree, const float* q, int m, int* idx, float* d2, int nthreads);
}  // namespace pcl
namespace refshim {
// the cloud most recently handed to any KdTreeFLANN::setInputCloud, as x y z aux rows (aux = curvature or intensity):
// lets the driver report the exact map a reference node searched
inline std::vector<float>& last_tree_input() { static std::vector<float> v; return v; }
template <class P> float aux_of(const P& p);
template <> inline float aux_of(const pcl::PointXYZI& p) { return p.intensity; }
template <> inline float aux_of(const pcl::PointXYZINormal& p) { return p.curvature; }
}
namespace pcl {
template <class P> struct KdTreeFLANN {
    typedef std::shared_ptr<KdTreeFLANN<P>> Ptr;
    std::vector<float> xyz;
    void* tree = nullptr;
    ~KdTreeFLANN() { if (tree) lo_kdtree_free(tree); }
    void setInputCloud(const typename PointCloud<P>::Ptr& c) {
        if (tree) { lo_kdtree_free(tree); tree = nullptr; }
        size_t n = c->points.size();
        xyz.resize(3 * n);
        std::vector<float>& keep = refshim::last_tree_input();
        keep.resize(4 * n);
        for (size_t i = 0; i < n; i++) {
            xyz[3 * i] = c->points[i].x; xyz[3 * i + 1] = c->points[i].y; xyz[3 * i + 2] = c->points[i].z;
            keep[4 * i] = c->points[i].x; keep[4 * i + 1] = c->points[i].y; keep[4 * i + 2] = c->points[i].z; keep[4 * i + 3] = refshim::aux_of(c->points[i]);
        }
        tree = lo_kdtree_build(xyz.data(), (int)n);
    }
    int nearestKSearch(const P& p, int k, std::vector<int>& idx, std::vector<float>& d2) const {
        if (k != 5) abort();
        idx.resize(5); d2.resize(5);
        float q[3] = {p.x, p.y, p.z};
        lo_knn5(tree, q, 1, idx.data(), d2.data(), 1);
        return 5;
    }
};
}  // namespace pcl
