// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md): the PCL half of the stand-ins (see ros_min.h for the header comment).
#pragma once
#include "eigen_min.h"
#include "ros_min.h"
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
