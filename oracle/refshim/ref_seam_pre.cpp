// ORACLE — TEST INFRASTRUCTURE ONLY.  The drop-in claim at the ROS-node seam in C++: the reference's
// LiLi-OM-ROT Preprocessing node (compiled unmodified, in namespace refnode) and a node with the SAME topics whose cloud
// handler is the binding of INTEGRATION.md §3 — lili_imu_integrate + lili_extract_rot through the C ABI of the product —
// receive the same sensor_msgs in the same order; every message the reference publishes must come out of the GPU node on
// the same topic with the same stamp and the same points.  Built into oracle/_ref/seam_pre_check, run by
// tests/test_reference_gpu.py (it links liblili_hip.so).
//
// Input: int32 n_scans, n_imu, line_num, ds_rate | double qlb[4] | per scan: double stamp, int32 n, float32 x y z intensity rows |
// double imu_stamp[n_imu] | double gyr[n_imu*3].   Output (stdout): key=value lines.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
#include "refshim_deps.h"
#include "utils/common.h"
#include "utils/timer.h"
#include "utils/math_tools.h"
#include "lili_hip.h"

namespace refnode {
#define main ref_rot_node_main
#include "src/Preprocessing.cpp"
#undef main
}  // namespace refnode

// ---- the maintainer's node: same subscriptions / publications, hot path behind the C ABI ---------------------------------
class GpuPreprocessing {
public:
    GpuPreprocessing(int line_num, int ds_rate, const double qlb[4]) {
        pub_surf = nh.advertise<sensor_msgs::PointCloud2>("/surf_features", 100);
        pub_edge = nh.advertise<sensor_msgs::PointCloud2>("/edge_features", 100);
        pub_cutted_cloud = nh.advertise<sensor_msgs::PointCloud2>("/lidar_cloud_cutted", 100);
        rp.n_scans = line_num; rp.ds_rate = ds_rate; rp.ds_v = 0.6f; rp.near_range = 3.0f;
        for (int k = 0; k < 4; k++) q_lb[k] = qlb[k];
        lili_imu_reset(&imu_state);
        ok = lili_ctx_create(&gpu, 0, nullptr) == LILI_OK;
    }
    ~GpuPreprocessing() { if (gpu) lili_ctx_destroy(gpu); }
    void imuHandler(const sensor_msgs::ImuConstPtr& m) {
        stamps.push_back(m->header.stamp.toSec());
        gyr.push_back(m->angular_velocity.x); gyr.push_back(m->angular_velocity.y); gyr.push_back(m->angular_velocity.z);
    }
    void cloudHandler(const sensor_msgs::PointCloud2ConstPtr& msg) {
        cloud_queue.push_back(*msg);                                  // the two-scan delay of the node (R/src/Preprocessing.cpp:250-262)
        if (cloud_queue.size() <= 2) return;
        sensor_msgs::PointCloud2 cur = cloud_queue.front();
        cloud_queue.pop_front();
        const double time_scan_next = cloud_queue.front().header.stamp.toSec();
        const size_t last = imu_state.idx > 0 ? (size_t)imu_state.idx - 1 : 0;
        if (stamps.empty() || stamps[last] > time_scan_next) return;   // "Waiting for IMU data ..."
        double q_imu[4];
        if (lili_imu_integrate(&imu_state, stamps.data(), gyr.data(), stamps.size(), time_scan_next, q_imu) != LILI_OK) return;
        pcl::PointCloud<pcl::PointXYZI> in;
        pcl::fromROSMsg(cur, in);
        const size_t n = in.points.size();
        lili_cloud scan{in.points.data(), n, sizeof(pcl::PointXYZI), (int)offsetof(pcl::PointXYZI, intensity), LILI_MEM_HOST};
        pcl::PointCloud<pcl::PointXYZI> full, edge, surf;
        full.points.resize(n); edge.points.resize(n); surf.points.resize(n);
        lili_feature_out fo{full.points.data(), n, sizeof(pcl::PointXYZI), LILI_MEM_HOST, 0}, eo{edge.points.data(), n, sizeof(pcl::PointXYZI), LILI_MEM_HOST, 0},
                         so{surf.points.data(), n, sizeof(pcl::PointXYZI), LILI_MEM_HOST, 0};
        if (lili_extract_rot(gpu, &scan, q_imu, q_lb, &rp, &fo, &eo, &so) != LILI_OK) { std::printf("error=%s\n", lili_last_error(gpu)); return; }
        auto finish = [](pcl::PointCloud<pcl::PointXYZI>& c, size_t cnt) {   // the library writes x y z intensity as one float4: move the 4th float to PCL's slot
            c.points.resize(cnt);
            for (auto& p : c.points) { p.intensity = p.pad0; p.pad0 = 1.0f; }
            c.width = (uint32_t)cnt; c.height = 1;
        };
        finish(full, fo.count); finish(edge, eo.count); finish(surf, so.count);
        auto publish = [&](ros::Publisher& pub, const pcl::PointCloud<pcl::PointXYZI>& c) {
            sensor_msgs::PointCloud2 m;
            pcl::toROSMsg(c, m);
            m.header.stamp = cur.header.stamp; m.header.frame_id = "lili_om_rot";
            pub.publish(m);
        };
        publish(pub_cutted_cloud, full); publish(pub_edge, edge); publish(pub_surf, surf);   // the reference's order (R:511-527)
    }
    bool ok = false;
private:
    ros::NodeHandle nh;
    ros::Publisher pub_surf, pub_edge, pub_cutted_cloud;
    std::deque<sensor_msgs::PointCloud2> cloud_queue;
    std::vector<double> stamps, gyr;
    lili_imu_state imu_state;
    lili_rot_params rp;
    double q_lb[4];
    lili_ctx* gpu = nullptr;
};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4]; double qlb[4];
    if (std::fread(hdr, sizeof(int), 4, f) != 4 || std::fread(qlb, sizeof(double), 4, f) != 4) return 2;
    struct Ev { double t; int kind; int idx; };
    std::vector<Ev> ev;
    std::vector<std::shared_ptr<sensor_msgs::PointCloud2>> clouds;
    for (int s = 0; s < hdr[0]; s++) {
        double stamp; int n;
        if (std::fread(&stamp, sizeof(double), 1, f) != 1 || std::fread(&n, sizeof(int), 1, f) != 1) return 2;
        std::vector<float> rows((size_t)n * 4);
        if (n && std::fread(rows.data(), sizeof(float) * 4, n, f) != (size_t)n) return 2;
        auto m = std::make_shared<sensor_msgs::PointCloud2>();
        m->header.stamp.t = stamp; m->point_step = 32; m->width = (uint32_t)n; m->row_step = 32u * (uint32_t)n;
        m->data.assign((size_t)n * 32, 0);
        for (int i = 0; i < n; i++) {
            float p[8] = {rows[4 * i], rows[4 * i + 1], rows[4 * i + 2], 1.0f, rows[4 * i + 3], 0, 0, 0};
            std::memcpy(m->data.data() + (size_t)i * 32, p, 32);
        }
        clouds.push_back(m);
        ev.push_back(Ev{stamp, 1, s});
    }
    std::vector<double> it(hdr[1]), ig((size_t)hdr[1] * 3);
    if (hdr[1] && (std::fread(it.data(), sizeof(double), hdr[1], f) != (size_t)hdr[1] || std::fread(ig.data(), sizeof(double) * 3, hdr[1], f) != (size_t)hdr[1])) return 2;
    std::fclose(f);
    for (int i = 0; i < hdr[1]; i++) ev.push_back(Ev{it[i], 0, i});
    std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.t < b.t || (a.t == b.t && a.kind < b.kind); });

    auto& P = refshim::params();
    P["/preprocessing/lidar_topic"] = refshim::ParamVal{1, 0, "/velodyne_points"};
    P["/preprocessing/line_num"] = refshim::ParamVal{0, (double)hdr[2], ""}; P["/preprocessing/ds_rate"] = refshim::ParamVal{0, (double)hdr[3], ""};
    P["/common/frame_id"] = refshim::ParamVal{1, 0, "lili_om_rot"}; P["/backend_fusion/imu_topic"] = refshim::ParamVal{1, 0, "/imu/data"};
    const char* qk[4] = {"/backend_fusion/ql2b_w", "/backend_fusion/ql2b_x", "/backend_fusion/ql2b_y", "/backend_fusion/ql2b_z"};
    for (int k = 0; k < 4; k++) P[qk[k]] = refshim::ParamVal{0, qlb[k], ""};

    auto imu_msg = [&](int i) { auto m = std::make_shared<sensor_msgs::Imu>(); m->header.stamp.t = it[i]; m->angular_velocity.x = ig[3 * i]; m->angular_velocity.y = ig[3 * i + 1]; m->angular_velocity.z = ig[3 * i + 2]; return m; };
    // ---- the reference node
    std::unique_ptr<refnode::Preprocessing> ref(new refnode::Preprocessing());
    refshim::sink().clear();
    for (const Ev& e : ev) { if (e.kind == 0) ref->imuHandler(imu_msg(e.idx)); else ref->cloudHandler(clouds[e.idx]); }
    std::vector<refshim::PubMsg> mr = refshim::sink();
    // ---- the GPU node
    GpuPreprocessing gpu(hdr[2], hdr[3], qlb);
    if (!gpu.ok) { std::printf("error=no_device\n"); return 3; }
    refshim::sink().clear();
    for (const Ev& e : ev) { if (e.kind == 0) gpu.imuHandler(imu_msg(e.idx)); else gpu.cloudHandler(clouds[e.idx]); }
    std::vector<refshim::PubMsg> mg = refshim::sink();

    std::printf("messages_reference=%zu\nmessages_gpu=%zu\n", mr.size(), mg.size());
    int same_hdr = 1; size_t pts = 0, same_pts = 0; double max_abs = 0; int same_count = 1;
    for (size_t k = 0; k < std::min(mr.size(), mg.size()); k++) {
        if (mr[k].topic != mg[k].topic || mr[k].stamp != mg[k].stamp || mr[k].point_step != mg[k].point_step) same_hdr = 0;
        if (mr[k].data.size() != mg[k].data.size()) { same_count = 0; continue; }
        const size_t n = mr[k].data.size() / 32;
        for (size_t i = 0; i < n; i++) {
            const float* a = (const float*)(mr[k].data.data() + 32 * i); const float* b = (const float*)(mg[k].data.data() + 32 * i);
            bool eq = true;
            for (int c : {0, 1, 2, 4}) { if (std::memcmp(a + c, b + c, 4) != 0) eq = false; max_abs = std::fmax(max_abs, std::fabs((double)a[c] - (double)b[c])); }
            pts++; same_pts += eq ? 1 : 0;
        }
    }
    std::printf("same_topics_stamps=%d\nsame_point_counts=%d\npoints=%zu\nbit_identical_points=%zu\nmax_abs_diff=%.3e\n", same_hdr, same_count, pts, same_pts, max_abs);
    return 0;
}
