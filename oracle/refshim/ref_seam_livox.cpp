// ORACLE — TEST INFRASTRUCTURE ONLY.  The ROS-node seam for the Livox chain, in C++: the reference's two nodes
//     FormatConvert (livox_ros_driver/CustomMsg -> /livox_ros_points)  ->  Preprocessing (/livox_ros_points + /livox/imu -> features)
// both compiled unmodified (namespaces refconv / refnode), next to ONE node that takes the CustomMsg and the IMU stream and
// calls the product's C ABI: lili_livox_custom_to_cloud into DEVICE memory, lili_imu_integrate, lili_extract_livox on that device
// cloud (the merged-nodelet variant of INTEGRATION.md §4).  Same messages in; every cloud the reference chain publishes must
// come out of the GPU node on the same topic with the same stamp and points.  Built into oracle/_ref/seam_livox_check.
//
// Input: int32 n_scans, n_imu | per scan: double stamp, int32 n, n x 19-byte CustomPoint | double imu_stamp[n_imu] | double gyr[n_imu*3].
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "livox_ros_driver/CustomMsg.h"
#include "utils/common.h"
#include "utils/timer.h"
#include "utils/math_tools.h"
#include "lili_hip.h"

namespace refconv {
#define main ref_format_node_main
#include "src/FormatConvert.cpp"
#undef main
}  // namespace refconv
#undef PI
namespace refnode {
#define main ref_livox_node_main
#include "src/Preprocessing.cpp"
#undef main
}  // namespace refnode

class GpuLivoxNode {
public:
    GpuLivoxNode() {
        pub_surf = nh.advertise<sensor_msgs::PointCloud2>("/surf_features", 100);
        pub_edge = nh.advertise<sensor_msgs::PointCloud2>("/edge_features", 100);
        pub_cutted_cloud = nh.advertise<sensor_msgs::PointCloud2>("/lidar_cloud_cutted", 100);
        lp.surf_thres = 0.28; lp.edge_thres = 4.0; lp.near_range = 0.1f;
        lili_imu_reset(&imu_state);
        ok = lili_ctx_create(&gpu, 0, nullptr) == LILI_OK;
    }
    ~GpuLivoxNode() { for (auto& s : cloud_queue) (void)hipFree(s.dev); if (gpu) lili_ctx_destroy(gpu); }
    void imuHandler(const sensor_msgs::ImuConstPtr& m) {
        stamps.push_back(m->header.stamp.toSec());
        gyr.push_back(m->angular_velocity.x); gyr.push_back(m->angular_velocity.y); gyr.push_back(m->angular_velocity.z);
    }
    void livoxHandler(const livox_ros_driver::CustomMsgConstPtr& msg) {
        // FormatConvert's job, on the device: 19-byte wire points of the deserialised struct (same field offsets) -> 48-byte cloud in HBM
        Scan s; s.stamp = msg->header.stamp.toSec(); s.n = msg->point_num; s.dev = nullptr;
        std::vector<unsigned char> wire((size_t)s.n * 19);
        for (size_t i = 0; i < s.n; i++) {
            const livox_ros_driver::CustomPoint& c = msg->points[i]; unsigned char* p = wire.data() + 19 * i;
            std::memcpy(p, &c.offset_time, 4); std::memcpy(p + 4, &c.x, 4); std::memcpy(p + 8, &c.y, 4); std::memcpy(p + 12, &c.z, 4);
            p[16] = c.reflectivity; p[17] = c.tag; p[18] = c.line;
        }
        if (hipMalloc(&s.dev, std::max<size_t>(s.n, 1) * 48) != hipSuccess) return;
        if (lili_livox_custom_to_cloud(gpu, wire.data(), s.n, 19, LILI_MEM_HOST, s.dev, LILI_MEM_DEVICE) != LILI_OK) { std::printf("error=%s\n", lili_last_error(gpu)); return; }
        cloud_queue.push_back(s);
        if (cloud_queue.size() <= 2) return;                          // the Preprocessing node's two-scan delay (L/src/Preprocessing.cpp:196-207)
        Scan cur = cloud_queue.front();
        cloud_queue.pop_front();
        const double time_scan_next = cloud_queue.front().stamp;
        const size_t last = imu_state.idx > 0 ? (size_t)imu_state.idx - 1 : 0;
        if (stamps.empty() || stamps[last] > time_scan_next) { (void)hipFree(cur.dev); return; }
        double q_imu[4];
        (void)lili_imu_integrate(&imu_state, stamps.data(), gyr.data(), stamps.size(), time_scan_next, q_imu);
        lili_cloud scan{cur.dev, cur.n, 48, 32, LILI_MEM_DEVICE};
        pcl::PointCloud<pcl::PointXYZINormal> cut, edge, surf;
        cut.points.resize(cur.n); edge.points.resize(24000); surf.points.resize(24000);
        lili_feature_out co{cut.points.data(), cur.n, 48, LILI_MEM_HOST, 0}, eo{edge.points.data(), 24000, 48, LILI_MEM_HOST, 0}, so{surf.points.data(), 24000, 48, LILI_MEM_HOST, 0};
        if (lili_extract_livox(gpu, &scan, 36, q_imu, &lp, &co, &eo, &so) != LILI_OK) { std::printf("error=%s\n", lili_last_error(gpu)); (void)hipFree(cur.dev); return; }
        (void)hipFree(cur.dev);
        cut.points.resize(co.count); edge.points.resize(eo.count); surf.points.resize(so.count);
        auto publish = [&](ros::Publisher& pub, const pcl::PointCloud<pcl::PointXYZINormal>& c) {
            sensor_msgs::PointCloud2 m; pcl::toROSMsg(c, m);
            m.header.stamp.t = cur.stamp; m.header.frame_id = "lili_om";
            pub.publish(m);
        };
        publish(pub_surf, surf); publish(pub_edge, edge); publish(pub_cutted_cloud, cut);   // the reference's order (L:385-401)
    }
    bool ok = false;
private:
    struct Scan { double stamp; size_t n; void* dev; };
    ros::NodeHandle nh;
    ros::Publisher pub_surf, pub_edge, pub_cutted_cloud;
    std::deque<Scan> cloud_queue;
    std::vector<double> stamps, gyr;
    lili_imu_state imu_state;
    lili_livox_params lp;
    lili_ctx* gpu = nullptr;
};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (std::fread(hdr, sizeof(int), 2, f) != 2) return 2;
    struct Ev { double t; int kind; int idx; };
    std::vector<Ev> ev;
    std::vector<std::shared_ptr<livox_ros_driver::CustomMsg>> msgs;
    for (int s = 0; s < hdr[0]; s++) {
        double stamp; int n;
        if (std::fread(&stamp, sizeof(double), 1, f) != 1 || std::fread(&n, sizeof(int), 1, f) != 1) return 2;
        std::vector<unsigned char> raw((size_t)n * 19);
        if (n && std::fread(raw.data(), 19, n, f) != (size_t)n) return 2;
        auto m = std::make_shared<livox_ros_driver::CustomMsg>();
        m->header.stamp.t = stamp; m->point_num = (uint32_t)n; m->points.resize(n);
        for (int i = 0; i < n; i++) {
            const unsigned char* p = raw.data() + 19 * (size_t)i; livox_ros_driver::CustomPoint& c = m->points[i];
            std::memcpy(&c.offset_time, p, 4); std::memcpy(&c.x, p + 4, 4); std::memcpy(&c.y, p + 8, 4); std::memcpy(&c.z, p + 12, 4);
            c.reflectivity = p[16]; c.tag = p[17]; c.line = p[18];
        }
        msgs.push_back(m);
        ev.push_back(Ev{stamp, 1, s});
    }
    std::vector<double> it(hdr[1]), ig((size_t)hdr[1] * 3);
    if (hdr[1] && (std::fread(it.data(), sizeof(double), hdr[1], f) != (size_t)hdr[1] || std::fread(ig.data(), sizeof(double) * 3, hdr[1], f) != (size_t)hdr[1])) return 2;
    std::fclose(f);
    for (int i = 0; i < hdr[1]; i++) ev.push_back(Ev{it[i], 0, i});
    std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.t < b.t || (a.t == b.t && a.kind < b.kind); });
    auto& P = refshim::params();
    P["/preprocessing/surf_thres"] = refshim::ParamVal{0, 0.28, ""}; P["/preprocessing/edge_thres"] = refshim::ParamVal{0, 4.0, ""};
    P["/common/frame_id"] = refshim::ParamVal{1, 0, "lili_om"};
    auto imu_msg = [&](int i) { auto m = std::make_shared<sensor_msgs::Imu>(); m->header.stamp.t = it[i]; m->angular_velocity.x = ig[3 * i]; m->angular_velocity.y = ig[3 * i + 1]; m->angular_velocity.z = ig[3 * i + 2]; return m; };

    // ---- the reference chain: FormatConvert's handler publishes /livox_ros_points, which the Preprocessing node receives
    refconv::pub_ros_points.topic = "/livox_ros_points";
    std::unique_ptr<refnode::Preprocessing> ref(new refnode::Preprocessing());
    std::vector<refshim::PubMsg> mr;
    for (const Ev& e : ev) {
        if (e.kind == 0) { ref->imuHandler(imu_msg(e.idx)); continue; }
        refshim::sink().clear();
        refconv::livoxLidarHandler(msgs[e.idx]);
        auto pc = std::make_shared<sensor_msgs::PointCloud2>();
        const refshim::PubMsg& o = refshim::sink().back();
        pc->header.stamp.t = o.stamp; pc->point_step = o.point_step; pc->data = o.data; pc->width = (uint32_t)(o.data.size() / 48);
        refshim::sink().clear();
        ref->cloudHandler(pc);
        for (const auto& m : refshim::sink()) mr.push_back(m);
    }
    // ---- the GPU node
    GpuLivoxNode gpu;
    if (!gpu.ok) { std::printf("error=no_device\n"); return 3; }
    refshim::sink().clear();
    for (const Ev& e : ev) { if (e.kind == 0) gpu.imuHandler(imu_msg(e.idx)); else gpu.livoxHandler(msgs[e.idx]); }
    std::vector<refshim::PubMsg> mg = refshim::sink();

    std::printf("messages_reference=%zu\nmessages_gpu=%zu\n", mr.size(), mg.size());
    int same_hdr = 1, same_count = 1; size_t pts = 0, same_payload = 0; double max_normal = 0;
    for (size_t k = 0; k < std::min(mr.size(), mg.size()); k++) {
        if (mr[k].topic != mg[k].topic || mr[k].stamp != mg[k].stamp || mr[k].point_step != mg[k].point_step) same_hdr = 0;
        if (mr[k].data.size() != mg[k].data.size()) { same_count = 0; continue; }
        const size_t n = mr[k].data.size() / 48;
        for (size_t i = 0; i < n; i++) {
            const float* a = (const float*)(mr[k].data.data() + 48 * i); const float* b = (const float*)(mg[k].data.data() + 48 * i);
            bool eq = true;
            for (int c : {0, 1, 2, 8, 9}) if (std::memcmp(a + c, b + c, 4) != 0) eq = false;                    // x y z intensity curvature
            for (int c : {4, 5, 6}) max_normal = std::fmax(max_normal, std::fabs(std::fabs((double)a[c]) - std::fabs((double)b[c])));   // sign of an eigenvector is arbitrary
            pts++; same_payload += eq ? 1 : 0;
        }
    }
    std::printf("same_topics_stamps=%d\nsame_point_counts=%d\npoints=%zu\nbit_identical_payload=%zu\nmax_abs_normal_diff=%.3e\n", same_hdr, same_count, pts, same_payload, max_normal);
    return 0;
}
