// Plain C++ host program on the C ABI (include/lili_hip.h): BASELINE configs[0] — one spinning-LiDAR scan through the LOAM-style extractor of LiLi-OM-ROT
// (R/src/Preprocessing.cpp:248-535) and the back end's matcher against the caller's surf + edge maps (R/src/BackendFusion.cpp:1408-1560, 830-1007: one outer iteration
// per call here) — as ONE lili_frontend_frame_rot call per scan (LILI_FRAME_EXTERNAL_MAP | LILI_FRAME_EDGES, leaf_query 0), next to the chain of
// separate calls a node without the frame call would issue.  No Python and no PyTorch in the process.
//
//   rot_scan_demo <scan.bin> [reps] [n_iters]
// scan.bin (little endian): int32 n_scan, int32 n_surf_map, int32 n_edge_map, double t0[3], double q0[4] (the predicted body pose), float scan[n_scan][4] (x, y, z,
//                           intensity), float surf_map[n_surf_map][3], float edge_map[n_edge_map][3]
// Prints the pose of the one-call path and of the separate calls (17 significant digits; they must agree to the last bit), the feature counts and, with reps > 1,
// the mean time per scan of both paths.  The scan lies in page-locked memory (lili_host_alloc), as a driver's DMA buffer would; the maps are indexed once.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>
#include "lili_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != LILI_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? lili_last_error(ctx) : "no context"); return 2; } \
    } while (0)

static lili_s2m_params rot_params() {      // R/config/config_fr_iosb.yaml, R/src/BackendFusion.cpp:843,861,1443,1504
    lili_s2m_params p{};
    p.variant = LILI_VARIANT_ROT; p.loss = LILI_LOSS_CAUCHY; p.loss_a = 1.0; p.lidar_const = 7.5;
    p.kd_max_radius = 1.0; p.edge_gate = 1.0; p.surf_dist_thres = 0.12; p.reflect_thres = 0.0; p.surf_weight_min = 0.3; p.edge_dist_max = 0.1;
    p.q_lb[0] = 0.7071; p.q_lb[3] = 0.7071;
    p.t_lb[0] = -0.18; p.t_lb[2] = -0.095;
    p.scale_surf_num = 1000.0; p.scale_edge_num = 200.0;
    return p;
}

int main(int argc, char** argv) {
    lili_ctx* ctx = nullptr;
    if (argc < 2) { std::fprintf(stderr, "usage: %s scan.bin [reps] [n_iters]\n", argv[0]); return 1; }
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1, n_iters = argc > 3 ? std::atoi(argv[3]) : 1;
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int32_t n[3];
    double t0[3], q0[4];
    if (std::fread(n, 4, 3, f) != 3 || std::fread(t0, 8, 3, f) != 3 || std::fread(q0, 8, 4, f) != 4 || n[0] <= 0 || n[1] <= 0 || n[2] <= 0) { std::fprintf(stderr, "short header\n"); return 1; }
    float* scan_h = static_cast<float*>(lili_host_alloc((size_t)n[0] * 16));
    std::vector<float> surf_map((size_t)n[1] * 3), edge_map((size_t)n[2] * 3);
    if (!scan_h || std::fread(scan_h, 16, (size_t)n[0], f) != (size_t)n[0] || std::fread(surf_map.data(), 12, (size_t)n[1], f) != (size_t)n[1] ||
        std::fread(edge_map.data(), 12, (size_t)n[2], f) != (size_t)n[2]) { std::fprintf(stderr, "short file\n"); return 1; }
    std::fclose(f);
    CHECK(lili_ctx_create(&ctx, 0, nullptr));
    const lili_s2m_params P = rot_params();
    const lili_rot_params RP{64, 4, 0.6f, 3.0f};                       // R/config/config_fr_iosb.yaml:13-14, R/src/Preprocessing.cpp:14,281
    const double q_imu[4] = {1.0, 0.0, 0.0, 0.0};
    const lili_cloud cs{surf_map.data(), (size_t)n[1], 12, -1, LILI_MEM_HOST}, ce{edge_map.data(), (size_t)n[2], 12, -1, LILI_MEM_HOST};
    CHECK(lili_map_set(ctx, LILI_KIND_SURF, &cs, P.kd_max_radius));    // kd_tree_surf_local_map->setInputCloud / kd_tree_edge_local_map->setInputCloud
    CHECK(lili_map_set(ctx, LILI_KIND_EDGE, &ce, P.edge_gate));
    // the scan in HBM (what a merged driver + Preprocessing nodelet holds); the frame call and the extractor read it in place
    void* scan_d = nullptr;
    if (hipMalloc(&scan_d, (size_t)n[0] * 16) != hipSuccess || hipMemcpy(scan_d, scan_h, (size_t)n[0] * 16, hipMemcpyHostToDevice) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 2; }
    const lili_cloud scan{scan_d, (size_t)n[0], 16, 12, LILI_MEM_DEVICE};
    lili_frontend_options opt{0.0f, 0.4f, 20, n_iters, 1, 0, LILI_FRAME_EXTERNAL_MAP | LILI_FRAME_EDGES};
    lili_frontend_result r{};
    double sec_one = 0, sec_sep = 0, ts[3] = {0, 0, 0}, qs[4] = {1, 0, 0, 0};
    int status_sep = 0;
    size_t n_surf = 0, n_edge = 0;
    for (int rep = 0; rep < reps; rep++) {
        // ---- ONE call per scan
        auto tic = std::chrono::steady_clock::now();
        CHECK(lili_frontend_frame_rot(ctx, &scan, q_imu, P.q_lb, &RP, &P, &opt, t0, q0, &r));
        if (rep > 0) sec_one += std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
        // ---- the same scan through the separate calls
        tic = std::chrono::steady_clock::now();
        lili_feature_out full{nullptr, 0, 0, LILI_MEM_DEVICE, 0}, edge{nullptr, 0, 0, LILI_MEM_DEVICE, 0}, surf{nullptr, 0, 0, LILI_MEM_DEVICE, 0};
        CHECK(lili_extract_rot(ctx, &scan, q_imu, P.q_lb, &RP, &full, &edge, &surf));
        lili_cloud d_full{}, d_edge{}, d_surf{};
        CHECK(lili_extract_rot_device(ctx, &d_full, &d_edge, &d_surf));
        CHECK(lili_s2m_set_queries(ctx, 0, LILI_KIND_SURF, &d_surf));
        CHECK(lili_s2m_set_queries(ctx, 0, LILI_KIND_EDGE, &d_edge));
        CHECK(lili_s2m_pose_set(ctx, 0, t0, q0));
        CHECK(lili_s2m_iterate(ctx, 0, LILI_MASK_SURF | LILI_MASK_EDGE, &P, n_iters));
        CHECK(lili_s2m_pose_get(ctx, 0, ts, qs, &status_sep));
        if (rep > 0) sec_sep += std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
        n_surf = d_surf.n; n_edge = d_edge.n;
    }
    std::printf("one_call  pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g status %d n_edge %d n_surf %d\n", r.t[0], r.t[1], r.t[2], r.q[0], r.q[1], r.q[2], r.q[3], r.gn_status, r.n_edge, r.n_surf);
    std::printf("separate  pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g status %d n_edge %zu n_surf %zu\n", ts[0], ts[1], ts[2], qs[0], qs[1], qs[2], qs[3], status_sep, n_edge, n_surf);
    const bool same = std::memcmp(r.t, ts, sizeof(ts)) == 0 && std::memcmp(r.q, qs, sizeof(qs)) == 0;
    std::printf("poses_equal_bit_for_bit %d\n", same ? 1 : 0);
    if (reps > 1) std::printf("ms_per_scan one_call %.4f separate_calls %.4f (%d repetitions, the first untimed)\n", sec_one / (reps - 1) * 1e3, sec_sep / (reps - 1) * 1e3, reps - 1);
    (void)hipFree(scan_d);
    lili_host_free(scan_h);
    lili_ctx_destroy(ctx);
    return same ? 0 : 3;
}
