// Plain C++ host program on the C ABI (include/lili_hip.h), no Python and no PyTorch in the process: the calls a
// maintainer places in BackendFusion / LidarOdometry, here fed from a small binary file.
//
//   s2m_demo <input.bin> [n_iters]
//   s2m_demo <input.bin> --window K [reps]     the blocking Ceres seam of a K-keyframe window timed from C++ (no ctypes in the way): the same scan in K
//                                              slots, lili_s2m_associate_window once, then `reps` evaluations through lili_s2m_linearize_window (ONE call per
//                                              evaluation of the window, what lili::LidarWindowFactor::Evaluate issues) next to K lili_s2m_linearize calls
// input.bin (little endian): int64 n_map, int64 n_query, int32 variant, int32 pad, double pose[7] (t xyz, q wxyz),
//                            float map[n_map][3], float query[n_query][3]
// Prints the pose after n_iters outer Gauss-Newton iterations with 17 significant digits (one line), the GN status
// and the correspondence count of the last association.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lili_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != LILI_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? lili_last_error(ctx) : "no context"); return 2; } \
    } while (0)

static lili_s2m_params rot_params() {      // R/config/config_fr_iosb.yaml (same values as lili_om_amd.api.make_params("rot"))
    lili_s2m_params p{};
    p.variant = LILI_VARIANT_ROT; p.loss = LILI_LOSS_CAUCHY; p.loss_a = 1.0; p.lidar_const = 7.5;
    p.kd_max_radius = 1.0; p.edge_gate = 1.0; p.surf_dist_thres = 0.12; p.reflect_thres = 0.0; p.surf_weight_min = 0.3; p.edge_dist_max = 0.1;
    p.q_lb[0] = 0.7071; p.q_lb[1] = 0.0; p.q_lb[2] = 0.0; p.q_lb[3] = 0.7071;
    p.t_lb[0] = -0.18; p.t_lb[1] = 0.0; p.t_lb[2] = -0.095;
    p.scale_surf_num = 1000.0; p.scale_edge_num = 200.0;
    return p;
}
static lili_s2m_params frontend_params() { // L/src/LidarOdometry.cpp:365,389,400,507
    lili_s2m_params p{};
    p.variant = LILI_VARIANT_FRONTEND; p.loss = LILI_LOSS_HUBER; p.loss_a = 0.1; p.lidar_const = 1.0;
    p.kd_max_radius = 1.0; p.edge_gate = 1.0; p.surf_dist_thres = 0.06; p.surf_weight_min = 0.4;
    p.q_lb[0] = 1.0;
    return p;
}

int main(int argc, char** argv) {
    lili_ctx* ctx = nullptr;
    if (argc < 2) { std::fprintf(stderr, "usage: %s input.bin [n_iters]\n", argv[0]); return 1; }
    const int n_iters = argc > 2 ? std::atoi(argv[2]) : 10;
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int64_t n_map = 0, n_q = 0; int32_t variant = 0, pad = 0; double pose[7];
    if (std::fread(&n_map, 8, 1, f) != 1 || std::fread(&n_q, 8, 1, f) != 1 || std::fread(&variant, 4, 1, f) != 1 || std::fread(&pad, 4, 1, f) != 1 ||
        std::fread(pose, 8, 7, f) != 7) { std::fprintf(stderr, "short header\n"); return 1; }
    std::vector<float> map((size_t)n_map * 3), qry((size_t)n_q * 3);
    if (std::fread(map.data(), 4, map.size(), f) != map.size() || std::fread(qry.data(), 4, qry.size(), f) != qry.size()) { std::fprintf(stderr, "short payload\n"); return 1; }
    std::fclose(f);

    CHECK(lili_ctx_create(&ctx, 0, nullptr));
    lili_s2m_params P = variant == LILI_VARIANT_FRONTEND ? frontend_params() : rot_params();
    lili_cloud cm{map.data(), (size_t)n_map, 12, -1, LILI_MEM_HOST};
    lili_cloud cq{qry.data(), (size_t)n_q, 12, -1, LILI_MEM_HOST};
    CHECK(lili_map_set(ctx, LILI_KIND_SURF, &cm, P.kd_max_radius));       // kd_tree_surf_local_map->setInputCloud
    if (argc > 3 && std::strcmp(argv[2], "--window") == 0) {
        const int K = std::atoi(argv[3]), reps = argc > 4 ? std::atoi(argv[4]) : 300;
        if (K < 1 || K > LILI_MAX_SLOTS) { std::fprintf(stderr, "--window K: 1..%d\n", LILI_MAX_SLOTS); return 1; }
        int slots[LILI_MAX_SLOTS]; double ts[3 * LILI_MAX_SLOTS], qs[4 * LILI_MAX_SLOTS], Ta[3 * LILI_MAX_SLOTS], Qa[4 * LILI_MAX_SLOTS];
        for (int k = 0; k < K; k++) {
            slots[k] = k;
            CHECK(lili_s2m_set_queries(ctx, k, LILI_KIND_SURF, &cq));
            for (int i = 0; i < 3; i++) ts[3 * k + i] = pose[i] + 0.001 * k;
            for (int i = 0; i < 4; i++) qs[4 * k + i] = pose[3 + i];
            // association pose (Q2, T2) = (Q q_lb^-1, T - Q2 t_lb), L/src/BackendFusion.cpp:929-930
            const double* b = P.q_lb; const double* q = qs + 4 * k; const double* t = ts + 3 * k;
            const double n2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
            const double iw = b[0] / n2, ix = -b[1] / n2, iy = -b[2] / n2, iz = -b[3] / n2;
            double* Q2 = Qa + 4 * k; double* T2 = Ta + 3 * k;
            Q2[0] = q[0] * iw - q[1] * ix - q[2] * iy - q[3] * iz;
            Q2[1] = q[0] * ix + q[1] * iw + q[2] * iz - q[3] * iy;
            Q2[2] = q[0] * iy - q[1] * iz + q[2] * iw + q[3] * ix;
            Q2[3] = q[0] * iz + q[1] * iy - q[2] * ix + q[3] * iw;
            const double ux = Q2[1], uy = Q2[2], uz = Q2[3], w = Q2[0], vx = P.t_lb[0], vy = P.t_lb[1], vz = P.t_lb[2];
            const double cx = 2 * (uy * vz - uz * vy), cy = 2 * (uz * vx - ux * vz), cz = 2 * (ux * vy - uy * vx);
            T2[0] = t[0] - (vx + w * cx + (uy * cz - uz * cy));
            T2[1] = t[1] - (vy + w * cy + (uz * cx - ux * cz));
            T2[2] = t[2] - (vz + w * cz + (ux * cy - uy * cx));
        }
        int n_res[2 * LILI_MAX_SLOTS];
        auto us = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
        CHECK(lili_s2m_associate_window(ctx, slots, K, LILI_MASK_SURF, Ta, Qa, &P, n_res));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 50; r++) CHECK(lili_s2m_associate_window(ctx, slots, K, LILI_MASK_SURF, Ta, Qa, &P, n_res));
        const double us_assoc = us(t0) / 50;
        double gram[64 * LILI_MAX_SLOTS], cost[LILI_MAX_SLOTS], g1[64 * LILI_MAX_SLOTS], c1[LILI_MAX_SLOTS]; int cnt[2 * LILI_MAX_SLOTS], n1[2 * LILI_MAX_SLOTS];
        for (int r = 0; r < 20; r++) CHECK(lili_s2m_linearize_window(ctx, slots, K, LILI_MASK_SURF, ts, qs, &P, gram, cost, cnt));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CHECK(lili_s2m_linearize_window(ctx, slots, K, LILI_MASK_SURF, ts, qs, &P, gram, cost, cnt));
        const double us_win = us(t0) / reps;
        for (int r = 0; r < 20; r++) for (int k = 0; k < K; k++) CHECK(lili_s2m_linearize(ctx, k, LILI_MASK_SURF, ts + 3 * k, qs + 4 * k, &P, g1 + 64 * k, c1 + k, n1 + 2 * k));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) for (int k = 0; k < K; k++) CHECK(lili_s2m_linearize(ctx, k, LILI_MASK_SURF, ts + 3 * k, qs + 4 * k, &P, g1 + 64 * k, c1 + k, n1 + 2 * k));
        const double us_per_slot = us(t0) / reps;
        const bool same = std::memcmp(gram, g1, sizeof(double) * 64 * K) == 0 && std::memcmp(cost, c1, sizeof(double) * K) == 0 && std::memcmp(cnt, n1, sizeof(int) * 2 * K) == 0;
        std::printf("{\"window_slots\": %d, \"features_per_slot\": %lld, \"correspondences_slot0\": %d, \"us_per_window_evaluation\": %.2f, \"us_per_evaluation_as_K_single_calls\": %.2f, "
                    "\"us_per_window_association\": %.2f, \"window_equals_single_calls_bit_for_bit\": %s}\n", K, (long long)n_q, n_res[0], us_win, us_per_slot, us_assoc, same ? "true" : "false");
        lili_ctx_destroy(ctx);
        return same ? 0 : 3;
    }
    CHECK(lili_s2m_set_queries(ctx, 0, LILI_KIND_SURF, &cq));
    CHECK(lili_s2m_pose_set(ctx, 0, pose, pose + 3));
    CHECK(lili_s2m_iterate(ctx, 0, LILI_MASK_SURF, &P, n_iters));          // n x (findCorrespondingSurfFeatures + linearise + GN)
    double t[3], q[4]; int status = -1;
    CHECK(lili_s2m_pose_get(ctx, 0, t, q, &status));
    // one more association at the final pose, through the host-pose entry point the Ceres adapter uses
    double Q2[4], T2[3];
    {   // (Q2, T2) = (Q q_lb^-1, T - Q2 t_lb), L/src/BackendFusion.cpp:929-930 — identity extrinsic for the front-end
        const double* b = P.q_lb; double n2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
        double iw = b[0] / n2, ix = -b[1] / n2, iy = -b[2] / n2, iz = -b[3] / n2;
        Q2[0] = q[0] * iw - q[1] * ix - q[2] * iy - q[3] * iz;
        Q2[1] = q[0] * ix + q[1] * iw + q[2] * iz - q[3] * iy;
        Q2[2] = q[0] * iy - q[1] * iz + q[2] * iw + q[3] * ix;
        Q2[3] = q[0] * iz + q[1] * iy - q[2] * ix + q[3] * iw;
        const double ux = Q2[1], uy = Q2[2], uz = Q2[3], w = Q2[0], vx = P.t_lb[0], vy = P.t_lb[1], vz = P.t_lb[2];
        double cx = 2 * (uy * vz - uz * vy), cy = 2 * (uz * vx - ux * vz), cz = 2 * (ux * vy - uy * vx);
        T2[0] = t[0] - (vx + w * cx + (uy * cz - uz * cy));
        T2[1] = t[1] - (vy + w * cy + (uz * cx - ux * cz));
        T2[2] = t[2] - (vz + w * cz + (ux * cy - uy * cx));
    }
    int n_res = 0;
    CHECK(lili_s2m_associate(ctx, 0, LILI_KIND_SURF, T2, Q2, &P, &n_res));
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g status %d n_res %d\n", t[0], t[1], t[2], q[0], q[1], q[2], q[3], status, n_res);
    lili_ctx_destroy(ctx);
    return 0;
}
