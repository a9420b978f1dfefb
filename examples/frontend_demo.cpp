// Plain C++ host program on the C ABI (include/lili_hip.h): the front-end odometry loop of LiLi-OM (L/src/LidarOdometry.cpp:652-707 behind
// L/src/Preprocessing.cpp:219-401) as ONE lili_frontend_frame call per Livox scan — what a merged Preprocessing + LidarOdometry nodelet would issue
// per /livox/lidar message.  No Python and no PyTorch in the process.
//
//   frontend_demo <frames.bin> [reps]
// frames.bin (little endian): int32 n_frames, int32 reference_startup, then per frame: int32 n_points, double t_first[3], double q_first[4] (used for frame 0
//                             only), float rows[n_points][5] = x, y, z, intensity (line + 0.1 * t), curvature (0.1 * reflectivity)
// The host side does what stays on the host in the reference: poseInitialization's constant-velocity prediction (L:415-441), computeRelative (L:443-480),
// unifyQuaternion (L:538-548).  Prints one line per frame (pose with 17 significant digits, status, sizes) and, with reps > 1, the mean time per frame of
// the repeated sequence.  Scans are read from page-locked memory (lili_host_alloc), as a driver's DMA buffer would be.
#include <chrono>
#include <cmath>
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

struct Quat { double w, x, y, z; };
static Quat qmul(const Quat& a, const Quat& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
static Quat qinv(const Quat& q) { const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2}; }
static void qrot(const Quat& q, const double v[3], double o[3]) {      // Eigen: v + w * (2 u x v) + u x (2 u x v)
    double c[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    for (double& e : c) e += e;
    const double d[3] = {q.y * c[2] - q.z * c[1], q.z * c[0] - q.x * c[2], q.x * c[1] - q.y * c[0]};
    for (int i = 0; i < 3; i++) o[i] = (v[i] + c[i] * q.w) + d[i];
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
    if (argc < 2) { std::fprintf(stderr, "usage: %s frames.bin [reps]\n", argv[0]); return 1; }
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1;
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    int32_t n_frames = 0, ref_start = 1;
    if (std::fread(&n_frames, 4, 1, f) != 1 || std::fread(&ref_start, 4, 1, f) != 1 || n_frames <= 0) { std::fprintf(stderr, "short header\n"); return 1; }
    struct Frame { int32_t n; double t0[3], q0[4]; float* rows; };
    std::vector<Frame> frames((size_t)n_frames);
    for (auto& fr : frames) {
        if (std::fread(&fr.n, 4, 1, f) != 1 || std::fread(fr.t0, 8, 3, f) != 3 || std::fread(fr.q0, 8, 4, f) != 4) { std::fprintf(stderr, "short frame header\n"); return 1; }
        fr.rows = static_cast<float*>(lili_host_alloc((size_t)fr.n * 20 + 16));
        if (!fr.rows || std::fread(fr.rows, 20, (size_t)fr.n, f) != (size_t)fr.n) { std::fprintf(stderr, "short frame\n"); return 1; }
    }
    std::fclose(f);
    CHECK(lili_ctx_create(&ctx, 0, nullptr));
    const lili_s2m_params P = frontend_params();
    const lili_livox_params LP{0.28, 4.0, 0.1f};                   // L/config/config_fr_iosb.yaml:5-6, L/src/Preprocessing.cpp:226
    const double q_imu[4] = {1.0, 0.0, 0.0, 0.0};
    double total_s = 0;
    for (int rep = 0; rep < reps; rep++) {
        CHECK(lili_frontend_reset(ctx));
        double abs_t[3] = {0, 0, 0}, rel_t[3] = {0, 0, 0}, prev_t[3] = {0, 0, 0};
        Quat abs_q{1, 0, 0, 0}, rel_q{1, 0, 0, 0}, prev_q{1, 0, 0, 0};
        const auto tic = std::chrono::steady_clock::now();
        for (int k = 0; k < n_frames; k++) {
            const Frame& fr = frames[(size_t)k];
            lili_frontend_options opt{0.4f, 0.4f, 20, 6, 0, 0, 0};
            double tp[3], qp[4];
            if (k == 0) {
                for (int i = 0; i < 3; i++) tp[i] = fr.t0[i];
                for (int i = 0; i < 4; i++) qp[i] = fr.q0[i];
                opt.n_iters = 0; opt.flags = ref_start ? LILI_FRAME_PUSH_EMPTY : 0;
            } else {                                                // poseInitialization: abs = abs o rel
                double r[3]; qrot(abs_q, rel_t, r);
                for (int i = 0; i < 3; i++) tp[i] = r[i] + abs_t[i];
                const Quat q0 = qmul(abs_q, rel_q);
                qp[0] = q0.w; qp[1] = q0.x; qp[2] = q0.y; qp[3] = q0.z;
                if (k == 1) { opt.n_iters = ref_start ? 8 : 12; opt.flags = ref_start ? LILI_FRAME_SELF_MAP : 0; }
            }
            const lili_cloud scan{fr.rows, (size_t)fr.n, 20, 12, LILI_MEM_HOST};
            lili_frontend_result res{};
            CHECK(lili_frontend_frame(ctx, &scan, 16, q_imu, &LP, &P, &opt, tp, qp, &res));
            Quat q{res.q[0], res.q[1], res.q[2], res.q[3]};
            if (q.w < 0) q = Quat{-q.w, -q.x, -q.y, -q.z};          // unifyQuaternion
            for (int i = 0; i < 3; i++) { prev_t[i] = abs_t[i]; abs_t[i] = res.t[i]; }
            prev_q = abs_q; abs_q = q;
            if (k > 0) {                                            // computeRelative
                const Quat pi = qinv(prev_q);
                rel_q = qmul(pi, abs_q);
                const double d[3] = {abs_t[0] - prev_t[0], abs_t[1] - prev_t[1], abs_t[2] - prev_t[2]};
                qrot(pi, d, rel_t);
            }
            if (rep == 0)
                std::printf("frame %d pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g status %d matched %d n_surf %d n_query %d n_map %d\n", k, abs_t[0], abs_t[1], abs_t[2],
                            abs_q.w, abs_q.x, abs_q.y, abs_q.z, res.gn_status, res.matched, res.n_surf, res.n_query, res.n_map);
        }
        if (rep > 0) total_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
    }
    if (reps > 1) std::printf("ms_per_frame %.4f (%d frames x %d repetitions, first repetition untimed)\n", total_s / (double)(reps - 1) / n_frames * 1e3, n_frames, reps - 1);
    for (auto& fr : frames) lili_host_free(fr.rows);
    lili_ctx_destroy(ctx);
    return 0;
}
