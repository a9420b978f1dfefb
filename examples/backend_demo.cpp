// Plain C++ host program on the C ABI (include/lili_hip.h): the keyframe loop of LiLi-OM's back end up to ceres::Solve (L/src/BackendFusion.cpp:830-980 with
// buildLocalMapWithLandMark :1387-1484 and downSampleCloud :1486-1528) as ONE lili_backend_keyframe_prepare call per keyframe, next to the same work as the calls one
// by one through host buffers (lili_localmap_push x 2, lili_localmap_commit x 2, lili_voxel_filter x 2, lili_s2m_set_queries x 2, lili_s2m_associate_window) — what a
// maintainer's BackendFusion::run would issue.  No Python and no PyTorch in the process.
//
//   backend_demo [n_keyframes = 60] [n_surf = 2500] [n_edge = 250] [width = 40] [reps = 3]
//
// Keyframes of a synthetic hall (floor, ceiling, four walls, twelve pillars whose corners give the edge features) seen from a sensor that drives a slow circuit;
// feature counts are the reference's own per keyframe (1-3 k surf + 0.1-1 k edge, L:1601-1681).  Window of 3 keyframes (slide_window_width), ROT back-end flavour.
// Both paths run on their own context over the same keyframes; the program compares counts and the Gram record of one window evaluation after every keyframe
// (exit 3 on any difference) and prints ONE JSON line: mean milliseconds per keyframe of either path and the synchronisation-free part by the call's own clock.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "lili_hip.h"

#define CHECK(ctx, call)                                                                      \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != LILI_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, lili_last_error(ctx)); return 2; } \
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

static lili_s2m_params rot_params() {      // R/config/config_fr_iosb.yaml
    lili_s2m_params p{};
    p.variant = LILI_VARIANT_ROT; p.loss = LILI_LOSS_CAUCHY; p.loss_a = 1.0; p.lidar_const = 7.5;
    p.kd_max_radius = 1.0; p.edge_gate = 1.0; p.surf_dist_thres = 0.12; p.reflect_thres = 0.0; p.surf_weight_min = 0.3; p.edge_dist_max = 0.1;
    p.q_lb[0] = 0.7071; p.q_lb[1] = 0; p.q_lb[2] = 0; p.q_lb[3] = 0.7071;
    p.t_lb[0] = -0.18; p.t_lb[1] = 0; p.t_lb[2] = -0.095;
    p.scale_surf_num = 1000.0; p.scale_edge_num = 200.0;
    return p;
}

struct Keyframe { std::vector<float> surf, edge; double t_body[3]; Quat q_body; double t_lidar[3]; Quat q_lidar; };

// the hall in the map frame: |x| <= 20, |y| <= 14, 0 <= z <= 6; pillars of 0.6 m at a 3 x 4 lattice
static void make_keyframe(int k, int n_surf, int n_edge, const lili_s2m_params& P, Keyframe& kf) {
    std::mt19937_64 rng(1000 + k);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 0.01);
    const double a = 0.05 * k, yaw = a + M_PI / 2;
    const double tl[3] = {8.0 * std::cos(a), 5.0 * std::sin(a), 1.6};
    const Quat ql{std::cos(yaw / 2), 0, 0, std::sin(yaw / 2)};
    kf.q_lidar = ql; std::memcpy(kf.t_lidar, tl, sizeof(tl));
    // body pose from the LiDAR pose: q_b = q_l * q_lb, t_b = t_l + q_l * t_lb (q_lb is the not-quite-unit quaternion of the reference's config: taken as is)
    const Quat qlb{P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]};
    kf.q_body = qmul(ql, qlb);
    double r[3]; qrot(ql, P.t_lb, r);
    for (int i = 0; i < 3; i++) kf.t_body[i] = tl[i] + r[i];
    const Quat qi = qinv(ql);
    auto to_sensor = [&](const double w[3], std::vector<float>& out, float aux) {
        const double d[3] = {w[0] - tl[0] + N(rng), w[1] - tl[1] + N(rng), w[2] - tl[2] + N(rng)};
        double s[3]; qrot(qi, d, s);
        if (s[0] * s[0] + s[1] * s[1] + s[2] * s[2] > 30.0 * 30.0) return false;
        out.push_back((float)s[0]); out.push_back((float)s[1]); out.push_back((float)s[2]); out.push_back(aux);
        return true;
    };
    kf.surf.clear(); kf.edge.clear();
    while ((int)kf.surf.size() / 4 < n_surf) {
        const int face = (int)(U(rng) * 6);
        double w[3];
        if (face < 2) { w[0] = -20 + 40 * U(rng); w[1] = -14 + 28 * U(rng); w[2] = face ? 6.0 : 0.0; }
        else if (face < 4) { w[0] = face == 2 ? -20.0 : 20.0; w[1] = -14 + 28 * U(rng); w[2] = 6 * U(rng); }
        else { w[0] = -20 + 40 * U(rng); w[1] = face == 4 ? -14.0 : 14.0; w[2] = 6 * U(rng); }
        to_sensor(w, kf.surf, (float)(k % 64) + 0.1f * (float)U(rng));
    }
    while ((int)kf.edge.size() / 4 < n_edge) {
        const int px = (int)(U(rng) * 4), py = (int)(U(rng) * 3), c = (int)(U(rng) * 4);
        double w[3] = {-15.0 + 10.0 * px + ((c & 1) ? 0.3 : -0.3), -9.0 + 9.0 * py + ((c & 2) ? 0.3 : -0.3), 6 * U(rng)};
        to_sensor(w, kf.edge, (float)(k % 64) + 0.1f * (float)U(rng));
    }
}

int main(int argc, char** argv) {
    const int n_kf = argc > 1 ? std::atoi(argv[1]) : 60, n_surf = argc > 2 ? std::atoi(argv[2]) : 2500, n_edge = argc > 3 ? std::atoi(argv[3]) : 250;
    const int width = argc > 4 ? std::atoi(argv[4]) : 40, reps = argc > 5 ? std::atoi(argv[5]) : 3;
    const int K = 3, mask = LILI_MASK_SURF | LILI_MASK_EDGE;
    const lili_s2m_params P = rot_params();
    std::vector<Keyframe> kfs(n_kf);
    for (int k = 0; k < n_kf; k++) make_keyframe(k, n_surf, n_edge, P, kfs[k]);
    lili_ctx* a = nullptr; lili_ctx* b = nullptr;
    if (lili_ctx_create(&a, 0, nullptr) != LILI_OK || lili_ctx_create(&b, 0, nullptr) != LILI_OK) { std::fprintf(stderr, "no gfx950 device\n"); return 2; }
    lili_backend_options opt{};
    opt.leaf_surf = opt.leaf_surf_map = 0.4f; opt.leaf_edge = opt.leaf_edge_map = 0.2f; opt.width = width; opt.want_timing = 1; opt.join_slot = -1;
    const size_t cap = (size_t)std::max(n_surf, n_edge);
    std::vector<float> ds_prev[2], ds_now[2] = {std::vector<float>(cap * 4), std::vector<float>(cap * 4)};
    double ms_fused = 0, ms_staged = 0, us_inside = 0;
    long long total_corr = 0;
    int n_timed = 0;
    bool same = true;
    for (int rep = 0; rep < reps; rep++) {
        CHECK(a, lili_localmap_reset(a, LILI_KIND_SURF)); CHECK(a, lili_localmap_reset(a, LILI_KIND_EDGE));
        CHECK(b, lili_localmap_reset(b, LILI_KIND_SURF)); CHECK(b, lili_localmap_reset(b, LILI_KIND_EDGE));
        for (int k = 0; k < n_kf; k++) {
            const Keyframe& kf = kfs[k];
            int slots[K]; double ta[3 * K], qa[4 * K], tb_[3 * K], qb_[4 * K];
            const int first = std::max(0, k - K + 1), n = k - first + 1;
            for (int i = 0; i < n; i++) {
                const Keyframe& w = kfs[first + i];
                slots[i] = (first + i) % K;
                // association pose (L:929-930): Q2 = q * q_lb^-1, T2 = t - Q2 t_lb; here the LiDAR pose the keyframe was generated at
                ta[3 * i] = w.t_lidar[0]; ta[3 * i + 1] = w.t_lidar[1]; ta[3 * i + 2] = w.t_lidar[2];
                qa[4 * i] = w.q_lidar.w; qa[4 * i + 1] = w.q_lidar.x; qa[4 * i + 2] = w.q_lidar.y; qa[4 * i + 3] = w.q_lidar.z;
                for (int c = 0; c < 3; c++) tb_[3 * i + c] = w.t_body[c];
                qb_[4 * i] = w.q_body.w; qb_[4 * i + 1] = w.q_body.x; qb_[4 * i + 2] = w.q_body.y; qb_[4 * i + 3] = w.q_body.z;
            }
            const lili_cloud ns{kf.surf.data(), (size_t)n_surf, 16, 12, LILI_MEM_HOST}, ne{kf.edge.data(), (size_t)n_edge, 16, 12, LILI_MEM_HOST};
            const double* tj = k ? kfs[k - 1].t_lidar : nullptr;
            const double qj[4] = {k ? kfs[k - 1].q_lidar.w : 1, k ? kfs[k - 1].q_lidar.x : 0, k ? kfs[k - 1].q_lidar.y : 0, k ? kfs[k - 1].q_lidar.z : 0};
            // ---- ONE call: the previous keyframe joins the maps from its slot on the device
            int32_t nf[2 * K] = {0}; lili_backend_result res{};
            opt.join_slot = k ? (k - 1) % K : -1;
            auto t0 = std::chrono::steady_clock::now();
            CHECK(a, lili_backend_keyframe_prepare(a, nullptr, nullptr, tj, qj, &ns, &ne, slots, n, ta, qa, &P, &opt, nf, &res));
            auto t1 = std::chrono::steady_clock::now();
            // ---- the calls one by one, clouds through host buffers
            int32_t nsg[2 * K] = {0};
            if (k) {
                const lili_cloud js{ds_prev[0].data(), ds_prev[0].size() / 4, 16, 12, LILI_MEM_HOST}, je{ds_prev[1].data(), ds_prev[1].size() / 4, 16, 12, LILI_MEM_HOST};
                CHECK(b, lili_localmap_push(b, LILI_KIND_SURF, &js, tj, qj, width)); CHECK(b, lili_localmap_push(b, LILI_KIND_EDGE, &je, tj, qj, width));
                CHECK(b, lili_localmap_commit(b, LILI_KIND_SURF, opt.leaf_surf_map, P.kd_max_radius, nullptr, nullptr));
                CHECK(b, lili_localmap_commit(b, LILI_KIND_EDGE, opt.leaf_edge_map, P.edge_gate, nullptr, nullptr));
            }
            for (int kind = 0; kind < 2; kind++) {
                lili_feature_out o{ds_now[kind].data(), cap, 16, LILI_MEM_HOST, 0};
                CHECK(b, lili_voxel_filter(b, kind ? &ne : &ns, kind ? opt.leaf_edge : opt.leaf_surf, &o, nullptr));
                const lili_cloud qc{ds_now[kind].data(), o.count, 16, 12, LILI_MEM_HOST};
                CHECK(b, lili_s2m_set_queries(b, slots[n - 1], kind, &qc));
                ds_prev[kind].assign(ds_now[kind].begin(), ds_now[kind].begin() + o.count * 4);
            }
            if (k) CHECK(b, lili_s2m_associate_window(b, slots, n, mask, ta, qa, &P, nsg));
            auto t2 = std::chrono::steady_clock::now();
            if (rep > 0 || reps == 1) {
                ms_fused += std::chrono::duration<double, std::milli>(t1 - t0).count(); ms_staged += std::chrono::duration<double, std::milli>(t2 - t1).count();
                us_inside += res.stage_us[2]; n_timed++;
            }
            if (k) {
                for (int i = 0; i < 2 * n; i++) { same = same && nf[i] == nsg[i]; total_corr += rep == 0 ? nf[i] : 0; }
                // one evaluation of the window at the body poses: the Gram records must agree bit for bit
                double ga[64 * K], gb[64 * K], ca[K], cb[K];
                CHECK(a, lili_s2m_linearize_window(a, slots, n, mask, tb_, qb_, &P, ga, ca, nullptr));
                CHECK(b, lili_s2m_linearize_window(b, slots, n, mask, tb_, qb_, &P, gb, cb, nullptr));
                same = same && std::memcmp(ga, gb, sizeof(double) * 64 * n) == 0 && std::memcmp(ca, cb, sizeof(double) * n) == 0;
            }
        }
    }
    std::printf("{\"keyframes\": %d, \"surf_features\": %d, \"edge_features\": %d, \"local_map_width\": %d, \"window\": %d, \"correspondences_total\": %lld, "
                "\"ms_per_keyframe_one_call\": %.4f, \"ms_per_keyframe_separate_calls\": %.4f, \"ms_inside_the_call_by_its_own_clock\": %.4f, "
                "\"one_call_equals_separate_calls_bit_for_bit\": %s}\n",
                n_kf, n_surf, n_edge, width, K, total_corr, ms_fused / std::max(n_timed, 1), ms_staged / std::max(n_timed, 1), us_inside / std::max(n_timed, 1) * 1e-3,
                same ? "true" : "false");
    lili_ctx_destroy(a); lili_ctx_destroy(b);
    return same ? 0 : 3;
}
