// ORACLE — TEST INFRASTRUCTURE ONLY.  The drop-in claim at the Ceres seam, checked with the reference's own
// factor header: inside one (stand-in) ceres::Problem,
//   ONE lili::LidarBatchFactor (include/lili_ceres_adapter.h, the binding INTEGRATION.md §1 hands to a maintainer), against
//   the THOUSANDS of AutoDiffCostFunction<LidarEdgeFactor|LidarPlaneNormFactor> + CauchyLoss(1.0) blocks the reference adds
//   for the same correspondences (L/src/BackendFusion.cpp:936-972 / R:836-866, restated in add_reference_blocks()),
// must present the solver with the same normal equations J^T J, J^T r and the same cost.  The correspondences come from the
// product (lili_s2m_associate + get_*_records on the GPU); the per-correspondence residual blocks come from the reference's
// LidarKeyframeFactor.h.  Built into oracle/_ref/seam_check by oracle/refshim/Makefile, run by tests/test_reference_gpu.py.
//
// Input file: int32 n_surf_map, n_edge_map, n_surf_q, n_edge_q, rot_flavour | lili_s2m_params bytes | float32 x4 rows of the four
// clouds (x y z aux) | double t[3], q[4] (body pose).  Output (stdout): key=value lines.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "refshim/eigen_min.h"
#include "refshim/ceres_min.h"
#include "factors/LidarKeyframeFactor.h"
#include "lili_ceres_adapter.h"

static void add_rows(const ceres::ResidualBlock& b, double H[49], double g[7], double* cost) {
    const int n = b.cost->num_residuals();
    std::vector<double> r(n), jt(n * 3), jq(n * 4);
    double* jac[2] = {jt.data(), jq.data()};
    const double* params[2] = {b.params[0], b.params[1]};
    if (!b.cost->Evaluate(params, r.data(), jac)) { std::printf("error=evaluate\n"); return; }
    double sq = 0; for (double x : r) sq += x * x;
    double rs = 1.0, sr1 = 1.0, asn = 0.0, rho0 = sq;
    if (b.loss) {          // ceres/corrector.cc (one block at a time; the lidar blocks have one residual each)
        double rho[3]; b.loss->Evaluate(sq, rho); rho0 = rho[0];
        sr1 = std::sqrt(rho[1]);
        if (sq == 0.0 || rho[2] <= 0.0) { rs = sr1; asn = 0.0; }
        else { const double D = 1.0 + 2.0 * sq * rho[2] / rho[1]; const double alpha = 1.0 - std::sqrt(D); rs = sr1 / (1 - alpha); asn = alpha / sq; }
    }
    *cost += 0.5 * rho0;
    for (int i = 0; i < n; i++) {
        double J[7];
        for (int c = 0; c < 3; c++) J[c] = jt[i * 3 + c];
        for (int c = 0; c < 4; c++) J[3 + c] = jq[i * 4 + c];
        if (b.loss) {      // J <- sqrt(rho') (J - alpha/|r|^2 r r^T J): per block r^T J is a row vector
            double rtJ[7] = {0};
            for (int k = 0; k < n; k++) { for (int c = 0; c < 3; c++) rtJ[c] += r[k] * jt[k * 3 + c]; for (int c = 0; c < 4; c++) rtJ[3 + c] += r[k] * jq[k * 4 + c]; }
            for (int c = 0; c < 7; c++) J[c] = sr1 * (J[c] - asn * r[i] * rtJ[c]);
        }
        const double ri = r[i] * rs;
        for (int a = 0; a < 7; a++) { for (int c = 0; c < 7; c++) H[a * 7 + c] += J[a] * J[c]; g[a] += J[a] * ri; }
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[5];
    lili_s2m_params P;
    if (std::fread(hdr, sizeof(int), 5, f) != 5 || std::fread(&P, sizeof(P), 1, f) != 1) return 2;
    std::vector<float> cl[4];
    for (int k = 0; k < 4; k++) { cl[k].resize((size_t)hdr[k] * 4); if (hdr[k] && std::fread(cl[k].data(), sizeof(float) * 4, hdr[k], f) != (size_t)hdr[k]) return 2; }
    double pose[7];
    if (std::fread(pose, sizeof(double), 7, f) != 7) return 2;
    std::fclose(f);
    const bool rot = hdr[4] != 0;
    double t[3] = {pose[0], pose[1], pose[2]}, q[4] = {pose[3], pose[4], pose[5], pose[6]};

    lili_ctx* ctx = nullptr;
    if (lili_ctx_create(&ctx, 0, nullptr) != LILI_OK) { std::printf("error=no_device\n"); return 3; }
    auto cloud = [&](int k, bool aux) { lili_cloud c{}; c.data = cl[k].data(); c.n = (size_t)hdr[k]; c.stride = 16; c.aux_offset = aux ? 12 : -1; c.mem = LILI_MEM_HOST; return c; };
    lili_cloud sm = cloud(0, !rot), em = cloud(1, false), sq = cloud(2, !rot), eq = cloud(3, false);
    int rc = lili_map_set(ctx, LILI_KIND_SURF, &sm, P.kd_max_radius) | lili_map_set(ctx, LILI_KIND_EDGE, &em, P.edge_gate) |
             lili_s2m_set_queries(ctx, 0, LILI_KIND_SURF, &sq) | lili_s2m_set_queries(ctx, 0, LILI_KIND_EDGE, &eq);
    // Q2 = Q * q_lb^-1, T2 = T - Q2 * t_lb                                                       (L/src/BackendFusion.cpp:929-930)
    Eigen::Quaterniond Qb(q[0], q[1], q[2], q[3]), q_lb(P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]);
    Eigen::Vector3d Tb(t[0], t[1], t[2]), t_lb(P.t_lb[0], P.t_lb[1], P.t_lb[2]);
    Eigen::Quaterniond Q2 = Qb * q_lb.inverse();
    Eigen::Vector3d T2 = Tb - Q2 * t_lb;
    const double t2[3] = {T2.x(), T2.y(), T2.z()}, q2[4] = {Q2.w(), Q2.x(), Q2.y(), Q2.z()};
    int ns = 0, ne = 0;
    rc |= lili_s2m_associate(ctx, 0, LILI_KIND_SURF, t2, q2, &P, &ns) | lili_s2m_associate(ctx, 0, LILI_KIND_EDGE, t2, q2, &P, &ne);
    if (rc != LILI_OK) { std::printf("error=%s\n", lili_last_error(ctx)); return 4; }

    // ---- problem A: the binding of INTEGRATION.md §1 — one batch factor, no loss function
    ceres::Problem pa;
    pa.AddResidualBlock(new lili::LidarBatchFactor(ctx, 0, LILI_MASK_SURF | LILI_MASK_EDGE, P), nullptr, t, q);
    double Ha[49] = {0}, ga[7] = {0}, ca = 0;
    add_rows(pa.blocks[0], Ha, ga, &ca);

    // ---- problem B: the reference's per-correspondence blocks over the same correspondences
    std::vector<float> scp(3 * (size_t)ns + 3), sn(3 * (size_t)ns + 3), sd(ns + 1), ecp(3 * (size_t)ne + 3), ea(3 * (size_t)ne + 3), eb(3 * (size_t)ne + 3), es(ne + 1);
    std::vector<double> ssc(ns + 1);
    size_t gs = 0, ge = 0;
    rc = lili_s2m_get_surf_records(ctx, 0, ns, nullptr, scp.data(), sn.data(), sd.data(), ssc.data(), &gs) |
         lili_s2m_get_edge_records(ctx, 0, ne, nullptr, ecp.data(), ea.data(), eb.data(), es.data(), &ge);
    if (rc != LILI_OK || (int)gs != ns || (int)ge != ne) { std::printf("error=records\n"); return 5; }
    ceres::Problem pb;
    ceres::LossFunction* lossFunction = new ceres::CauchyLoss(1.0);                                           // L:845
    for (int i = 0; i < ne; ++i) {                                                                            // L:938-955 / R:836-847
        Eigen::Vector3d currentPt(ecp[3 * i], ecp[3 * i + 1], ecp[3 * i + 2]), lastPtJ(ea[3 * i], ea[3 * i + 1], ea[3 * i + 2]), lastPtL(eb[3 * i], eb[3 * i + 1], eb[3 * i + 2]);
        const float intensity = es[i];
        ceres::CostFunction* costFunction = rot ? LidarEdgeFactor::Create(currentPt, lastPtJ, lastPtL, q_lb, t_lb, intensity * 200 / ne)
                                                : LidarEdgeFactor::Create(currentPt, lastPtJ, lastPtL, q_lb, t_lb, intensity);
        pb.AddResidualBlock(costFunction, lossFunction, t, q);
    }
    for (int i = 0; i < ns; ++i) {                                                                            // L:957-972 / R:849-866
        Eigen::Vector3d currentPt(scp[3 * i], scp[3 * i + 1], scp[3 * i + 2]), norm(sn[3 * i], sn[3 * i + 1], sn[3 * i + 2]);
        const double normInverse = sd[i];
        ceres::CostFunction* costFunction = rot ? LidarPlaneNormFactor::Create(currentPt, norm, q_lb, t_lb, normInverse, ssc[i] * 1000 / ns)
                                                : LidarPlaneNormFactor::Create(currentPt, norm, q_lb, t_lb, normInverse, ssc[i]);
        pb.AddResidualBlock(costFunction, lossFunction, t, q);
    }
    double Hb[49] = {0}, gb[7] = {0}, cb = 0;
    for (const ceres::ResidualBlock& b : pb.blocks) add_rows(b, Hb, gb, &cb);

    double hmax = 0, hdiff = 0, gmax = 0, gdiff = 0;
    for (int k = 0; k < 49; k++) { hmax = std::fmax(hmax, std::fabs(Hb[k])); hdiff = std::fmax(hdiff, std::fabs(Ha[k] - Hb[k])); }
    for (int k = 0; k < 7; k++) { gmax = std::fmax(gmax, std::fabs(gb[k])); gdiff = std::fmax(gdiff, std::fabs(ga[k] - gb[k])); }
    std::printf("n_surf=%d\nn_edge=%d\nblocks_reference=%zu\nblocks_batch=%zu\nH_rel_diff=%.3e\ng_rel_diff=%.3e\ncost_batch=%.17g\ncost_reference=%.17g\n",
                ns, ne, pb.blocks.size(), pa.blocks.size(), hdiff / hmax, gdiff / gmax, ca, cb);

    // ---- problem C: the window — three keyframes (the same scan in three slots, evaluated at three different poses) as ONE LidarWindowFactor
    //      against one LidarBatchFactor per keyframe: the same residuals and Jacobian entries, bit for bit
    {
        const int K = 3;
        for (int k = 1; k < K; k++) {
            rc = lili_s2m_set_queries(ctx, k, LILI_KIND_SURF, &sq) | lili_s2m_set_queries(ctx, k, LILI_KIND_EDGE, &eq);
            int a = 0, b2 = 0;
            rc |= lili_s2m_associate(ctx, k, LILI_KIND_SURF, t2, q2, &P, &a) | lili_s2m_associate(ctx, k, LILI_KIND_EDGE, t2, q2, &P, &b2);
            if (rc != LILI_OK || a != ns || b2 != ne) { std::printf("error=window_setup\n"); return 6; }
        }
        double tw[3][3], qw[3][4];
        for (int k = 0; k < K; k++) {
            for (int c = 0; c < 3; c++) tw[k][c] = t[c] + 0.01 * (k + 1) * (c == 0 ? 1.0 : c == 1 ? -0.5 : 0.25);
            const double ang = 0.002 * (k + 1);
            Eigen::Quaterniond d(std::cos(ang / 2), 0.0, 0.0, std::sin(ang / 2));
            Eigen::Quaterniond qq = Qb * d;
            qw[k][0] = qq.w(); qw[k][1] = qq.x(); qw[k][2] = qq.y(); qw[k][3] = qq.z();
        }
        const int mask = LILI_MASK_SURF | LILI_MASK_EDGE;
        lili::LidarWindowFactor wf(ctx, std::vector<int>{0, 1, 2}, mask, P);
        const double* pw[6] = {tw[0], qw[0], tw[1], qw[1], tw[2], qw[2]};
        std::vector<double> rw(9 * K), jw[6];
        double* jwp[6];
        for (int j = 0; j < 6; j++) { jw[j].assign((size_t)9 * K * (j % 2 ? 4 : 3), -1.0); jwp[j] = jw[j].data(); }
        if (!wf.Evaluate(pw, rw.data(), jwp)) { std::printf("error=window_evaluate\n"); return 7; }
        double wdiff = 0.0, off = 0.0;
        for (int k = 0; k < K; k++) {
            lili::LidarBatchFactor bf(ctx, k, mask, P);
            const double* pk[2] = {tw[k], qw[k]};
            double rk[9], jt[27], jq[36];
            double* jk[2] = {jt, jq};
            if (!bf.Evaluate(pk, rk, jk)) { std::printf("error=batch_evaluate\n"); return 8; }
            for (int i = 0; i < 9; i++) wdiff = std::fmax(wdiff, std::fabs(rk[i] - rw[9 * k + i]));
            for (int r = 0; r < 9 * K; r++) {
                const bool own = r / 9 == k;
                for (int c = 0; c < 3; c++) { const double v = jw[2 * k][r * 3 + c]; if (own) wdiff = std::fmax(wdiff, std::fabs(v - jt[(r % 9) * 3 + c])); else off = std::fmax(off, std::fabs(v)); }
                for (int c = 0; c < 4; c++) { const double v = jw[2 * k + 1][r * 4 + c]; if (own) wdiff = std::fmax(wdiff, std::fabs(v - jq[(r % 9) * 4 + c])); else off = std::fmax(off, std::fabs(v)); }
            }
        }
        std::printf("window_blocks=%d\nwindow_residuals=%d\nwindow_vs_batch_max_abs_diff=%.3e\nwindow_off_diagonal_max=%.3e\n", K, wf.num_residuals(), wdiff, off);
    }
    lili_ctx_destroy(ctx);
    return 0;
}
