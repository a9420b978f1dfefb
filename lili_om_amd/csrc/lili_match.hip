// C-ABI of liblili_hip.so, scan-to-map part (include/lili_hip.h: lili_s2m_*, lili_gn_step_host, lili_gram_to_factor): queries, association, linearisation,
// Gauss-Newton / LM loops, window and multi-rank variants.  Host code only; the kernels live in lili_s2m*.hip.  Reference call sites: L/src/LidarOdometry.cpp:352-561,
// L/src/BackendFusion.cpp:780-1336 (per function below).
#include "lili_launch.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static MatchParams to_device_params(const lili_s2m_params* p) {
    MatchParams m{};
    m.variant = p->variant; m.loss = p->loss; m.loss_a = p->loss_a; m.lidar_const = p->lidar_const;
    m.kd_max_radius = p->kd_max_radius; m.edge_gate = p->edge_gate; m.surf_dist_thres = p->surf_dist_thres;
    m.reflect_thres = p->reflect_thres; m.surf_weight_min = p->surf_weight_min; m.edge_dist_max = p->edge_dist_max;
    for (int i = 0; i < 4; i++) m.q_lb[i] = p->q_lb[i];
    {   // the same expression as qinv() in lili_device_math.h (this file is compiled with -ffp-contract=off as well)
        const double w = p->q_lb[0], qx = p->q_lb[1], qy = p->q_lb[2], qz = p->q_lb[3];
        const double n2 = w * w + qx * qx + qy * qy + qz * qz;
        if (n2 > 0) { m.q_lb_inv[0] = w / n2; m.q_lb_inv[1] = -qx / n2; m.q_lb_inv[2] = -qy / n2; m.q_lb_inv[3] = -qz / n2; }
        else m.q_lb_inv[0] = m.q_lb_inv[1] = m.q_lb_inv[2] = m.q_lb_inv[3] = 0.0;
        // ceres::Jet's operator/ multiplies by the reciprocal (tests/golden/ref_factors.npz pins this against the reference functor)
        const double gi = 1.0 / n2;
        if (n2 > 0) { m.q_lb_inv_jet[0] = w * gi; m.q_lb_inv_jet[1] = (-qx) * gi; m.q_lb_inv_jet[2] = (-qy) * gi; m.q_lb_inv_jet[3] = (-qz) * gi; }
        else m.q_lb_inv_jet[0] = m.q_lb_inv_jet[1] = m.q_lb_inv_jet[2] = m.q_lb_inv_jet[3] = 0.0;
    }
    for (int i = 0; i < 3; i++) m.t_lb[i] = p->t_lb[i];
    m.scale_surf_num = p->scale_surf_num; m.scale_edge_num = p->scale_edge_num;
    // profiling aid only (tools/): skip phases of the association kernels; re-read at every call so that a tool can
    // switch it on after the pose has converged
    const char* dbg = std::getenv("LILI_DEBUG");
    m.debug = dbg ? std::atoi(dbg) : 0;
    return m;
}

extern "C" {
// --------------------------------------------------------------------------------------------
// queries / associate / linearize
// --------------------------------------------------------------------------------------------
int lili_s2m_set_queries(lili_ctx* ctx, int slot, int kind, const lili_cloud* cloud) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "set_queries: bad slot");
    ARGCHK(kind == 0 || kind == 1, "set_queries: bad kind");
    ARGCHK(cloud, "set_queries: null cloud");
    HIPCHK(hipSetDevice(ctx->device));
    KindSlot& ks = ctx->slots[slot].k[kind];
    ks.has_queries = false; ks.has_records = false; ks.launches = 0;
    int rc = lili_ingest_cloud(ctx, cloud, ks.q);
    if (rc != LILI_OK) return rc;
    ks.n_q = (int64_t)cloud->n;
    ks.has_aux = cloud->aux_offset >= 0;
    ks.n_blocks = nblocks(ks.n_q, kAssocBlock);
    // linearisation: kLinBlock (1024) threads per block, at most kMaxLinBlocks partials.  The block size is FIXED (it was chosen
    // per scan size in round 1): the block partition of the Gram sum — hence its rounding — must not depend on options that may
    // change between set_queries and iterate (the fused tail needs all 1024 threads for its reduction).
    ks.n_lin_blocks = std::min(nblocks(ks.n_q, kLinBlock), kMaxLinBlocks);
    size_t n = (size_t)ks.n_q;
    if (n) {
        HIPCHK(ks.rec0.ensure(n * sizeof(float4)));
        HIPCHK(ks.rec1.ensure(n * sizeof(float4)));   // surf: doubles (8 B) fit in the float4 budget
        HIPCHK(ks.valid.ensure(n));
        HIPCHK(ks.partials.ensure((size_t)ks.n_lin_blocks * kPartialStride * sizeof(double)));
        HIPCHK(ks.block_counts.ensure((size_t)ks.n_blocks * sizeof(int)));
    }
    ks.has_queries = true;
    return LILI_OK;
}

}  // extern "C"

// Queries whose NUMBER the host does not know yet (lili_pipeline.hip: a frame's features while the extractor's kernels are still on the stream): the slot is sized for
// `n_guess` queries; the producer (k_rot_ring through a lili_query_sink) writes the first min(count, n_guess) rows and fills the rest with NaN rows — a non-finite query
// selects nothing (cell_of, the key selector), leaves no record and no count, so every sum of the iteration is the sum over the real queries in the order a slot of exactly
// `count` queries would take it (the partition of the queries into association / linearisation workgroups depends on the query's index alone; the padding workgroups add
// +0.0).  lili_s2m_trim_queries afterwards, once the count is known (count <= n_guess, else the caller sets the queries again).
int lili_s2m_set_queries_counted(lili_ctx* ctx, int slot, int kind, int n_guess) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && (kind == 0 || kind == 1) && n_guess > 0, "set_queries_counted: bad argument");
    KindSlot& ks = ctx->slots[slot].k[kind];
    ks.has_queries = false; ks.has_records = false; ks.launches = 0;
    const size_t n = (size_t)n_guess;
    HIPCHK(ks.q.ensure(n * sizeof(float4)));
    ks.n_q = n_guess;
    ks.has_aux = true;
    ks.n_blocks = nblocks(ks.n_q, kAssocBlock);
    ks.n_lin_blocks = std::min(nblocks(ks.n_q, kLinBlock), kMaxLinBlocks);
    HIPCHK(ks.rec0.ensure(n * sizeof(float4)));
    HIPCHK(ks.rec1.ensure(n * sizeof(float4)));
    HIPCHK(ks.valid.ensure(n));
    HIPCHK(ks.partials.ensure((size_t)ks.n_lin_blocks * kPartialStride * sizeof(double)));
    HIPCHK(ks.block_counts.ensure((size_t)ks.n_blocks * sizeof(int)));
    ks.has_queries = true;
    return LILI_OK;
}
// the slot's query count after the fact (n <= the count the slot was sized for): what lies behind n was padding
int lili_s2m_trim_queries(lili_ctx* ctx, int slot, int kind, int n) {
    if (!ctx) return LILI_E_ARG;
    KindSlot& ks = ctx->slots[slot].k[kind];
    ARGCHK(ks.has_queries && n >= 0 && n <= ks.n_q, "trim_queries: bad argument");
    ks.n_q = n;
    ks.n_blocks = nblocks(ks.n_q, kAssocBlock);
    ks.n_assoc_blocks = std::min(ks.n_assoc_blocks, ks.n_blocks);
    ks.n_lin_blocks = std::min(nblocks(ks.n_q, kLinBlock), kMaxLinBlocks);
    return LILI_OK;
}
extern "C" {

static int launch_associate(lili_ctx* ctx, int slot, int kind, const PoseArg& pa, const MatchParams& P) {
    KindSlot& ks = ctx->slots[slot].k[kind];
    if (!ks.has_queries) return ctx->fail(LILI_E_STATE, "associate: set_queries first");
    MapIndex& m = ctx->map[kind];
    if (!m.valid) return ctx->fail(LILI_E_STATE, "associate: map_set first");
    double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
    if (m.n > 0 && !(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach))
        return ctx->fail(LILI_E_STATE, "associate: gate radius exceeds the radius the map index was built for");
    ks.has_records = true;
    if (ks.n_q == 0) return LILI_OK;
    const int n = (int)ks.n_q;
    int* dbg_i = nullptr; float* dbg_d = nullptr;
    if (ctx->keep_nn) {
        HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
        HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
        dbg_i = ks.dbg_idx.as<int>(); dbg_d = ks.dbg_d2.as<float>();
    }
    if (m.n < 5) {   // fewer than 5 map points: the reference reads pt_search_sq_dists[4] out of bounds; we reject all
        HIPCHK(hipMemsetAsync(ks.valid.p, 0, (size_t)n, ctx->stream));
        ks.n_assoc_blocks = ks.n_blocks;
        HIPCHK(hipMemsetAsync(ks.block_counts.p, 0, (size_t)ks.n_blocks * sizeof(int), ctx->stream));
        if (dbg_i) { HIPCHK(hipMemsetAsync(dbg_i, 0xFF, (size_t)n * 5 * sizeof(int), ctx->stream)); HIPCHK(hipMemsetAsync(dbg_d, 0x7F, (size_t)n * 5 * sizeof(float), ctx->stream)); }
        return LILI_OK;
    }
    ks.n_assoc_blocks = ks.n_blocks;
    const bool any_order = pa.wait_key != 0ull;
    if (m.has_fine) {      // dense map: fine index first, gate-sized index for the queries it cannot settle
        if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux))
            return ctx->fail(LILI_E_STATE, "associate: Livox variant needs reflectivity (aux_offset) on the surf map and the surf queries");
        // the fine index alone (lili_s2m_dense.hip): one lane per query for the queries their inner 27 fine cells settle, then the wave serves the rest 16 lanes per query
        AssocArgs a{};
        a.queries = ks.q.as<float4>(); a.n_q = n; a.g = m.fview;
        a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
        a.dbg_idx = dbg_i; a.dbg_d2 = dbg_d; a.block_counts = ks.block_counts.as<int>(); a.nb = ks.n_blocks;
        ks.launches++;
        const int r_max = std::max(1, (int)std::ceil(std::sqrt(gate) * 1.01 / m.fine_cell));      // fine cells the search has to reach for the gate ball
        const int qpw = ks.n_blocks < std::max(ctx->n_simd, 256) ? 16 : kAssocBlock;                // queries per wave (see the kernel)
        const int nb = nblocks(n, qpw);
        HIPCHK(ks.block_counts.ensure((size_t)nb * sizeof(int)));
        a.nb = nb; a.block_counts = ks.block_counts.as<int>();
        ks.n_assoc_blocks = nb;
        launch_k(ctx->stream, any_order, k_associate_fine, dim3(nb), dim3(kAssocBlock), 0, a, r_max, qpw, kind, pa, P);
        HIPCHK(hipGetLastError());
        return LILI_OK;
    }
    ks.launches++;
    // one wave per workgroup (kAssocBlock): the dispatcher balances the SIMDs wave by wave
    const dim3 grid(ks.n_assoc_blocks);
    if (kind == LILI_KIND_SURF) {
        if (P.variant == LILI_VARIANT_LIVOX && !m.has_aux) return ctx->fail(LILI_E_STATE, "associate: Livox variant needs reflectivity (aux_offset) on the surf map");
        if (P.variant == LILI_VARIANT_LIVOX && !ks.has_aux) return ctx->fail(LILI_E_STATE, "associate: Livox variant needs reflectivity (aux_offset) on the surf queries");
        launch_k(ctx->stream, any_order, k_associate_surf<kAssocBlock>, grid, dim3(kAssocBlock), 0, (const float4*)ks.q.as<float4>(), n, m.view, pa, P,
                 ks.rec0.as<float4>(), ks.rec1.as<double>(), ks.valid.as<unsigned char>(), dbg_i, dbg_d, ks.block_counts.as<int>());
    } else {
        launch_k(ctx->stream, any_order, k_associate_edge<kAssocBlock>, grid, dim3(kAssocBlock), 0, (const float4*)ks.q.as<float4>(), n, m.view, pa, P,
                 ks.rec0.as<float4>(), ks.rec1.as<float4>(), ks.valid.as<unsigned char>(), dbg_i, dbg_d, ks.block_counts.as<int>());
    }
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// sums the per-block correspondence counts of the last association(s) into SlotState::n_res
static int launch_sum_counts(lili_ctx* ctx, int slot, int kind_mask, int* d_out = nullptr, const P2PView* xv = nullptr) {
    Slot& s = ctx->slots[slot];
    const int* bs = nullptr; const int* be = nullptr; int nbs = 0, nbe = 0;
    if ((kind_mask & LILI_MASK_SURF) && s.k[0].has_records && s.k[0].n_q > 0) { bs = s.k[0].block_counts.as<int>(); nbs = s.k[0].n_assoc_blocks; }
    if ((kind_mask & LILI_MASK_EDGE) && s.k[1].has_records && s.k[1].n_q > 0) { be = s.k[1].block_counts.as<int>(); nbe = s.k[1].n_assoc_blocks; }
    hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(kBlock), 0, ctx->stream, bs, nbs, be, nbe, ctx->state(slot), d_out, xv ? *xv : P2PView{});
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// Both kinds of a keyframe in one launch (k_associate_both): only the plain direct path — one wave per workgroup, caller's
// query order, no dispatch-order or binning experiments.  Returns 1 if the slot is not eligible (the caller then launches per kind).
static int launch_associate_both(lili_ctx* ctx, int slot, const PoseArg& pa, const MatchParams& P) {
    AssocArgs A[2];
    for (int kind = 0; kind < 2; kind++) {
        KindSlot& ks = ctx->slots[slot].k[kind];
        MapIndex& m = ctx->map[kind];
        if (!ks.has_queries || !m.valid || ks.n_q == 0 || m.n < 5 || m.has_fine) return 1;
        const double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
        if (!(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach)) return 1;     // the per-kind path reports the error
        if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux)) return 1;
    }
    for (int kind = 0; kind < 2; kind++) {
        KindSlot& ks = ctx->slots[slot].k[kind];
        const int n = (int)ks.n_q;
        AssocArgs& a = A[kind];
        a = AssocArgs{};
        a.queries = ks.q.as<float4>(); a.n_q = n; a.g = ctx->map[kind].view;
        a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
        if (ctx->keep_nn) {
            HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
            HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
            a.dbg_idx = ks.dbg_idx.as<int>(); a.dbg_d2 = ks.dbg_d2.as<float>();
        }
        a.block_counts = ks.block_counts.as<int>(); a.nb = ks.n_blocks;
        ks.n_assoc_blocks = ks.n_blocks; ks.has_records = true; ks.launches++;
    }
    launch_k(ctx->stream, pa.wait_key != 0ull, k_associate_both, dim3(A[0].nb + A[1].nb), dim3(kAssocBlock), 0, A[0], A[1], pa, P);
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// Lanes per query for an association launch over n queries (both kinds): small launches leave most SIMDs without a wave, so several lanes
// share a query's candidate walk (k_associate_coop).  Measured on MI355X against the 5 M-point map (tools/coop_sweep.py, wall time per
// outer iteration, ROT / front-end flavour): the best L keeps n * L near two waves per SIMD (131 072 lanes) — 2 k queries L = 16
// (25.8 -> 19.7 / 19.9 -> 15.1 us), 10 k L = 8, 20-25 k L = 4, 50 k L = 2, from 100 k on the one-lane kernels.  The FIRST association
// after a pose reset is a different launch: a good part of the queries walks the shell of the 5x5x5 block (18 more runs per query), which
// the lanes of a group split among themselves — twice the lanes pay there up to 200 k queries (200 k: 32.0 -> 25.8 us with L = 2,
// 25 k: 30.3 -> 16.4 with L = 8).  1 = the one-lane kernels.
static int coop_lanes(const lili_ctx* ctx, int64_t n, bool first_after_reset) {
    if (ctx->assoc_lpq) return ctx->assoc_lpq;
    const int64_t lanes = (int64_t)std::max(ctx->n_simd, 256) * 64 * (first_after_reset ? 4 : 2);
    int L = 1;
    while (L < 16 && n * (2 * L) <= lanes) L *= 2;
    if (first_after_reset && L == 1 && n * 2 <= lanes * 2) L = 2;
    return L;
}
// Association of the kinds in kind_mask by k_associate_coop (lili_s2m_coop.hip); `lin`: also linearise (flavours without count scaling) and
// reduce + GN-update in a second launch.  Returns 1 if the configuration is not eligible — the caller then takes the one-lane kernels.
static int launch_associate_coop(lili_ctx* ctx, int slot, int kind_mask, const PoseArg& pa, const MatchParams& P, bool lin, double* d_out) {
    if (P.debug & 4096) return 1;
    Slot& sl = ctx->slots[slot];
    int64_t n_all = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        MapIndex& m = ctx->map[kind];
        if (!ks.has_queries || !m.valid || ks.n_q == 0 || m.n < 5 || m.has_fine) return 1;
        const double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
        if (!(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach)) return 1;     // the per-kind path reports the error
        if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux)) return 1;
        n_all += ks.n_q;
    }
    const bool first = sl.assoc_since_pose == 0;      // the first association after lili_s2m_pose_set / _pose_copy: far from converged
    sl.assoc_since_pose++;
    const int L = coop_lanes(ctx, n_all, first);
    if (L < 2 || n_all == 0) return 1;
    const int qpb = 256 / L;
    // count-scaled flavours (ROT) may linearise in the association launch only through its in-launch count barrier: small grids
    const bool scaled = P.scale_surf_num > 0 || P.scale_edge_num > 0;
    int cb_blocks = 0;
    if (lin && scaled) {
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) cb_blocks += nblocks((int)sl.k[kind].n_q, qpb);
        if (cb_blocks > 256 || !ctx->count_barrier) { sl.assoc_since_pose--; return 1; }      // (the caller's three-launch path calls in again and counts the launch itself)
    }
    AssocArgs A[2] = {AssocArgs{}, AssocArgs{}};
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        const int n = (int)ks.n_q;
        AssocArgs& a = A[kind];
        a.queries = ks.q.as<float4>(); a.n_q = n; a.g = ctx->map[kind].view;
        a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
        if (ctx->keep_nn) {
            HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
            HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
            a.dbg_idx = ks.dbg_idx.as<int>(); a.dbg_d2 = ks.dbg_d2.as<float>();
        }
        a.nb = nblocks(n, qpb);
        HIPCHK(ks.block_counts.ensure((size_t)a.nb * sizeof(int)));
        if (lin) HIPCHK(ks.partials_wave.ensure((size_t)a.nb * kPartialStride * sizeof(double)));
        a.block_counts = ks.block_counts.as<int>();
        ks.n_assoc_blocks = a.nb; ks.has_records = true; ks.launches++;
    }
    const dim3 grid(A[0].nb + A[1].nb), block(256);
    double* ps = sl.k[0].partials_wave.as<double>(); double* pe = sl.k[1].partials_wave.as<double>();
#define LILI_COOP_CASE(LL) case LL: if (lin) hipLaunchKernelGGL((k_associate_coop<LL, true>), grid, block, 0, ctx->stream, A[0], A[1], pa, P, ps, pe, ctx->state(slot), cb_blocks); \
                                    else launch_k(ctx->stream, pa.wait_key != 0ull, (k_associate_coop<LL, false>), grid, block, 0, A[0], A[1], pa, P, ps, pe, ctx->state(slot), 0); break;
    switch (L) { LILI_COOP_CASE(2) LILI_COOP_CASE(4) LILI_COOP_CASE(8) LILI_COOP_CASE(16) default: return 1; }
#undef LILI_COOP_CASE
    if (lin) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, (const double*)ps, A[0].nb, (const double*)pe, A[1].nb, d_out, ctx->state(slot),
                                1 | (P.debug & 256), P2PView{}, 0ull, (double*)nullptr, ctx->take_state_mirror(slot));
    HIPCHK(hipGetLastError());
    sl.use_global_counts = false;
    return LILI_OK;
}

// The cooperative association of EVERY slot of a window in ONE launch (k_associate_coop_window): the conditions of launch_associate_coop for every slot,
// one L for all (by the total number of queries: the records do not depend on it).  Returns 1 if not eligible — the caller then launches slot by slot.
static int launch_associate_coop_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t_assoc, const double* q_assoc, const MatchParams& P) {
    if (n_slots < 2 || (P.debug & 4096)) return 1;
    int64_t n_all = 0;
    bool first = false;
    for (int i = 0; i < n_slots; i++) {
        Slot& sl = ctx->slots[slots[i]];
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
            KindSlot& ks = sl.k[kind];
            MapIndex& m = ctx->map[kind];
            if (!ks.has_queries || !m.valid || ks.n_q == 0 || m.n < 5 || m.has_fine) return 1;
            const double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
            if (!(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach)) return 1;     // the per-kind path reports the error
            if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux)) return 1;
            n_all += ks.n_q;
        }
        first = first || sl.assoc_since_pose == 0;
    }
    const int L = coop_lanes(ctx, n_all, first);
    if (L < 2 || n_all == 0) return 1;
    const int qpb = 256 / L;
    WinAssocArgs W{};
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) W.g[kind] = ctx->map[kind].view;
    int nb = 0;
    for (int i = 0; i < n_slots; i++) {
        Slot& sl = ctx->slots[slots[i]];
        WinAssocSlot& ws = W.s[i];
        ws = WinAssocSlot{};
        for (int k = 0; k < 3; k++) ws.pa.t[k] = t_assoc[3 * i + k];
        for (int k = 0; k < 4; k++) ws.pa.q[k] = q_assoc[4 * i + k];
        ws.first_block = nb;
        // (edge workgroups first, then surf: the order of k_associate_coop's grid)
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
            KindSlot& ks = sl.k[kind];
            const int n = (int)ks.n_q;
            WinAssocKind& a = ws.k[kind];
            a.queries = ks.q.as<float4>(); a.n_q = n;
            a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
            if (ctx->keep_nn) {
                HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
                HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
                a.dbg_idx = ks.dbg_idx.as<int>(); a.dbg_d2 = ks.dbg_d2.as<float>();
            }
            a.nb = nblocks(n, qpb);
            HIPCHK(ks.block_counts.ensure((size_t)a.nb * sizeof(int)));
            a.block_counts = ks.block_counts.as<int>();
            ks.n_assoc_blocks = a.nb; ks.has_records = true; ks.launches++;
            nb += a.nb;
        }
        sl.assoc_since_pose++;
        sl.use_global_counts = false; sl.sticky_global_counts = false;
    }
    W.n = n_slots;
    const dim3 grid(nb), block(256);
    switch (L) {
        case 2: hipLaunchKernelGGL((k_associate_coop_window<2>), grid, block, 0, ctx->stream, W, P); break;
        case 4: hipLaunchKernelGGL((k_associate_coop_window<4>), grid, block, 0, ctx->stream, W, P); break;
        case 8: hipLaunchKernelGGL((k_associate_coop_window<8>), grid, block, 0, ctx->stream, W, P); break;
        case 16: hipLaunchKernelGGL((k_associate_coop_window<16>), grid, block, 0, ctx->stream, W, P); break;
        default: return 1;
    }
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// n_iters outer iterations of a SMALL scan as ONE persistent launch (k_iterate_coop, lili_s2m_coop.hip): every workgroup keeps the pose in LDS,
// the workgroups exchange counts and Gram partials inside the launch and each applies the same Gauss-Newton step.  Returns 1 if not eligible
// (the caller then iterates launch by launch).  Eligible: the configurations of launch_associate_coop with at most 256 workgroups.
static int launch_iterate_persistent(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters) {
    if (!ctx->persistent_iterate || n_iters < 2 || n_iters > 2000) return 1;
    if (ctx->fuse_tail) return 1;
    MatchParams P = to_device_params(params);
    if (P.debug & (1 | 2 | 256 | 512 | 4096)) return 1;
    P.no_cost = 1;
    Slot& sl = ctx->slots[slot];
    int64_t n_all = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        MapIndex& m = ctx->map[kind];
        if (!ks.has_queries || !m.valid || ks.n_q == 0 || m.n < 5 || m.has_fine) return 1;
        const double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
        if (!(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach)) return 1;
        if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux)) return 1;
        n_all += ks.n_q;
    }
    if (n_all == 0) return 1;
    // Lanes per query: one choice for the whole registration (the converged launches decide), halved until the launch has at most 128 workgroups —
    // every workgroup has to be resident, two fit a CU, and up to four such launches may run side by side (lili_s2m_iterate_window).  The bound does
    // not depend on what else runs, so a slot iterated alone and inside a window partitions its Gram sums identically (same bits).
    int L = coop_lanes(ctx, n_all, false);
    if (L < 2) return 1;
    auto blocks_for = [&](int lanes) { int b = 0; for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) b += nblocks((int)sl.k[kind].n_q, 256 / lanes); return b; };
    while (L > 2 && blocks_for(L) > 128) L /= 2;
    const int qpb = 256 / L;
    AssocArgs A[2] = {AssocArgs{}, AssocArgs{}};
    const int nb = blocks_for(L);
    if (nb > 128 || ctx->persistent_off_now) return 1;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        const int n = (int)ks.n_q;
        AssocArgs& a = A[kind];
        a.queries = ks.q.as<float4>(); a.n_q = n; a.g = ctx->map[kind].view;
        a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
        if (ctx->keep_nn) {
            HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
            HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
            a.dbg_idx = ks.dbg_idx.as<int>(); a.dbg_d2 = ks.dbg_d2.as<float>();
        }
        a.nb = nblocks(n, qpb);
        HIPCHK(ks.block_counts.ensure((size_t)a.nb * sizeof(int)));
        a.block_counts = ks.block_counts.as<int>();
        ks.n_assoc_blocks = a.nb; ks.has_records = true; ks.launches += n_iters;
    }
    IterArgs it{};
    it.state = ctx->state(slot);
    it.nb = nb; it.ng = nb > 16 ? nblocks(nb, 16) : 1; it.n_iters = n_iters;
    it.derive_assoc = params->variant == LILI_VARIANT_FRONTEND ? 0 : 1;
    it.launch = ++ctx->lm_launches;
    HIPCHK(sl.lm_part.ensure((size_t)2 * nb * kPartialStride * sizeof(double)));
    HIPCHK(sl.lm_gsum.ensure((size_t)2 * it.ng * kPartialStride * sizeof(double)));
    HIPCHK(sl.lm_cnt.ensure((size_t)2 * (nb + it.ng) * 4 * sizeof(double)));
    it.part = sl.lm_part.as<double>(); it.gsum = sl.lm_gsum.as<double>(); it.cpart = sl.lm_cnt.as<double>();
    const dim3 grid(nb), block(256);
    switch (L) {
        case 2: hipLaunchKernelGGL(k_iterate_coop<2>, grid, block, 0, ctx->stream, A[0], A[1], P, it); break;
        case 4: hipLaunchKernelGGL(k_iterate_coop<4>, grid, block, 0, ctx->stream, A[0], A[1], P, it); break;
        case 8: hipLaunchKernelGGL(k_iterate_coop<8>, grid, block, 0, ctx->stream, A[0], A[1], P, it); break;
        case 16: hipLaunchKernelGGL(k_iterate_coop<16>, grid, block, 0, ctx->stream, A[0], A[1], P, it); break;
        default: return 1;
    }
    HIPCHK(hipGetLastError());
    sl.use_global_counts = false; sl.sticky_global_counts = false;
    sl.assoc_since_pose += n_iters;
    return LILI_OK;
}

static LinArgs lin_args_of(lili_ctx* ctx, int slot, int kind) {
    KindSlot& ks = ctx->slots[slot].k[kind];
    LinArgs A{};
    A.queries = ks.q.as<float4>(); A.n_q = (int)ks.n_q; A.rec0 = ks.rec0.as<float4>(); A.rec1 = ks.rec1.p; A.valid = ks.valid.as<unsigned char>();
    A.block_counts = ctx->slots[slot].use_global_counts ? nullptr : ks.block_counts.as<int>(); A.n_bc = ks.n_assoc_blocks;
    A.partials = ks.partials.as<double>(); A.nb = ks.n_lin_blocks;
    return A;
}

// Linearisation of the kinds in kind_mask and the reduction of their block partials to the 72-double record (and the GN update
// if do_gn).  Default: ONE launch — k_linearize covers both kinds and its last block to finish reduces (+ solves), see fused_tail.
// Options for A/B: merge_kinds = 0 (one launch per kind), fuse_tail = 0 (k_reduce_partials as its own launch).  These variants
// add the same numbers in the same order: the record is bit-identical.  NOT so the paths that linearise inside the association launch
// (fuse_lin / k_associate_lin, k_associate_coop<L, true>): there the PARTITION of the Gram sum follows the association's workgroups
// (64 or 256 queries per partial, 256 / L with L lanes per query), so lili_s2m_iterate, lili_s2m_iterate_restart with assoc_ms (which
// takes the three-launch path to bracket the association) and fuse_lin = 0 agree to ~1e-16 relative per entry, not bit for bit, and the
// low-order bits can change across the size thresholds of launch_associate_lin_reduce / coop_lanes (tests: <= 1e-10 on the pose).
// One outer iteration in TWO launches for the flavours without count scaling (k_associate_lin: association + linearisation, then the
// reduction + GN update).  Returns 1 if the configuration is not eligible (the caller then takes the three-launch path).
static int launch_associate_lin_reduce(lili_ctx* ctx, int slot, int kind_mask, const PoseArg& pa, const MatchParams& P, double* d_out) {
    if (!ctx->fuse_lin || !pa.state) return 1;
    if (ctx->fuse_tail || (P.debug & 4096)) return 1;
    const bool scaled = P.scale_surf_num > 0 || P.scale_edge_num > 0;      // ROT: only through the count barrier of the cooperative kernel (small launches)
    Slot& sl = ctx->slots[slot];
    AssocArgs A[2] = {AssocArgs{}, AssocArgs{}};
    int n_kinds = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        MapIndex& m = ctx->map[kind];
        if (!ks.has_queries || !m.valid || ks.n_q == 0 || m.n < 5 || m.has_fine) return 1;
        const double gate = kind == LILI_KIND_SURF ? P.kd_max_radius : P.edge_gate;
        if (!(std::sqrt(gate) * 1.0099 <= m.cell * (double)m.view.reach)) return 1;     // the per-kind path reports the error
        if (kind == LILI_KIND_SURF && P.variant == LILI_VARIANT_LIVOX && (!m.has_aux || !ks.has_aux)) return 1;
        n_kinds++;
    }
    if (n_kinds == 0) return 1;
    {   // small launches: several lanes per query (k_associate_coop, which linearises as well)
        const int rc = launch_associate_coop(ctx, slot, kind_mask, pa, P, true, d_out);
        if (rc != 1) return rc;
    }
    if (scaled) return 1;
    // one partial per workgroup: per wave while the reducer can take them in one round of loads (25 groups x 32), else per four waves
    int waves = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) waves += sl.k[kind].n_blocks;
    // Measured (front-end flavour, 5 M-point map): 10 k queries 19.8 vs 23.3 us per iteration, 30 k 24.6 vs 26.5, 60 k 24.8 vs 27.4, 200 k 31.6 vs 30.8 —
    // small scans are latency-bound and gain a launch; at 200 k the association waves are issue-bound and the extra rows cost more than the
    // separate linearisation launch (which runs on otherwise idle SIMDs at four waves each).
    if (waves > 1600 && !ctx->fuse_lin_block) return 1;
    const int bs = ctx->fuse_lin_block ? ctx->fuse_lin_block : (waves <= 800 ? kAssocBlock : kBlock);
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        const int n = (int)ks.n_q;
        AssocArgs& a = A[kind];
        a.queries = ks.q.as<float4>(); a.n_q = n; a.g = ctx->map[kind].view;
        a.rec0 = ks.rec0.as<float4>(); a.rec1 = ks.rec1.p; a.valid = ks.valid.as<unsigned char>();
        if (ctx->keep_nn) {
            HIPCHK(ks.dbg_idx.ensure((size_t)n * 5 * sizeof(int)));
            HIPCHK(ks.dbg_d2.ensure((size_t)n * 5 * sizeof(float)));
            a.dbg_idx = ks.dbg_idx.as<int>(); a.dbg_d2 = ks.dbg_d2.as<float>();
        }
        HIPCHK(ks.partials_wave.ensure((size_t)ks.n_blocks * kPartialStride * sizeof(double)));
        a.block_counts = ks.block_counts.as<int>(); a.nb = nblocks(n, bs);
        ks.n_assoc_blocks = a.nb; ks.has_records = true; ks.launches++;
    }
    // k_associate_lin: blocks [0, E.nb) edge, the rest surf
    if (bs == kAssocBlock) hipLaunchKernelGGL(k_associate_lin<kAssocBlock>, dim3(A[0].nb + A[1].nb), dim3(kAssocBlock), 0, ctx->stream, A[0], A[1], pa, P,
                                              sl.k[0].partials_wave.as<double>(), sl.k[1].partials_wave.as<double>());
    else hipLaunchKernelGGL(k_associate_lin<kBlock>, dim3(A[0].nb + A[1].nb), dim3(kBlock), 0, ctx->stream, A[0], A[1], pa, P,
                            sl.k[0].partials_wave.as<double>(), sl.k[1].partials_wave.as<double>());
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, (const double*)sl.k[0].partials_wave.as<double>(), A[0].nb,
                       (const double*)sl.k[1].partials_wave.as<double>(), A[1].nb, d_out, ctx->state(slot), 1 | (P.debug & 256), P2PView{}, 0ull, (double*)nullptr, ctx->take_state_mirror(slot));
    HIPCHK(hipGetLastError());
    sl.use_global_counts = false;
    return LILI_OK;
}

// pub_key (in / out, optional): != 0 asks the reduction + GN kernel to publish the new pose for an association launch that starts without waiting for it (option
// "overlap_gn", iterate_impl); set to 0 here when this call's structure has no such kernel (fused tail) — the caller then launches the association the plain way.
static int launch_linearize_reduce(lili_ctx* ctx, int slot, int kind_mask, const PoseArg& pa, const MatchParams& P, double* d_out, int do_gn,
                                   const P2PView* xv = nullptr, unsigned long long* pub_key = nullptr) {
    Slot& s = ctx->slots[slot];
    LinArgs A[2] = {LinArgs{}, LinArgs{}};
    int n_kinds = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = s.k[kind];
        if (!ks.has_records) return ctx->fail(LILI_E_STATE, "linearize: associate first");
        if (ks.n_q == 0) continue;
        A[kind] = lin_args_of(ctx, slot, kind);
        n_kinds++;
    }
    const int* ng = s.use_global_counts ? s.global_counts : nullptr;
    FuseTail fz{};
    fz.mode = (ctx->fuse_tail && n_kinds > 0 && !xv) ? (do_gn ? 2 : 1) : 0;     // the exchange across ranks lives in k_reduce_partials
    fz.out = d_out; fz.state = ctx->state(slot); fz.debug = P.debug;
    fz.part_surf = A[0].partials; fz.nb_surf = A[0].nb; fz.part_edge = A[1].partials; fz.nb_edge = A[1].nb;
    const FuseTail off{};
    const size_t lds = lds_linearize(kLinBlock);
    if (n_kinds == 2 && !ctx->merge_kinds) {       // A/B: one launch per kind, the tail on the second
        FuseTail pub = off;
        if (fz.mode) { pub = fz; pub.mode = 3; }     // publish granules under the same key; the edge launch reduces both kinds
        hipLaunchKernelGGL(k_linearize, dim3(A[0].nb), dim3(kLinBlock), lds, ctx->stream, A[0], LinArgs{}, pa, P, ctx->state(slot), ng, pub);
        hipLaunchKernelGGL(k_linearize, dim3(A[1].nb), dim3(kLinBlock), lds, ctx->stream, LinArgs{}, A[1], pa, P, ctx->state(slot), ng, fz);
    } else if (n_kinds > 0) {
        hipLaunchKernelGGL(k_linearize, dim3(A[0].nb + A[1].nb), dim3(kLinBlock), lds, ctx->stream, A[0], A[1], pa, P, ctx->state(slot), ng, fz);
    }
    HIPCHK(hipGetLastError());
    if (!fz.mode) {
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, fz.part_surf, fz.nb_surf, fz.part_edge, fz.nb_edge, d_out, ctx->state(slot), (do_gn ? 1 : 0) | (P.debug & 256),
                           xv ? *xv : P2PView{}, (pub_key && do_gn) ? *pub_key : 0ull, ctx->pub_of(slot), do_gn ? ctx->take_state_mirror(slot) : nullptr);
        HIPCHK(hipGetLastError());
    } else if (pub_key) *pub_key = 0ull;
    if (pub_key && !do_gn) *pub_key = 0ull;
    return LILI_OK;
}

// Page-locked landing area of the blocking calls' results, written by their last kernel across PCIe (a record is 576 bytes): the host reads it after the one
// synchronisation of the call — no device-to-host copy launch in between (~4 us of GPU time and an API call per blocking evaluation).
constexpr size_t kHRecordDoubles = (size_t)LILI_MAX_SLOTS * LILI_GRAM_DOUBLES + 2 * LILI_MAX_SLOTS;      // records | 2 x MAX_SLOTS window counts | 2 x MAX_SLOTS per-slot counts (ints)
static int ensure_h_records(lili_ctx* ctx) {
    if (ctx->h_records) return LILI_OK;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_records), kHRecordDoubles * sizeof(double), hipHostMallocDefault));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, ctx->h_records, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
    ctx->h_records_dev = static_cast<double*>(d);
    return LILI_OK;
}

int lili_s2m_associate(lili_ctx* ctx, int slot, int kind, const double t_assoc[3], const double q_assoc[4],
                       const lili_s2m_params* params, int* n_res) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "associate: bad slot");
    ARGCHK(kind == 0 || kind == 1, "associate: bad kind");
    ARGCHK(t_assoc && q_assoc && params, "associate: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    PoseArg pa{};
    for (int i = 0; i < 3; i++) pa.t[i] = t_assoc[i];
    for (int i = 0; i < 4; i++) pa.q[i] = q_assoc[i];
    pa.state = nullptr; pa.derive_assoc = 0;
    MatchParams P = to_device_params(params);
    ctx->slots[slot].use_global_counts = false; ctx->slots[slot].sticky_global_counts = false;
    int rc = launch_associate_coop(ctx, slot, 1 << kind, pa, P, false, nullptr);
    if (rc == 1) rc = launch_associate(ctx, slot, kind, pa, P);
    if (rc != LILI_OK) return rc;
    if (n_res) {
        if ((rc = ensure_h_records(ctx)) != LILI_OK) return rc;
        if (ctx->h_records_dev) {              // k_sum_counts writes the two counts into page-locked memory itself: no copy launch before the synchronisation
            const size_t off = (size_t)LILI_MAX_SLOTS * LILI_GRAM_DOUBLES * sizeof(double) + (size_t)(2 * LILI_MAX_SLOTS + 2 * slot) * sizeof(int);
            rc = launch_sum_counts(ctx, slot, 1 << kind, reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->h_records_dev) + off));
            if (rc != LILI_OK) return rc;
            HIPCHK(hipStreamSynchronize(ctx->stream));
            *n_res = reinterpret_cast<const int*>(reinterpret_cast<const char*>(ctx->h_records) + off)[kind];
        } else {
            rc = launch_sum_counts(ctx, slot, 1 << kind);
            if (rc != LILI_OK) return rc;
            rc = lili_readback_add(ctx, n_res, &ctx->state(slot)->n_res[kind], sizeof(int));
            if (rc == LILI_OK) rc = lili_readback_finish(ctx);
            if (rc != LILI_OK) return rc;
        }
    }
    return LILI_OK;
}

static int window_args(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, WindowArgs& w, const char* who);
// Association of SEVERAL slots in one call (the keyframes of the sliding window, L/src/BackendFusion.cpp:919-936: findCorrespondingSurfFeatures +
// findCorrespondingCornerFeatures per keyframe): both kinds of a slot share a launch where possible, the slots run on forked streams, the
// correspondence counts come back in ONE synchronisation.  Results are those of lili_s2m_associate per slot and kind.
int lili_s2m_associate_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t_assoc /*3 per slot*/, const double* q_assoc /*4 per slot*/,
                              const lili_s2m_params* params, int* n_res /*2 per slot: surf, edge; optional*/) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, "associate_window: 1..LILI_MAX_SLOTS slots");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "associate_window: bad kind mask");
    ARGCHK(t_assoc && q_assoc && params, "associate_window: null argument");
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "associate_window: bad slot");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], "associate_window: duplicate slot");
    }
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t main_stream = ctx->stream;
    const MatchParams P = to_device_params(params);
    int rc = launch_associate_coop_window(ctx, slots, n_slots, kind_mask, t_assoc, q_assoc, P);      // every keyframe in ONE launch where the cooperative kernel applies
    const bool one_launch = rc == LILI_OK;
    if (rc != LILI_OK && rc != 1) return rc;
    rc = LILI_OK;
    if (!one_launch) {
        if (!ctx->fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));
    }
    for (int i = 0; i < n_slots && rc == LILI_OK && !one_launch; i++) {
        if (i > 0) {
            if (!ctx->side[i]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
            if (!ctx->join_ev[i]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming));
            HIPCHK(hipStreamWaitEvent(ctx->side[i], ctx->fork_ev, 0));
            ctx->stream = ctx->side[i];
        }
        PoseArg pa{};
        for (int k = 0; k < 3; k++) pa.t[k] = t_assoc[3 * i + k];
        for (int k = 0; k < 4; k++) pa.q[k] = q_assoc[4 * i + k];
        ctx->slots[slots[i]].use_global_counts = false; ctx->slots[slots[i]].sticky_global_counts = false;
        rc = launch_associate_coop(ctx, slots[i], kind_mask, pa, P, false, nullptr);
        if (rc == 1 && kind_mask == (LILI_MASK_SURF | LILI_MASK_EDGE) && ctx->merge_kinds) rc = launch_associate_both(ctx, slots[i], pa, P);
        if (rc == 1) {                                   // not eligible (or one kind only): one launch per kind
            rc = LILI_OK;
            for (int kind = 0; kind < 2 && rc == LILI_OK; kind++) if (kind_mask & (1 << kind)) rc = launch_associate(ctx, slots[i], kind, pa, P);
        }
        hipError_t e = hipSuccess;
        if (i > 0) {
            e = hipEventRecord(ctx->join_ev[i], ctx->side[i]);
            ctx->stream = main_stream;
            if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, ctx->join_ev[i], 0);
        }
        if (e != hipSuccess) { ctx->stream = main_stream; return ctx->fail(LILI_E_HIP, std::string("associate_window: ") + hipGetErrorString(e)); }
    }
    ctx->stream = main_stream;
    if (rc != LILI_OK) { HIPCHK(hipStreamSynchronize(ctx->stream)); return rc; }      // nothing of this call stays in flight
    if (n_res) {
        // the counts of every slot in ONE launch (k_window_counts: [surf, edge] per slot, also left in the slots' states) and ONE read-back
        // through the page-locked scratch — it was one k_sum_counts + one copy per slot on the forked streams
        WindowArgs w;
        if ((rc = window_args(ctx, slots, n_slots, kind_mask, w, "associate_window")) != LILI_OK) return rc;
        if ((rc = ensure_h_records(ctx)) != LILI_OK) return rc;
        int host[2 * LILI_MAX_SLOTS];
        if (ctx->h_records_dev) {          // the counts land in page-locked memory straight from the kernel
            int* d_out = reinterpret_cast<int*>(ctx->h_records_dev + (size_t)LILI_MAX_SLOTS * LILI_GRAM_DOUBLES);
            hipLaunchKernelGGL(k_window_counts, dim3(1), dim3(kBlock), 0, ctx->stream, w, d_out, P2PView{});
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::memcpy(host, ctx->h_records + (size_t)LILI_MAX_SLOTS * LILI_GRAM_DOUBLES, sizeof(int) * 2 * n_slots);
        } else {
            HIPCHK(ctx->win_counts.ensure(sizeof(int) * 2 * LILI_MAX_SLOTS));
            hipLaunchKernelGGL(k_window_counts, dim3(1), dim3(kBlock), 0, ctx->stream, w, ctx->win_counts.as<int>(), P2PView{});
            HIPCHK(hipGetLastError());
            rc = lili_readback_add(ctx, host, ctx->win_counts.p, sizeof(int) * 2 * n_slots);
            if (rc == LILI_OK) rc = lili_readback_finish(ctx);
            if (rc != LILI_OK) return rc;
        }
        for (int i = 0; i < n_slots; i++) {
            n_res[2 * i] = (kind_mask & LILI_MASK_SURF) ? host[2 * i] : 0;
            n_res[2 * i + 1] = (kind_mask & LILI_MASK_EDGE) ? host[2 * i + 1] : 0;
        }
    }
    return LILI_OK;
}

// Linearisation of SEVERAL slots in one call (one evaluation of the joint sliding window: a Gram per keyframe, L/src/BackendFusion.cpp:919-980 under
// ceres::Solve's up to 15 evaluations; the body of lili::LidarWindowFactor::Evaluate in include/lili_ceres_adapter.h).  Round 4: TWO launches and
// ONE read-back whatever the number of keyframes — k_linearize_window (every slot's records, the slot in the block index), k_window_reduce (every
// slot's block partials -> n x 72 doubles), one copy through the page-locked scratch, one synchronisation.  It was three forked streams, three
// event pairs, 2 n launches and n copies (80 us per 3-keyframe evaluation against 29 us for one keyframe alone).  Results are those of
// lili_s2m_linearize per slot, bit for bit (same bodies, same block geometry, the additions of reduce_partials_block in the same order).
static int linearize_window_impl(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const double* t, const double* q,
                                 lili_allreduce_fn allreduce, void* comm, double* d_gram, int do_gn);
int lili_s2m_linearize_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t /*3 per slot*/, const double* q /*4 per slot*/,
                              const lili_s2m_params* params, double* gram /*64 per slot*/, double* cost /*1 per slot, optional*/, int* counts /*2 per slot, optional*/) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, "linearize_window: 1..LILI_MAX_SLOTS slots");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "linearize_window: bad kind mask");
    ARGCHK(t && q && params && gram, "linearize_window: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = ensure_h_records(ctx);
    if (rc != LILI_OK) return rc;
    double host[LILI_GRAM_DOUBLES * LILI_MAX_SLOTS];
    if (ctx->h_records_dev) {              // k_window_reduce writes the n records into page-locked memory itself
        rc = linearize_window_impl(ctx, slots, n_slots, kind_mask, params, t, q, nullptr, nullptr, ctx->h_records_dev, 0);
        if (rc != LILI_OK) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::memcpy(host, ctx->h_records, sizeof(double) * LILI_GRAM_DOUBLES * n_slots);
    } else {
        HIPCHK(ctx->win_rec.ensure(sizeof(double) * LILI_GRAM_DOUBLES * LILI_MAX_SLOTS));
        rc = linearize_window_impl(ctx, slots, n_slots, kind_mask, params, t, q, nullptr, nullptr, ctx->win_rec.as<double>(), 0);
        if (rc != LILI_OK) return rc;
        rc = lili_readback_add(ctx, host, ctx->win_rec.p, sizeof(double) * LILI_GRAM_DOUBLES * n_slots);
        if (rc == LILI_OK) rc = lili_readback_finish(ctx);
        if (rc != LILI_OK) return rc;
    }
    for (int i = 0; i < n_slots; i++) {
        const double* h = host + (size_t)i * LILI_GRAM_DOUBLES;
        std::memcpy(gram + (size_t)64 * i, h, 64 * sizeof(double));
        if (cost) cost[i] = h[64];
        if (counts) { counts[2 * i] = (int)h[65]; counts[2 * i + 1] = (int)h[66]; }
    }
    return LILI_OK;
}

int lili_s2m_linearize(lili_ctx* ctx, int slot, int kind_mask, const double t[3], const double q[4],
                       const lili_s2m_params* params, double gram[64], double* cost, int counts[2]) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "linearize: bad slot");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "linearize: bad kind mask");
    ARGCHK(t && q && params && gram, "linearize: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    PoseArg pa{};
    for (int i = 0; i < 3; i++) pa.t[i] = t[i];
    for (int i = 0; i < 4; i++) pa.q[i] = q[i];
    MatchParams P = to_device_params(params);
    int rc = ensure_h_records(ctx);
    if (rc != LILI_OK) return rc;
    double host[LILI_GRAM_DOUBLES];
    if (ctx->h_records_dev) {              // k_reduce_partials writes the record into page-locked memory itself
        double* d_out = ctx->h_records_dev + (size_t)slot * LILI_GRAM_DOUBLES;
        rc = launch_linearize_reduce(ctx, slot, kind_mask, pa, P, d_out, 0);
        if (rc != LILI_OK) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::memcpy(host, ctx->h_records + (size_t)slot * LILI_GRAM_DOUBLES, sizeof(host));
    } else {
        rc = launch_linearize_reduce(ctx, slot, kind_mask, pa, P, ctx->gram_of(slot), 0);
        if (rc != LILI_OK) return rc;
        rc = lili_readback_add(ctx, host, ctx->gram_of(slot), sizeof(host));
        if (rc == LILI_OK) rc = lili_readback_finish(ctx);
        if (rc != LILI_OK) return rc;
    }
    std::memcpy(gram, host, 64 * sizeof(double));
    if (cost) *cost = host[64];
    if (counts) { counts[0] = (int)host[65]; counts[1] = (int)host[66]; }
    return LILI_OK;
}

// --------------------------------------------------------------------------------------------
// record copy-out (debug / parity; ordered like the reference's push_back lists)
// --------------------------------------------------------------------------------------------
int lili_s2m_get_surf_records(lili_ctx* ctx, int slot, size_t capacity, int32_t* query_index, float* cur_pt, float* normal,
                              float* neg_oa_dot_norm, double* score, size_t* n_out) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "get_surf_records: bad slot");
    KindSlot& ks = ctx->slots[slot].k[0];
    if (!ks.has_records) return ctx->fail(LILI_E_STATE, "get_surf_records: associate first");
    HIPCHK(hipSetDevice(ctx->device));
    size_t n = (size_t)ks.n_q;
    std::vector<unsigned char> v(n); std::vector<float4> q(n), nd(n); std::vector<double> sc(n);
    if (n) {
        HIPCHK(hipMemcpyAsync(v.data(), ks.valid.p, n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(q.data(), ks.q.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(nd.data(), ks.rec0.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(sc.data(), ks.rec1.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (!v[i]) continue;
        if (k < capacity) {
            if (query_index) query_index[k] = (int32_t)i;
            if (cur_pt) { cur_pt[3 * k] = q[i].x; cur_pt[3 * k + 1] = q[i].y; cur_pt[3 * k + 2] = q[i].z; }
            if (normal) { normal[3 * k] = nd[i].x; normal[3 * k + 1] = nd[i].y; normal[3 * k + 2] = nd[i].z; }
            if (neg_oa_dot_norm) neg_oa_dot_norm[k] = nd[i].w;
            if (score) score[k] = sc[i];
        }
        k++;
    }
    if (n_out) *n_out = k;
    return LILI_OK;
}

int lili_s2m_get_edge_records(lili_ctx* ctx, int slot, size_t capacity, int32_t* query_index, float* cur_pt, float* pt_a, float* pt_b,
                              float* s, size_t* n_out) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "get_edge_records: bad slot");
    KindSlot& ks = ctx->slots[slot].k[1];
    if (!ks.has_records) return ctx->fail(LILI_E_STATE, "get_edge_records: associate first");
    HIPCHK(hipSetDevice(ctx->device));
    size_t n = (size_t)ks.n_q;
    std::vector<unsigned char> v(n); std::vector<float4> q(n), a(n), b(n);
    if (n) {
        HIPCHK(hipMemcpyAsync(v.data(), ks.valid.p, n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(q.data(), ks.q.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(a.data(), ks.rec0.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(b.data(), ks.rec1.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (!v[i]) continue;
        if (k < capacity) {
            if (query_index) query_index[k] = (int32_t)i;
            if (cur_pt) { cur_pt[3 * k] = q[i].x; cur_pt[3 * k + 1] = q[i].y; cur_pt[3 * k + 2] = q[i].z; }
            if (pt_a) { pt_a[3 * k] = a[i].x; pt_a[3 * k + 1] = a[i].y; pt_a[3 * k + 2] = a[i].z; }
            if (pt_b) { pt_b[3 * k] = b[i].x; pt_b[3 * k + 1] = b[i].y; pt_b[3 * k + 2] = b[i].z; }
            if (s) s[k] = a[i].w;
        }
        k++;
    }
    if (n_out) *n_out = k;
    return LILI_OK;
}

int lili_s2m_get_neighbors(lili_ctx* ctx, int slot, int kind, size_t n_q, int32_t* idx, float* d2) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && (kind == 0 || kind == 1), "get_neighbors: bad slot/kind");
    KindSlot& ks = ctx->slots[slot].k[kind];
    if (!ks.has_records || !ctx->keep_nn || !ks.dbg_idx.p) return ctx->fail(LILI_E_STATE, "get_neighbors: enable lili_set_debug before associate");
    ARGCHK(n_q == (size_t)ks.n_q, "get_neighbors: n_q mismatch");
    HIPCHK(hipSetDevice(ctx->device));
    if (n_q) {
        if (idx) HIPCHK(hipMemcpyAsync(idx, ks.dbg_idx.p, n_q * 5 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        if (d2) HIPCHK(hipMemcpyAsync(d2, ks.dbg_d2.p, n_q * 5 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return LILI_OK;
}

// --------------------------------------------------------------------------------------------
// device-resident iterations
// --------------------------------------------------------------------------------------------
int lili_s2m_pose_set(lili_ctx* ctx, int slot, const double t[3], const double q[4]) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && t && q, "pose_set: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    SlotState s{};
    for (int i = 0; i < 3; i++) s.pose[i] = t[i];
    for (int i = 0; i < 4; i++) s.pose[3 + i] = q[i];
    // everything but the launch epoch (the last member): its key tags the granules of the fused linearisation launches and must
    // never repeat while stale granules of this slot's partial buffers may still carry it
    HIPCHK(hipMemcpyAsync(ctx->state(slot), &s, offsetof(SlotState, epoch), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // `s` is on this stack frame
    ctx->slots[slot].assoc_since_pose = 0;
    return LILI_OK;
}

int lili_s2m_pose_get(lili_ctx* ctx, int slot, double t[3], double q[4], int* gn_status) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "pose_get: bad slot");
    HIPCHK(hipSetDevice(ctx->device));
    SlotState s{};
    { int rb = lili_readback_add(ctx, &s, ctx->state(slot), sizeof(s)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    if (t) for (int i = 0; i < 3; i++) t[i] = s.pose[i];
    if (q) for (int i = 0; i < 4; i++) q[i] = s.pose[3 + i];
    if (gn_status) *gn_status = s.gn_status;
    if (s.wait_failed) return ctx->fail(LILI_E_STATE, "pose_get: an association launched ahead of its Gauss-Newton update (option overlap_gn) gave up waiting for the published pose; "
                                                       "the slot's results since then are not valid — set_option(\"overlap_gn\", 0) and restart the registration");
    return LILI_OK;
}

// The step the last Gauss-Newton update of `slot` took (lili_s2m_iterate* apply undamped GN steps; ceres::Solve in the reference rejects
// steps that do not decrease the cost — a caller that starts far from the solution can guard with this, or drive
// lili_s2m_linearize + its own trust region as include/lili_ceres_adapter.h does).
int lili_s2m_last_step(lili_ctx* ctx, int slot, double delta[6], int* n_updates, int* gn_status) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "last_step: bad slot");
    HIPCHK(hipSetDevice(ctx->device));
    SlotState s{};
    { int rb = lili_readback_add(ctx, &s, ctx->state(slot), sizeof(s)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    if (delta) for (int i = 0; i < 6; i++) delta[i] = s.last_delta[i];
    if (n_updates) *n_updates = s.iters;
    if (gn_status) *gn_status = s.gn_status;
    return LILI_OK;
}

// profiling aid (LILI_DEBUG bit 256): the 16 device timestamps (100 MHz ticks) of the slot's last launches
int lili_s2m_debug_times(lili_ctx* ctx, int slot, long long out[16]) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && out, "debug_times: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    SlotState s{};
    { int rb = lili_readback_add(ctx, &s, ctx->state(slot), sizeof(s)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    for (int i = 0; i < 16; i++) out[i] = s.tprof[i];
    return LILI_OK;
}

static int associate_dev_impl(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, unsigned long long wait_key);
int lili_s2m_associate_dev(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params) { return associate_dev_impl(ctx, slot, kind_mask, params, 0ull); }
// wait_key != 0 (iterate_impl, option "overlap_gn"): the launch carries no barrier against the reduction + GN kernel enqueued right before it and takes the pose from
// the granules that kernel publishes (load_assoc_pose / wait_published_pose)
static int associate_dev_impl(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, unsigned long long wait_key) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "associate_dev: bad slot");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "associate_dev: bad kind mask");
    ARGCHK(params, "associate_dev: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    Slot& s = ctx->slots[slot];
    s.use_global_counts = false; s.sticky_global_counts = false;
    PoseArg pa{};
    pa.state = ctx->state(slot);
    pa.derive_assoc = params->variant == LILI_VARIANT_FRONTEND ? 0 : 1;
    pa.wait_key = wait_key;
    pa.pub = ctx->pub_of(slot);
    MatchParams P = to_device_params(params);
    {   // small launches: several lanes per query
        int rc = launch_associate_coop(ctx, slot, kind_mask, pa, P, false, nullptr);
        if (rc != 1) return rc;
    }
    if (kind_mask == (LILI_MASK_SURF | LILI_MASK_EDGE) && ctx->merge_kinds) {
        int rc = launch_associate_both(ctx, slot, pa, P);
        if (rc != 1) return rc;     // 1 = not eligible: one launch per kind below
    }
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        int rc = launch_associate(ctx, slot, kind, pa, P);
        if (rc != LILI_OK) return rc;
    }
    return LILI_OK;
}

int lili_s2m_counts_export(lili_ctx* ctx, int slot, int32_t* d_counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && d_counts, "counts_export: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    return launch_sum_counts(ctx, slot, LILI_MASK_SURF | LILI_MASK_EDGE, d_counts);   // the kernel writes the caller's buffer directly
}

int lili_s2m_counts_import(lili_ctx* ctx, int slot, const int32_t* d_counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && d_counts, "counts_import: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->slots[slot].global_counts = d_counts;   // read by the next linearize_dev's kernels (no copy): keep it valid until then
    ctx->slots[slot].use_global_counts = true;
    return LILI_OK;
}

static int linearize_dev_impl(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, double* d_gram, int do_gn, int want_cost = 0, unsigned long long* pub_key = nullptr) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "linearize_dev: bad slot");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "linearize_dev: bad kind mask");
    ARGCHK(params && d_gram, "linearize_dev: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    PoseArg pa{};
    pa.state = ctx->state(slot);
    MatchParams P = to_device_params(params);
    if (do_gn && d_gram == ctx->gram_of(slot) && !want_cost) P.no_cost = 1;   // lili_s2m_iterate*: the record stays inside the library, only the GN step is used
    int rc = launch_linearize_reduce(ctx, slot, kind_mask, pa, P, d_gram, do_gn, nullptr, pub_key);
    if (rc != LILI_OK) return rc;
    ctx->slots[slot].use_global_counts = false;
    return LILI_OK;
}

int lili_s2m_linearize_dev(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, double* d_gram) {
    return linearize_dev_impl(ctx, slot, kind_mask, params, d_gram, 0);
}

// The reference back-end's INNER iteration (ceres::Solve's loop on fixed correspondences, L/src/BackendFusion.cpp:984-992: up to
// max_num_iter evaluations of every residual block + one dense solve each): n_iters x [linearise at the device pose + reduce +
// GN update] on the records of the last association, one launch each (fused tail).
int lili_s2m_iterate_inner(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int want_cost) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(n_iters >= 0, "iterate_inner: negative n_iters");
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "iterate_inner: bad slot");
    for (int it = 0; it < n_iters; it++) {
        int rc = linearize_dev_impl(ctx, slot, kind_mask, params, ctx->gram_of(slot), 1, want_cost);
        if (rc != LILI_OK) return rc;
    }
    return LILI_OK;
}

// --------------------------------------------------------------------------------------------
// Levenberg-Marquardt on the device (lili_s2m_lm.hip)
// --------------------------------------------------------------------------------------------
void lili_lm_default_options(lili_lm_options* o) {      // Ceres 2.0 Solver::Options defaults; max_num_iterations as the reference sets it (max_num_iter = 15)
    if (!o) return;
    o->max_iterations = 15; o->reserved_ = 0;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
}

// enqueues the persistent launch of one slot on ctx->stream; max_blocks bounds the grid (all workgroups have to be resident)
static int launch_solve_lm(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, const lili_lm_options* options, int max_blocks, LmArgs* prepared = nullptr /* fill the arguments only: the caller launches (k_solve_lm_window) */) {
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS, "solve_lm: bad slot");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, "solve_lm: bad kind mask");
    ARGCHK(params, "solve_lm: null params");
    lili_lm_options opt;
    if (options) opt = *options; else lili_lm_default_options(&opt);
    ARGCHK(opt.max_iterations >= 1 && opt.max_iterations <= 1000, "solve_lm: max_iterations must be in 1..1000");
    ARGCHK(opt.initial_radius > 0 && opt.max_radius >= opt.initial_radius && opt.min_radius > 0 && opt.min_lm_diagonal > 0 && opt.max_lm_diagonal >= opt.min_lm_diagonal,
           "solve_lm: inconsistent trust-region options");
    Slot& sl = ctx->slots[slot];
    LmArgs a{};
    int64_t n_all = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = sl.k[kind];
        if (!ks.has_records) return ctx->fail(LILI_E_STATE, "solve_lm: associate first");
        if (ks.n_q == 0) continue;
        LinArgs A = lin_args_of(ctx, slot, kind);
        A.block_counts = ks.block_counts.as<int>(); A.partials = nullptr;
        (kind == 0 ? a.S : a.E) = A;
        n_all += ks.n_q;
    }
    if (n_all == 0) return ctx->fail(LILI_E_STATE, "solve_lm: no records");
    // workgroups: 512 records each, split between the kinds in proportion, at most max_blocks in total (a kind that is present gets at least one)
    constexpr int kLmThreads = 512;      // must match lili_s2m_lm.hip
    const int want_s = a.S.n_q > 0 ? nblocks(a.S.n_q, kLmThreads) : 0, want_e = a.E.n_q > 0 ? nblocks(a.E.n_q, kLmThreads) : 0;
    int nb_s = want_s, nb_e = want_e;
    if (want_s > 0 && want_e > 0 && max_blocks < 2)     // every kind present needs a workgroup of its own, and all of them must be resident
        return ctx->fail(LILI_E_STATE, "solve_lm: fewer resident workgroups available than feature kinds (too many slots side by side)");
    if (nb_s + nb_e > max_blocks) {
        nb_e = want_e ? std::max(1, (int)((int64_t)max_blocks * want_e / (want_s + want_e))) : 0;
        nb_s = want_s ? std::max(1, max_blocks - nb_e) : 0;
    }
    a.S.nb = nb_s; a.E.nb = nb_e;
    a.nb = nb_s + nb_e;
    a.ng = a.nb > 16 ? nblocks(a.nb, 16) : 1;       // kLmGroup of lili_s2m_lm.hip
    HIPCHK(sl.lm_part.ensure((size_t)2 * a.nb * kPartialStride * sizeof(double)));
    HIPCHK(sl.lm_gsum.ensure((size_t)2 * a.ng * kPartialStride * sizeof(double)));
    HIPCHK(sl.lm_summary.ensure(sizeof(lili_lm_summary)));
    a.part = sl.lm_part.as<double>(); a.gsum = sl.lm_gsum.as<double>();
    a.state = ctx->state(slot);
    a.max_iter = opt.max_iterations;
    a.launch = ++ctx->lm_launches;
    a.summary = sl.lm_summary.as<lili_lm_summary>();
    a.function_tolerance = opt.function_tolerance; a.gradient_tolerance = opt.gradient_tolerance; a.parameter_tolerance = opt.parameter_tolerance;
    a.initial_radius = opt.initial_radius; a.max_radius = opt.max_radius; a.min_radius = opt.min_radius; a.min_relative_decrease = opt.min_relative_decrease;
    a.min_lm_diagonal = opt.min_lm_diagonal; a.max_lm_diagonal = opt.max_lm_diagonal;
    sl.use_global_counts = false;
    sl.assoc_since_pose = 1;
    if (prepared) { *prepared = a; return LILI_OK; }
    MatchParams P = to_device_params(params);
    P.no_cost = 0;                              // the robust cost drives the accept / reject decisions
    hipLaunchKernelGGL(k_solve_lm, dim3(a.nb), dim3(kLmThreads), lds_linearize(kLmThreads), ctx->stream, a, P);
    HIPCHK(hipGetLastError());
    sl.assoc_since_pose = 1;                    // the pose moved, but stays near the association's: the next association is no "first" one
    return LILI_OK;
}

int lili_s2m_solve_lm(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, const lili_lm_options* options, lili_lm_summary* summary) {
    if (!ctx) return LILI_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const int max_blocks = std::max(1, std::min(ctx->n_simd / 4 - 16, 240));     // one workgroup per CU, a few CUs left to whatever else runs
    int rc = launch_solve_lm(ctx, slot, kind_mask, params, options, max_blocks);
    if (rc != LILI_OK) return rc;
    if (summary) {
        rc = lili_readback_add(ctx, summary, ctx->slots[slot].lm_summary.p, sizeof(lili_lm_summary));
        if (rc == LILI_OK) rc = lili_readback_finish(ctx);
        if (rc != LILI_OK) return rc;
    }
    return LILI_OK;
}

int lili_s2m_solve_lm_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const lili_lm_options* options,
                             lili_lm_summary* summaries) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, "solve_lm_window: 1..LILI_MAX_SLOTS slots");
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "solve_lm_window: bad slot");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], "solve_lm_window: duplicate slot");
    }
    HIPCHK(hipSetDevice(ctx->device));
    // ONE launch for all slots (round 6: k_solve_lm_window); every slot needs all its workgroups resident: the CUs are shared out
    const int max_blocks = std::max(1, std::min(ctx->n_simd / 4 - 16, 240) / n_slots);
    WinLmArgs W{};
    W.n = n_slots;
    int nb = 0;
    for (int i = 0; i < n_slots; i++) {
        const int rc = launch_solve_lm(ctx, slots[i], kind_mask, params, options, max_blocks, &W.a[i]);
        if (rc != LILI_OK) return rc;
        W.first_block[i] = nb;
        nb += W.a[i].nb;
    }
    MatchParams P = to_device_params(params);
    P.no_cost = 0;                              // the robust cost drives the accept / reject decisions
    hipLaunchKernelGGL(k_solve_lm_window, dim3(nb), dim3(512 /* kLmThreads of lili_s2m_lm.hip */), lds_linearize(512), ctx->stream, W, P);
    HIPCHK(hipGetLastError());
    if (summaries) {
        for (int i = 0; i < n_slots; i++) { const int rb = lili_readback_add(ctx, summaries + i, ctx->slots[slots[i]].lm_summary.p, sizeof(lili_lm_summary)); if (rb != LILI_OK) { (void)lili_readback_finish(ctx); return rb; } }
        return lili_readback_finish(ctx);
    }
    return LILI_OK;
}

int lili_s2m_accumulate(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, double* d_gram) {
    int rc = lili_s2m_associate_dev(ctx, slot, kind_mask, params);
    if (rc != LILI_OK) return rc;
    return lili_s2m_linearize_dev(ctx, slot, kind_mask, params, d_gram);
}

int lili_s2m_gn_update(lili_ctx* ctx, int slot, const double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && d_gram, "gn_update: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_gn_update, dim3(1), dim3(64), 0, ctx->stream, d_gram, ctx->state(slot));
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

int lili_s2m_pose_copy(lili_ctx* ctx, int dst_slot, int src_slot) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(dst_slot >= 0 && dst_slot < LILI_MAX_SLOTS && src_slot >= 0 && src_slot < LILI_MAX_SLOTS, "pose_copy: bad slot");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_pose_copy, dim3(1), dim3(8), 0, ctx->stream, ctx->state(dst_slot), ctx->state(src_slot));
    HIPCHK(hipGetLastError());
    ctx->slots[dst_slot].assoc_since_pose = 0;
    return LILI_OK;
}

// n_iters outer iterations; if restart_every > 0 the pose of `slot` is re-initialised from `restart_slot` before
// iterations 0, restart_every, 2*restart_every, ... (device-to-device, async) — "one registration = restart_every
// GN iterations".  If assoc_ms is non-NULL the association launches are bracketed by HIP events on the context's
// stream and their total duration is returned (this variant synchronises at the end).
// one outer iteration through k_associate_lin (see launch_associate_lin_reduce); 1 = not eligible
static int iterate_fused_lin(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params) {
    if (!ctx->fuse_lin) return 1;
    PoseArg pa{};
    pa.state = ctx->state(slot);
    pa.derive_assoc = params->variant == LILI_VARIANT_FRONTEND ? 0 : 1;
    MatchParams P = to_device_params(params);
    P.no_cost = 1;             // the record stays inside the library, only the GN step is used
    return launch_associate_lin_reduce(ctx, slot, kind_mask, pa, P, ctx->gram_of(slot));
}
static int iterate_impl(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int restart_every, int restart_slot, float* assoc_ms) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(n_iters >= 0, "iterate: negative n_iters");
    ARGCHK(restart_every >= 0 && (restart_every == 0 || (restart_slot >= 0 && restart_slot < LILI_MAX_SLOTS && restart_slot != slot)), "iterate: bad restart arguments");
    struct EventList {      // released on every return path
        std::vector<hipEvent_t> v;
        ~EventList() { for (auto e : v) if (e) (void)hipEventDestroy(e); }
    } evl;
    std::vector<hipEvent_t>& ev = evl.v;
    if (assoc_ms) {
        ev.assign((size_t)2 * n_iters, nullptr);
        for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    }
    unsigned long long wait_key = 0ull;      // != 0: the reduction + GN kernel of the previous iteration publishes its pose under this key (option "overlap_gn")
    const bool want_mirror = ctx->state_mirror_want;      // (lili_pipeline.hip: the pose of the LAST update also into the page-locked mirror; one shot)
    ctx->state_mirror_want = false; ctx->state_mirror_armed = false;
    for (int it = 0; it < n_iters; it++) {   // 3 launches per outer iteration: associate, linearise, reduce+GN
        ctx->state_mirror_armed = want_mirror && it == n_iters - 1;
        if (restart_every > 0 && it % restart_every == 0) { int rc = lili_s2m_pose_copy(ctx, slot, restart_slot); if (rc != LILI_OK) return rc; wait_key = 0ull; }
        if (!assoc_ms) {        // small scans: the whole registration (up to the next restart) as ONE persistent launch
            const int seg = restart_every > 0 ? std::min(restart_every - it % restart_every, n_iters - it) : n_iters - it;
            const int rcp = launch_iterate_persistent(ctx, slot, kind_mask, params, seg);
            if (rcp == LILI_OK) { it += seg - 1; wait_key = 0ull; continue; }
            if (rcp != 1) return rcp;
        }
        if (!assoc_ms) {        // flavours without count scaling: association + linearisation in one launch (2 launches per iteration)
            int rc2 = iterate_fused_lin(ctx, slot, kind_mask, params);
            if (rc2 == LILI_OK) { wait_key = 0ull; continue; }
            if (rc2 != 1) return rc2;
        }
        if (assoc_ms) HIPCHK(hipEventRecord(ev[2 * it], ctx->stream));
        int rc = associate_dev_impl(ctx, slot, kind_mask, params, wait_key);
        if (rc != LILI_OK) return rc;
        if (assoc_ms) HIPCHK(hipEventRecord(ev[2 * it + 1], ctx->stream));
        // Round 5, "overlap_gn": when the NEXT thing on the stream is this slot's next association, the reduction + GN kernel publishes the new pose as keyed granules and
        // that association is launched WITHOUT a barrier against it: its 3 125 waves are dispatched, load their queries and poll for the pose while the one-workgroup
        // reduction still runs — two barriers and one flag hop per iteration instead of three barriers.  Keys never repeat within a context.
        unsigned long long pub_key = 0ull;
        if (ctx->overlap_gn && !assoc_ms && it + 1 < n_iters && !(restart_every > 0 && (it + 1) % restart_every == 0)) pub_key = (++ctx->gn_seq) * 0x9E3779B97F4A7C15ull;
        rc = linearize_dev_impl(ctx, slot, kind_mask, params, ctx->gram_of(slot), 1, 0, &pub_key);
        if (rc != LILI_OK) return rc;
        wait_key = pub_key;
    }
    ctx->state_mirror_armed = false;      // (a path without a reduction + GN kernel has not taken it)
    if (assoc_ms) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        double tot = 0;
        for (int it = 0; it < n_iters; it++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev[2 * it], ev[2 * it + 1])); tot += ms; }
        *assoc_ms = (float)tot;
    }
    return LILI_OK;
}

int lili_s2m_iterate_sharded(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int restart_every,
                             int restart_slot, lili_allreduce_fn allreduce, void* comm, int32_t* d_counts, double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(n_iters >= 0 && params && d_counts && d_gram, "iterate_sharded: bad argument");
    ARGCHK(restart_every >= 0 && (restart_every == 0 || (restart_slot >= 0 && restart_slot < LILI_MAX_SLOTS && restart_slot != slot)), "iterate_sharded: bad restart arguments");
    const bool count_scaled = params->scale_surf_num > 0 || params->scale_edge_num > 0;   // ROT: residual scale = num / GLOBAL count
    // The library's own peer-to-peer exchange (lili_p2p_allreduce) is folded INTO the count kernel and INTO the partial-reduction /
    // Gauss-Newton kernel: 4 launches per iteration (associate, counts + exchange, linearise, reduce + exchange + GN) instead of 7.
    lili_p2p* p2p = (allreduce == &lili_p2p_allreduce && lili_p2p_usable(reinterpret_cast<lili_p2p*>(comm), ctx) && !ctx->no_p2p_fusion) ? reinterpret_cast<lili_p2p*>(comm) : nullptr;
    // a communicator on which an exchange has given up is dead for good (its ranks no longer hold the same pose): refuse, loudly
    if (allreduce == &lili_p2p_allreduce && comm && lili_p2p_status(reinterpret_cast<lili_p2p*>(comm)) != 0)
        return ctx->fail(LILI_E_STATE, "iterate_sharded: the lili_p2p communicator has failed (a peer's record did not arrive within its timeout); SlotState gn_status is 2 on the ranks that noticed");
    for (int it = 0; it < n_iters; it++) {
        int rc;
        if (restart_every > 0 && it % restart_every == 0 && (rc = lili_s2m_pose_copy(ctx, slot, restart_slot)) != LILI_OK) return rc;
        if ((rc = lili_s2m_associate_dev(ctx, slot, kind_mask, params)) != LILI_OK) return rc;
        if (p2p) {
            ARGCHK(slot >= 0 && slot < LILI_MAX_SLOTS && (kind_mask & ~3) == 0 && kind_mask != 0, "iterate_sharded: bad slot / kind mask");
            if (count_scaled) {
                const P2PView v = lili_p2p_next_view(p2p);
                if ((rc = launch_sum_counts(ctx, slot, LILI_MASK_SURF | LILI_MASK_EDGE, d_counts, &v)) != LILI_OK) return rc;
                if ((rc = lili_s2m_counts_import(ctx, slot, d_counts)) != LILI_OK) return rc;
            }
            PoseArg pa{};
            pa.state = ctx->state(slot);
            const MatchParams P = to_device_params(params);
            const P2PView v = lili_p2p_next_view(p2p);
            if ((rc = launch_linearize_reduce(ctx, slot, kind_mask, pa, P, d_gram, 1, &v)) != LILI_OK) return rc;
            ctx->slots[slot].use_global_counts = false;
            continue;
        }
        if (count_scaled) {
            if ((rc = lili_s2m_counts_export(ctx, slot, d_counts)) != LILI_OK) return rc;
            if (allreduce && allreduce(d_counts, d_counts, 2, /*ncclInt32*/ 2, /*ncclSum*/ 0, comm, (void*)ctx->stream) != 0)
                return ctx->fail(LILI_E_HIP, "iterate_sharded: all-reduce of the correspondence counts failed");
            if ((rc = lili_s2m_counts_import(ctx, slot, d_counts)) != LILI_OK) return rc;
        }
        if ((rc = lili_s2m_linearize_dev(ctx, slot, kind_mask, params, d_gram)) != LILI_OK) return rc;
        if (allreduce && allreduce(d_gram, d_gram, LILI_GRAM_DOUBLES, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (void*)ctx->stream) != 0)
            return ctx->fail(LILI_E_HIP, "iterate_sharded: all-reduce of the Gram record failed");
        if ((rc = lili_s2m_gn_update(ctx, slot, d_gram)) != LILI_OK) return rc;
    }
    return LILI_OK;
}

// --------------------------------------------------------------------------------------------
// the sliding window across ranks (BASELINE configs[4])
// --------------------------------------------------------------------------------------------
static int window_args(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, WindowArgs& w, const char* who) {
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, std::string(who) + ": 1..LILI_MAX_SLOTS slots");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0, std::string(who) + ": bad kind mask");
    w = WindowArgs{};
    w.n = n_slots;
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, std::string(who) + ": bad slot");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], std::string(who) + ": duplicate slot");
        Slot& sl = ctx->slots[slots[i]];
        WindowSlot& ws = w.s[i];
        ws.state = ctx->state(slots[i]);
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
            KindSlot& ks = sl.k[kind];
            if (!ks.has_records) return ctx->fail(LILI_E_STATE, std::string(who) + ": associate first");
            if (ks.n_q == 0) continue;
            if (kind == 0) { ws.part_surf = ks.partials.as<double>(); ws.nb_surf = ks.n_lin_blocks; ws.bc_surf = ks.block_counts.as<int>(); ws.nbc_surf = ks.n_assoc_blocks; }
            else { ws.part_edge = ks.partials.as<double>(); ws.nb_edge = ks.n_lin_blocks; ws.bc_edge = ks.block_counts.as<int>(); ws.nbc_edge = ks.n_assoc_blocks; }
        }
    }
    return LILI_OK;
}
// k_linearize of one slot at `pa`, block partials left in the slot's buffers (no reduction): the first half of launch_linearize_reduce
static int launch_linearize_only(lili_ctx* ctx, int slot, int kind_mask, const PoseArg& pa, const MatchParams& P) {
    Slot& s = ctx->slots[slot];
    LinArgs A[2] = {LinArgs{}, LinArgs{}};
    int n_kinds = 0;
    for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
        KindSlot& ks = s.k[kind];
        if (!ks.has_records) return ctx->fail(LILI_E_STATE, "linearize: associate first");
        if (ks.n_q == 0) continue;
        A[kind] = lin_args_of(ctx, slot, kind);
        n_kinds++;
    }
    if (n_kinds == 0) return LILI_OK;
    const int* ng = s.use_global_counts ? s.global_counts : nullptr;
    hipLaunchKernelGGL(k_linearize, dim3(A[0].nb + A[1].nb), dim3(kLinBlock), lds_linearize(kLinBlock), ctx->stream, A[0], A[1], pa, P, ctx->state(slot), ng, FuseTail{});
    HIPCHK(hipGetLastError());
    return LILI_OK;
}
// k_linearize of EVERY slot of a window in ONE launch (k_linearize_window), block partials left in the slots' buffers.  t / q: 3 / 4 doubles per
// slot (host poses), or both null for the slots' device poses.  Bit-identical to launch_linearize_only slot by slot (same bodies, same block geometry).
static int launch_linearize_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q, const MatchParams& P) {
    if (n_slots == 1) {
        PoseArg pa{};
        if (t && q) { for (int k = 0; k < 3; k++) pa.t[k] = t[k]; for (int k = 0; k < 4; k++) pa.q[k] = q[k]; }
        else pa.state = ctx->state(slots[0]);
        return launch_linearize_only(ctx, slots[0], kind_mask, pa, P);
    }
    WinLinArgs W{};
    int nb = 0;
    for (int i = 0; i < n_slots; i++) {
        Slot& s = ctx->slots[slots[i]];
        WinLinSlot& ws = W.s[W.n];
        ws = WinLinSlot{};
        int n_kinds = 0;
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
            KindSlot& ks = s.k[kind];
            if (!ks.has_records) return ctx->fail(LILI_E_STATE, "linearize: associate first");
            if (ks.n_q == 0) continue;
            (kind == 0 ? ws.S : ws.E) = lin_args_of(ctx, slots[i], kind);
            n_kinds++;
        }
        if (n_kinds == 0) continue;             // nothing to linearise for this slot: k_window_reduce sees nb = 0 and writes a zero record
        if (t && q) { for (int k = 0; k < 3; k++) ws.pa.t[k] = t[3 * i + k]; for (int k = 0; k < 4; k++) ws.pa.q[k] = q[4 * i + k]; }
        else ws.pa.state = ctx->state(slots[i]);
        ws.state = ctx->state(slots[i]);
        ws.n_global = s.use_global_counts ? s.global_counts : nullptr;
        ws.first_block = nb;
        nb += ws.S.nb + ws.E.nb;
        W.n++;
    }
    if (nb == 0) return LILI_OK;
    hipLaunchKernelGGL(k_linearize_window, dim3(nb), dim3(kLinBlock), lds_linearize(kLinBlock), ctx->stream, W, P);
    HIPCHK(hipGetLastError());
    return LILI_OK;
}
static bool p2p_comm(lili_ctx* ctx, lili_allreduce_fn allreduce, void* comm, lili_p2p** out) {
    *out = (allreduce == &lili_p2p_allreduce && lili_p2p_usable(reinterpret_cast<lili_p2p*>(comm), ctx) && !ctx->no_p2p_fusion) ? reinterpret_cast<lili_p2p*>(comm) : nullptr;
    return !(allreduce == &lili_p2p_allreduce && comm && lili_p2p_status(reinterpret_cast<lili_p2p*>(comm)) != 0);      // false: the communicator has failed
}

// The correspondence counts of every slot, summed over the ranks in ONE exchange of 2 n int32, into d_counts ([surf, edge] per slot, DEVICE,
// caller-owned) — the ROT residual scale needs the GLOBAL counts, R/src/BackendFusion.cpp:843,861.  The following linearisations of these
// slots (until their next association) scale with d_counts: keep it valid and unchanged.  Async.
int lili_s2m_counts_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, lili_allreduce_fn allreduce, void* comm, int32_t* d_counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(d_counts, "counts_window_sharded: null d_counts");
    HIPCHK(hipSetDevice(ctx->device));
    WindowArgs w;
    int rc = window_args(ctx, slots, n_slots, kind_mask, w, "counts_window_sharded");
    if (rc != LILI_OK) return rc;
    lili_p2p* p2p = nullptr;
    if (!p2p_comm(ctx, allreduce, comm, &p2p)) return ctx->fail(LILI_E_STATE, "counts_window_sharded: the lili_p2p communicator has failed");
    hipLaunchKernelGGL(k_window_counts, dim3(1), dim3(kBlock), 0, ctx->stream, w, d_counts, p2p ? lili_p2p_next_view(p2p) : P2PView{});
    HIPCHK(hipGetLastError());
    if (!p2p && allreduce && allreduce(d_counts, d_counts, (size_t)2 * n_slots, /*ncclInt32*/ 2, /*ncclSum*/ 0, comm, (void*)ctx->stream) != 0)
        return ctx->fail(LILI_E_HIP, "counts_window_sharded: all-reduce of the correspondence counts failed");
    for (int i = 0; i < n_slots; i++) { ctx->slots[slots[i]].global_counts = d_counts + 2 * i; ctx->slots[slots[i]].use_global_counts = true; ctx->slots[slots[i]].sticky_global_counts = true; }
    return LILI_OK;
}

// ONE evaluation of the joint window on the device: every slot linearised at its device pose (this rank's shard of its records), the
// n x LILI_GRAM_DOUBLES records reduced in one launch and summed over the ranks in ONE exchange into d_gram (DEVICE, caller-owned).  Async.
static int linearize_window_impl(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const double* t, const double* q,
                                 lili_allreduce_fn allreduce, void* comm, double* d_gram, int do_gn) {
    WindowArgs w;
    int rc = window_args(ctx, slots, n_slots, kind_mask, w, "linearize_window_sharded");
    if (rc != LILI_OK) return rc;
    lili_p2p* p2p = nullptr;
    if (!p2p_comm(ctx, allreduce, comm, &p2p)) return ctx->fail(LILI_E_STATE, "linearize_window_sharded: the lili_p2p communicator has failed");
    MatchParams P = to_device_params(params);
    if (do_gn) P.no_cost = 1;
    if ((rc = launch_linearize_window(ctx, slots, n_slots, kind_mask, t, q, P)) != LILI_OK) return rc;
    const bool in_kernel = p2p != nullptr || allreduce == nullptr;        // exchange (or none: one rank) and GN inside the reduction launch
    hipLaunchKernelGGL(k_window_reduce, dim3(1), dim3(1024), 0, ctx->stream, w, d_gram, (do_gn && in_kernel) ? 1 : 0, p2p ? lili_p2p_next_view(p2p) : P2PView{});
    HIPCHK(hipGetLastError());
    if (!in_kernel) {
        if (allreduce(d_gram, d_gram, (size_t)LILI_GRAM_DOUBLES * n_slots, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (void*)ctx->stream) != 0)
            return ctx->fail(LILI_E_HIP, "linearize_window_sharded: all-reduce of the Gram records failed");
        if (do_gn) { hipLaunchKernelGGL(k_window_gn, dim3(1), dim3(64), 0, ctx->stream, w, (const double*)d_gram); HIPCHK(hipGetLastError()); }
    }
    for (int i = 0; i < n_slots; i++) if (!ctx->slots[slots[i]].sticky_global_counts) ctx->slots[slots[i]].use_global_counts = false;
    return LILI_OK;
}
int lili_s2m_linearize_window_dev(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params,
                                  lili_allreduce_fn allreduce, void* comm, double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(params && d_gram, "linearize_window_dev: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return linearize_window_impl(ctx, slots, n_slots, kind_mask, params, nullptr, nullptr, allreduce, comm, d_gram, 0);
}
// The same at host-provided poses, blocking, records copied out: what one evaluation of the caller's solver costs on the lidar side when the
// window is sharded over ranks (every rank calls it with the same poses and gets the same bits).
int lili_s2m_linearize_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q, const lili_s2m_params* params,
                                      lili_allreduce_fn allreduce, void* comm, double* d_gram, double* gram, double* cost, int* counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(t && q && params && d_gram && gram, "linearize_window_sharded: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = linearize_window_impl(ctx, slots, n_slots, kind_mask, params, t, q, allreduce, comm, d_gram, 0);
    if (rc != LILI_OK) return rc;
    { const int rc_h = ensure_h_records(ctx); if (rc_h != LILI_OK) return rc_h; }
    HIPCHK(hipMemcpyAsync(ctx->h_records, d_gram, (size_t)n_slots * LILI_GRAM_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_slots; i++) {
        const double* h = ctx->h_records + (size_t)i * LILI_GRAM_DOUBLES;
        std::memcpy(gram + (size_t)64 * i, h, 64 * sizeof(double));
        if (cost) cost[i] = h[64];
        if (counts) { counts[2 * i] = (int)h[65]; counts[2 * i + 1] = (int)h[66]; }
    }
    return LILI_OK;
}
// n_iters outer iterations of every slot of the window, queries sharded over the ranks: per iteration the slots' associations, ONE exchange
// of the 2 n counts (count-scaled flavours only), the slots' linearisations, ONE reduction launch ending with ONE exchange of the n x 72
// doubles and the Gauss-Newton update of every slot — 2 exchanges per iteration whatever the number of keyframes.  (The reference couples
// the keyframes through IMU factors in its own solver; this loop is the lidar-only registration of each keyframe, as lili_s2m_iterate_window.)
int lili_s2m_iterate_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters,
                                    lili_allreduce_fn allreduce, void* comm, int32_t* d_counts, double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(n_iters >= 0 && params && d_counts && d_gram, "iterate_window_sharded: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    const bool count_scaled = params->scale_surf_num > 0 || params->scale_edge_num > 0;
    for (int it = 0; it < n_iters; it++) {
        int rc;
        for (int i = 0; i < n_slots; i++) {
            ARGCHK(slots && slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "iterate_window_sharded: bad slot");
            ctx->slots[slots[i]].sticky_global_counts = false;
            if ((rc = lili_s2m_associate_dev(ctx, slots[i], kind_mask, params)) != LILI_OK) return rc;
        }
        if (count_scaled && (rc = lili_s2m_counts_window_sharded(ctx, slots, n_slots, kind_mask, allreduce, comm, d_counts)) != LILI_OK) return rc;
        if ((rc = linearize_window_impl(ctx, slots, n_slots, kind_mask, params, nullptr, nullptr, allreduce, comm, d_gram, 1)) != LILI_OK) return rc;
    }
    return LILI_OK;
}

// ---- slot-per-rank window (round 5; SURVEY §8e "the three keyframes of the window are also independent"): keyframe i of the window lives on rank
// owner[i] at FULL size — all its queries, no shard — and every evaluation ends with ONE exchange of the n x 72 doubles in which every record has exactly one
// non-zero contributor, i.e. an all-gather carried by the same rank-order sum (x + 0 + ... + 0: the owner's bits, except that a -0.0 entry becomes +0.0).  The ranks
// that do not own a keyframe need neither its queries nor its records, only its pose slot (every rank applies the same Gauss-Newton update to the same record).
// Counts need no exchange: the owner's count IS the global one (ROT residual scale, R/src/BackendFusion.cpp:843,861).  The latency floor of an iteration is the one of a
// single full-size keyframe (the query-sharded modes shrink the association's throughput part but keep its latency chain — DESIGN §5), and K keyframes advance on K ranks.
static int window_gather_impl(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const int* owner, int rank,
                              lili_allreduce_fn allreduce, void* comm, double* d_gram, int do_gn, const double* t = nullptr, const double* q = nullptr /* body poses of ALL slots (else the device poses) */) {
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, "window_gather: 1..LILI_MAX_SLOTS slots");
    ARGCHK((kind_mask & ~3) == 0 && kind_mask != 0 && owner && rank >= 0, "window_gather: bad argument");
    int mine[LILI_MAX_SLOTS], n_mine = 0;
    double t_mine[3 * LILI_MAX_SLOTS], q_mine[4 * LILI_MAX_SLOTS];
    WindowArgs w{};
    w.n = n_slots;
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS && owner[i] >= 0, "window_gather: bad slot / owner");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], "window_gather: duplicate slot");
        w.s[i].state = ctx->state(slots[i]);
        if (owner[i] != rank) continue;             // a zero record from this rank: no partials (k_window_reduce sums nothing)
        if (t && q) { for (int k = 0; k < 3; k++) t_mine[3 * n_mine + k] = t[3 * i + k]; for (int k = 0; k < 4; k++) q_mine[4 * n_mine + k] = q[4 * i + k]; }
        mine[n_mine++] = slots[i];
        Slot& sl = ctx->slots[slots[i]];
        sl.use_global_counts = false; sl.sticky_global_counts = false;
        for (int kind = 0; kind < 2; kind++) if (kind_mask & (1 << kind)) {
            KindSlot& ks = sl.k[kind];
            if (!ks.has_records) return ctx->fail(LILI_E_STATE, "window_gather: associate the owned keyframes first");
            if (ks.n_q == 0) continue;
            if (kind == 0) { w.s[i].part_surf = ks.partials.as<double>(); w.s[i].nb_surf = ks.n_lin_blocks; }
            else { w.s[i].part_edge = ks.partials.as<double>(); w.s[i].nb_edge = ks.n_lin_blocks; }
        }
    }
    lili_p2p* p2p = nullptr;
    if (!p2p_comm(ctx, allreduce, comm, &p2p)) return ctx->fail(LILI_E_STATE, "window_gather: the lili_p2p communicator has failed");
    MatchParams P = to_device_params(params);
    if (do_gn) P.no_cost = 1;
    if (n_mine > 0) { const int rc = launch_linearize_window(ctx, mine, n_mine, kind_mask, t && q ? t_mine : nullptr, t && q ? q_mine : nullptr, P); if (rc != LILI_OK) return rc; }
    const bool in_kernel = p2p != nullptr || allreduce == nullptr;
    hipLaunchKernelGGL(k_window_reduce, dim3(1), dim3(1024), 0, ctx->stream, w, d_gram, (do_gn && in_kernel) ? 1 : 0, p2p ? lili_p2p_next_view(p2p) : P2PView{});
    HIPCHK(hipGetLastError());
    if (!in_kernel) {
        if (allreduce(d_gram, d_gram, (size_t)LILI_GRAM_DOUBLES * n_slots, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (void*)ctx->stream) != 0)
            return ctx->fail(LILI_E_HIP, "window_gather: exchange of the Gram records failed");
        if (do_gn) { hipLaunchKernelGGL(k_window_gn, dim3(1), dim3(64), 0, ctx->stream, w, (const double*)d_gram); HIPCHK(hipGetLastError()); }
    }
    return LILI_OK;
}
int lili_s2m_linearize_window_gather(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const int* owner, int rank,
                                     lili_allreduce_fn allreduce, void* comm, double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(params && d_gram, "linearize_window_gather: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    return window_gather_impl(ctx, slots, n_slots, kind_mask, params, owner, rank, allreduce, comm, d_gram, 0);
}
// The evaluation a host solver asks for (Ceres' Evaluate at ITS parameter values, include/lili_ceres_adapter.h LidarWindowFactor::gather_over_ranks): the owned keyframes
// are linearised at the body poses (t, q) of the call, the records exchanged and copied out — what lili_s2m_linearize_window_sharded is for the query-sharded window.
int lili_s2m_linearize_window_gather_at(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q, const lili_s2m_params* params,
                                        const int* owner, int rank, lili_allreduce_fn allreduce, void* comm, double* d_gram, double* gram, double* cost, int* counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(t && q && params && d_gram && gram, "linearize_window_gather_at: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    int rc = window_gather_impl(ctx, slots, n_slots, kind_mask, params, owner, rank, allreduce, comm, d_gram, 0, t, q);
    if (rc != LILI_OK) return rc;
    { const int rc_h = ensure_h_records(ctx); if (rc_h != LILI_OK) return rc_h; }
    HIPCHK(hipMemcpyAsync(ctx->h_records, d_gram, (size_t)n_slots * LILI_GRAM_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_slots; i++) {
        const double* h = ctx->h_records + (size_t)i * LILI_GRAM_DOUBLES;
        std::memcpy(gram + (size_t)64 * i, h, 64 * sizeof(double));
        if (cost) cost[i] = h[64];
        if (counts) { counts[2 * i] = (int)h[65]; counts[2 * i + 1] = (int)h[66]; }
    }
    return LILI_OK;
}
int lili_s2m_iterate_window_gather(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters, const int* owner, int rank,
                                   lili_allreduce_fn allreduce, void* comm, double* d_gram) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(n_iters >= 0 && params && d_gram && owner && slots, "iterate_window_gather: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    for (int it = 0; it < n_iters; it++) {
        for (int i = 0; i < n_slots; i++) if (owner[i] == rank) {
            ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "iterate_window_gather: bad slot");
            const int rc = lili_s2m_associate_dev(ctx, slots[i], kind_mask, params);
            if (rc != LILI_OK) return rc;
        }
        const int rc = window_gather_impl(ctx, slots, n_slots, kind_mask, params, owner, rank, allreduce, comm, d_gram, 1);
        if (rc != LILI_OK) return rc;
    }
    return LILI_OK;
}

// Several independent registrations (the keyframes of one sliding window, L/src/BackendFusion.cpp:843-1007 loops over
// them per outer iteration; or several sensors) advanced concurrently: slot i runs its own associate / linearise /
// reduce+GN chain on its own stream, forked from and joined to the context's stream with events.  The association
// kernel fills the chip by itself, but the latency-bound linearise / reduce launches of one slot overlap the
// association of another.  Results are identical to calling lili_s2m_iterate slot after slot.
int lili_s2m_iterate_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS, "iterate_window: 1..LILI_MAX_SLOTS slots");
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "iterate_window: bad slot");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], "iterate_window: duplicate slot");
    }
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));
    hipStream_t main_stream = ctx->stream;
    int rc = LILI_OK;
    ctx->persistent_off_now = n_slots > 4;       // more than four persistent launches side by side could not all be resident
    for (int i = 0; i < n_slots && rc == LILI_OK; i++) {
        if (i > 0) {
            if (!ctx->side[i]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
            if (!ctx->join_ev[i]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming));
            HIPCHK(hipStreamWaitEvent(ctx->side[i], ctx->fork_ev, 0));
            ctx->stream = ctx->side[i];          // the launch helpers enqueue on ctx->stream (one thread per context)
        }
        rc = iterate_impl(ctx, slots[i], kind_mask, params, n_iters, 0, 0, nullptr);
        if (i + 1 == n_slots || rc != LILI_OK) ctx->persistent_off_now = false;
        if (i > 0) {
            hipError_t e = hipEventRecord(ctx->join_ev[i], ctx->side[i]);
            ctx->stream = main_stream;
            if (e != hipSuccess) return ctx->fail(LILI_E_HIP, "iterate_window: hipEventRecord failed");
            HIPCHK(hipStreamWaitEvent(main_stream, ctx->join_ev[i], 0));
        }
    }
    ctx->stream = main_stream;
    return rc;
}

int lili_s2m_iterate(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters) {
    return iterate_impl(ctx, slot, kind_mask, params, n_iters, 0, 0, nullptr);
}

int lili_s2m_iterate_restart(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int restart_every, int restart_slot, float* assoc_ms) {
    return iterate_impl(ctx, slot, kind_mask, params, n_iters, restart_every, restart_slot, assoc_ms);
}

// host mirror of k_gn_update (used by the ceres adapter / host LM; plain C++ on the host by design —
// a 6x6 solve is O(1) work outside the data-parallel path)
int lili_gn_step_host(const double gram[64], double t[3], double q[4], double delta[6]) {
    if (!gram || !t || !q) return LILI_E_ARG;
    double Pm[7][6] = {};
    Pm[0][0] = Pm[1][1] = Pm[2][2] = 1.0;
    const double x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
    const double tab[4][3] = {{-x1, -x2, -x3}, {x0, x3, -x2}, {-x3, x0, x1}, {x2, -x1, x0}};
    for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) Pm[3 + r][3 + c] = tab[r][c];
    double H[6][6], g[6];
    for (int a = 0; a < 6; a++) {
        for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int i = 0; i < 7; i++) { double gi = 0; for (int j = 0; j < 7; j++) gi += gram[i * 8 + j] * Pm[j][b]; s += Pm[i][a] * gi; }
            H[a][b] = s;
        }
        double s = 0; for (int i = 0; i < 7; i++) s += Pm[i][a] * gram[i * 8 + 7];
        g[a] = -s;
    }
    // LDL^T with reciprocal pivots, the same operation order as gn_update_block on the device
    double W[6][6], dinv[6];
    for (int j = 0; j < 6; j++) {
        double dj = H[j][j];
        for (int k = 0; k < j; k++) dj -= H[j][k] * W[j][k];
        if (!(dj > 0)) return 1;
        dinv[j] = 1.0 / dj;
        for (int i = j + 1; i < 6; i++) {
            double sv = H[i][j];
            for (int k = 0; k < j; k++) sv -= H[i][k] * W[j][k];
            W[i][j] = sv; H[i][j] = sv * dinv[j];
        }
    }
    double d[6];
    for (int i = 0; i < 6; i++) { double sv = g[i]; for (int k = 0; k < i; k++) sv -= H[i][k] * d[k]; d[i] = sv; }
    for (int i = 0; i < 6; i++) d[i] = d[i] * dinv[i];
    for (int i = 5; i >= 0; i--) { double sv = d[i]; for (int k = i + 1; k < 6; k++) sv -= H[k][i] * d[k]; d[i] = sv; }
    for (int i = 0; i < 6; i++) if (!(d[i] == d[i])) return 1;
    if (delta) for (int i = 0; i < 6; i++) delta[i] = d[i];
    t[0] += d[0]; t[1] += d[1]; t[2] += d[2];
    double nd = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (nd > 0.0) {
        double sbd = std::sin(nd) / nd, cw = std::cos(nd);
        double a[4] = {cw, sbd * d[3], sbd * d[4], sbd * d[5]};
        double r[4] = {a[0] * x0 - a[1] * x1 - a[2] * x2 - a[3] * x3, a[0] * x1 + a[1] * x0 + a[2] * x3 - a[3] * x2,
                       a[0] * x2 + a[2] * x0 + a[3] * x1 - a[1] * x3, a[0] * x3 + a[3] * x0 + a[1] * x2 - a[2] * x1};
        for (int i = 0; i < 4; i++) q[i] = r[i];
    }
    return 0;
}


// Square-root form of a Gram for the ceres adapter: returns a 9-residual block (residuals[9], jacobian 9x7
// row-major) with  J~^T J~ = G[0:7,0:7],  J~^T r~ = G[0:7,7],  |r~|^2 = 2*cost  (the 9th residual pads the cost
// to sum 1/2 rho(s_i); its Jacobian row is zero).  Uses a symmetric Jacobi eigen-decomposition G = V L V^T
// (rank-deficient Grams are fine), [J~ r~] = L^1/2 V^T.  Host-side O(1) algebra by design.
int lili_gram_to_factor(const double gram[64], double cost, double residuals[9], double jacobian[63]) {
    if (!gram || !residuals || !jacobian) return LILI_E_ARG;
    double A[8][8], V[8][8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) { A[i][j] = 0.5 * (gram[i * 8 + j] + gram[j * 8 + i]); V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < 8; i++) { diag += std::fabs(A[i][i]); for (int j = i + 1; j < 8; j++) off += std::fabs(A[i][j]); }
        if (!(off > 1e-300) || off <= 1e-17 * diag) break;
        for (int p = 0; p < 7; p++) for (int q = p + 1; q < 8; q++) {
            if (A[p][q] == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            double tt = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            if (theta < 0) tt = -tt;
            double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
            for (int k = 0; k < 8; k++) { double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
            for (int k = 0; k < 8; k++) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
            for (int k = 0; k < 8; k++) { double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
        }
    }
    for (int r = 0; r < 8; r++) {
        double lam = A[r][r] > 0 ? std::sqrt(A[r][r]) : 0.0;   // tiny negative eigenvalues are rounding noise
        for (int c = 0; c < 7; c++) jacobian[r * 7 + c] = lam * V[c][r];
        residuals[r] = lam * V[7][r];
    }
    for (int c = 0; c < 7; c++) jacobian[8 * 7 + c] = 0.0;
    double pad = 2.0 * cost - gram[63];
    residuals[8] = pad > 0 ? std::sqrt(pad) : 0.0;   // rho concave => sum rho(s) >= sum rho'(s) s, so pad >= 0 up to rounding
    return LILI_OK;
}

}  // extern "C"