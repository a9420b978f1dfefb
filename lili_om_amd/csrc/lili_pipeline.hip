// Front-end frame pipeline: the per-scan chain of the reference's odometry node as ONE C-ABI call whose data never leaves HBM.
//
// Reference behaviour replaced (L/ = LiLi-OM/):
//   Preprocessing::cloudHandler            L/src/Preprocessing.cpp:219-401     (lili_extract_livox)
//   LidarOdometry::run                     L/src/LidarOdometry.cpp:652-707
//     buildLocalMap / downSampleCloud      L:280-303, 314-322                  (keyframe ring + VoxelGrid(0.4) + setInputCloud, lili_voxel.hip)
//     updateTransformationWithCeres        L:483-561                           (outer iterations of the front-end flavour, lili_s2m*.hip)
//     savePoses + transformCloud           L:324-350, 292-297                  (the frame joins the ring at the pose found)
// The stages are the library's own entry points; what this file adds is the ORDER (the next frame's local map is built behind this frame's
// iterations, the ring push reads the pose on the device) and the absence of host copies between them.
#include <chrono>
#include <functional>

#include "lili_ctx.h"

namespace lili {

// lili_s2m_pose_set without the host staging copy and its synchronisation: the pose travels in the kernel-argument segment
struct PoseVal { double v[7]; };
__global__ void k_pose_set(SlotState* __restrict__ s, PoseVal p) {
    const int i = threadIdx.x;
    if (i < 7) s->pose[i] = p.v[i];
    if (i < 6) s->last_delta[i] = 0.0;
    if (i == 7) { s->n_res[0] = 0; s->n_res[1] = 0; s->gn_status = 0; s->iters = 0; s->cnt_word = 0ull; }
}

// A frame's queries (device float4 rows) into the matcher slot's arrays and the slot's pose in ONE launch (it was a conversion launch per kind and one for the pose)
__global__ __launch_bounds__(256) void k_frame_queries(const float4* __restrict__ src_s, int n_s, float4* __restrict__ out_s, const float4* __restrict__ src_e, int n_e,
                                                       float4* __restrict__ out_e, SlotState* __restrict__ s, PoseVal p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_s) out_s[i] = src_s[i];
    else if (i - n_s < n_e) out_e[i - n_s] = src_e[i - n_s];
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 7) s->pose[t] = p.v[t];
        if (t < 6) s->last_delta[t] = 0.0;
        if (t == 7) { s->n_res[0] = 0; s->n_res[1] = 0; s->gn_status = 0; s->iters = 0; s->cnt_word = 0ull; }
    }
}

}  // namespace lili

extern "C" {

int lili_frontend_reset(lili_ctx* ctx) {
    if (!ctx) return LILI_E_ARG;
    ctx->frontend_commit_pending = false;
    return lili_localmap_reset(ctx, LILI_KIND_SURF);
}

// Builds the local map of the keyframes pushed so far if the last lili_frontend_frame left that to the next one (it does: the commit runs under the next frame's
// extraction).  A caller that wants the map between frames (lili_localmap_get, lili_map_info) calls this first.  Blocking.
int lili_frontend_flush(lili_ctx* ctx, const lili_s2m_params* match, const lili_frontend_options* opt, int32_t* n_map_raw, int32_t* n_map) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(match && opt && opt->leaf_map > 0, "frontend_flush: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->frontend_commit_pending) return LILI_OK;
    int64_t a = 0, b = 0;
    const int rc = lili_localmap_commit(ctx, LILI_KIND_SURF, opt->leaf_map, match->kd_max_radius, &a, &b);
    if (rc != LILI_OK) return rc;
    ctx->frontend_commit_pending = false;
    ctx->frontend_map_raw = (int32_t)a;
    if (n_map_raw) *n_map_raw = (int32_t)a;
    if (n_map) *n_map = (int32_t)b;
    return LILI_OK;
}

}  // extern "C"

// "Iterate, then read the pose" with ONE synchronisation and no copy launch: the reduction + GN kernel of the last iteration writes pose and status into a page-locked mirror
// of the slot's state as well (gn_update_block).  arm_pose_mirror before lili_s2m_iterate; read_pose_after_iterate synchronises and takes the mirror — or, if the
// iteration ran a path without that kernel (the sentinel is still there), reads the state back the plain way.
static int arm_pose_mirror(lili_ctx* ctx, int slot) {
    if (!ctx->h_state_mirror) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_state_mirror), sizeof(SlotState) * LILI_MAX_SLOTS, hipHostMallocDefault));
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, ctx->h_state_mirror, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
        ctx->h_state_mirror_dev = static_cast<SlotState*>(d);
    }
    *reinterpret_cast<volatile int*>(&ctx->h_state_mirror[slot].gn_status) = -1;
    ctx->state_mirror_want = ctx->h_state_mirror_dev != nullptr;
    return LILI_OK;
}
static int read_pose_after_iterate(lili_ctx* ctx, int slot, SlotState* st) {
    ctx->state_mirror_want = false;
    int rc = lili_readback_finish(ctx);
    if (rc != LILI_OK) return rc;
    const int status = ctx->h_state_mirror ? *reinterpret_cast<volatile int*>(&ctx->h_state_mirror[slot].gn_status) : -1;
    if (status != -1) { for (int i = 0; i < 7; i++) st->pose[i] = ctx->h_state_mirror[slot].pose[i]; st->gn_status = status; return LILI_OK; }
    rc = lili_readback_add(ctx, st, ctx->state(slot), sizeof(*st));
    if (rc != LILI_OK) return rc;
    return lili_readback_finish(ctx);
}

// What a frame needs from its extractor: the kernels enqueued (no synchronisation), the counts taken afterwards, the feature lists as device clouds.
struct FrameExtractor {
    std::function<int()> enqueue, complete;
    std::function<int(lili_cloud* edge, lili_cloud* surf)> lists;
    // optional (frame_guess_counts): the lists before their lengths are known — device words that will hold the lengths, the previous scan's lengths as guesses; and whether
    // `complete` rewrote the lists (second passes of the extractor)
    std::function<void(int* prev_edge, int* prev_surf)> prev;
    std::function<int(const lili_query_sink*)> enqueue_sink;      // the extraction enqueued with the matcher slot as a second destination of its lists
    std::function<bool()> redone;
};
static int frame_impl(lili_ctx* ctx, const FrameExtractor& ex, const lili_s2m_params* match, const lili_frontend_options* opt, const double t_pred[3], const double q_pred[4],
                      lili_frontend_result* res) {
    ARGCHK(match && opt && t_pred && q_pred && res, "frontend_frame: null argument");
    ARGCHK(opt->leaf_query >= 0 && opt->leaf_map > 0 && opt->width >= 1 && opt->n_iters >= 0 && opt->slot >= 0 && opt->slot < LILI_MAX_SLOTS, "frontend_frame: bad options");
    const bool ext_map = (opt->flags & LILI_FRAME_EXTERNAL_MAP) != 0, edges = (opt->flags & LILI_FRAME_EDGES) != 0;
    ARGCHK(!edges || ext_map, "frontend_frame: LILI_FRAME_EDGES needs LILI_FRAME_EXTERNAL_MAP (the front end keeps a surf ring only)");
    ARGCHK(!(ext_map && (opt->flags & LILI_FRAME_SELF_MAP)), "frontend_frame: LILI_FRAME_EXTERNAL_MAP and LILI_FRAME_SELF_MAP exclude each other");
    ARGCHK(opt->leaf_query > 0 || ext_map, "frontend_frame: leaf_query = 0 (the features themselves are the queries) only with LILI_FRAME_EXTERNAL_MAP");
    HIPCHK(hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](int k) { if (opt->want_timing) res->stage_us[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
    *res = lili_frontend_result{};
    const int slot = opt->slot;
    // ---- extraction (Preprocessing::cloudHandler), enqueued only: features stay in the extractor's device lists.  Under its kernels the host builds the local map the
    //      previous frame left pending (buildLocalMap + downSampleCloud + setInputCloud, L:280-318, 490).  That commit synchronises twice anyway: its first read-back
    //      delivers the extraction's counts with its own; behind the index build's kernels — before the build's read-back synchronises — the frame's QUERY filter is
    //      enqueued (down_size_filter_surf, L:320-322, on the extractor's device list, into the filter's second output) so that its voxel count comes back with the
    //      build's density words.  Three synchronisations per frame (ring merge, index build + query filter, pose) where the stages called one by one take five.
    const bool self_map = (opt->flags & LILI_FRAME_SELF_MAP) != 0;
    lili_cloud d_edge{}, d_surf{};
    bool lists = false, filter_enqueued = false, filter_pending = false;
    int rc = LILI_OK;
    // ---- A caller's maps and the features themselves as queries (BASELINE configs[0]): nothing between the extraction and the matcher needs the host — except the NUMBER
    //      of features, which sizes the matcher's launches.  Guess it (the previous scan's counts + 1/8 + 64): the slot is sized for that many rows, the extractor's last
    //      kernel writes its lists into the slot's query arrays as well, fills the rows behind them with NaN (such a row selects nothing, so every sum is the one an exactly
    //      sized slot takes: lili_s2m_set_queries_counted) and sets the slot's pose; the matcher is enqueued right behind it and the call synchronises ONCE — the pose comes
    //      back with the counts.  A scan with more features than guessed, or one whose lists a second pass of the extractor rewrote, is matched again below the plain way.
    bool guessed = false;
    int ge = 0, gs = 0;
    if (ctx->frame_guess_counts && ctx->rot_fold && ext_map && !(opt->leaf_query > 0) && ex.prev && ex.enqueue_sink && opt->n_iters > 0 && ctx->map[LILI_KIND_SURF].valid &&
        ctx->map[LILI_KIND_SURF].n >= 10 && (!edges || ctx->map[LILI_KIND_EDGE].valid)) {
        int pe = 0, ps = 0;
        ex.prev(&pe, &ps);
        if (ps > 0) {
            ge = ((pe + pe / 8 + 64 + 63) / 64) * 64; gs = ((ps + ps / 8 + 64 + 63) / 64) * 64;
            rc = lili_s2m_set_queries_counted(ctx, slot, LILI_KIND_SURF, gs);
            if (rc == LILI_OK && edges) rc = lili_s2m_set_queries_counted(ctx, slot, LILI_KIND_EDGE, ge);
            if (rc != LILI_OK) return rc;
            lili_query_sink sink{};
            sink.q_surf = ctx->slots[slot].k[LILI_KIND_SURF].q.as<float4>(); sink.cap_surf = gs;
            if (edges) { sink.q_edge = ctx->slots[slot].k[LILI_KIND_EDGE].q.as<float4>(); sink.cap_edge = ge; }
            sink.state = ctx->state(slot);
            for (int i = 0; i < 3; i++) sink.pose[i] = t_pred[i];
            for (int i = 0; i < 4; i++) sink.pose[3 + i] = q_pred[i];
            rc = ex.enqueue_sink(&sink);      // (1: not applicable — an empty scan —: the plain chain)
            if (rc != LILI_OK && rc != 1) return rc;
            guessed = rc == LILI_OK;
        }
    }
    if (!guessed) {
        // (the local map of the previous frame is still to be built: its ring merge and this extraction share nothing — the extraction takes a stream of its own)
        ctx->extract_side_next = ctx->frame_extract_stream && ctx->frontend_commit_pending && !((opt->flags & LILI_FRAME_SELF_MAP) != 0) && !ext_map;
        rc = ex.enqueue();
        ctx->extract_side_next = false;
        if (rc != LILI_OK) return rc;
    }
    if (guessed) {
        ctx->slots[slot].assoc_since_pose = 0;
        rc = arm_pose_mirror(ctx, slot);
        if (rc != LILI_OK) return rc;
        rc = lili_s2m_iterate(ctx, slot, LILI_MASK_SURF | (edges ? LILI_MASK_EDGE : 0), match, opt->n_iters);
        if (rc != LILI_OK) return rc;
        stamp(0); stamp(1); stamp(2);
        SlotState st{};
        rc = read_pose_after_iterate(ctx, slot, &st);
        if (rc != LILI_OK) return rc;
        rc = ex.complete();                  // (the stream has been synchronised: the counts lie in the extractor's page-locked state)
        if (rc != LILI_OK) return rc;
        rc = ex.lists(&d_edge, &d_surf);
        if (rc != LILI_OK) return rc;
        lists = true;
        if (!(ex.redone && ex.redone()) && d_surf.n > 0 && (int64_t)d_surf.n <= gs && (!edges || (int64_t)d_edge.n <= ge)) {
            rc = lili_s2m_trim_queries(ctx, slot, LILI_KIND_SURF, (int)d_surf.n);
            if (rc == LILI_OK && edges) rc = lili_s2m_trim_queries(ctx, slot, LILI_KIND_EDGE, (int)d_edge.n);
            if (rc != LILI_OK) return rc;
            stamp(3);
            res->n_edge = (int32_t)d_edge.n; res->n_surf = (int32_t)d_surf.n; res->n_query = (int32_t)d_surf.n;
            res->matched = 1;
            for (int i = 0; i < 3; i++) res->t[i] = st.pose[i];
            for (int i = 0; i < 4; i++) res->q[i] = st.pose[3 + i];
            res->gn_status = st.gn_status;
            res->n_map = (int32_t)ctx->map[LILI_KIND_SURF].n; res->n_map_raw = res->n_map;
            return LILI_OK;
        }
        ctx->frame_guess_misses++;
        filter_enqueued = true;              // the lists are known; the features themselves are the queries: on with the plain chain
    }
    // extraction's counts -> device lists -> query filter enqueued (no synchronisation of its own unless the counts have not come back yet)
    bool in_hook = false;      // called from inside the index build: its scratch fill has just zeroed the filter's box words
    auto enqueue_query_filter = [&]() -> int {
        int r = ex.complete();
        if (r != LILI_OK) return r;
        r = ex.lists(&d_edge, &d_surf);
        if (r != LILI_OK) return r;
        lists = true;
        if (!(opt->leaf_query > 0)) { filter_enqueued = true; return LILI_OK; }      // the features themselves are the queries
        r = lili_voxel_filter_dev_enqueue(ctx, static_cast<const float4*>(d_surf.data), (int)d_surf.n, opt->leaf_query, in_hook, &filter_pending);
        filter_enqueued = r == LILI_OK;
        return r;
    };
    int64_t n_raw = 0, n_map = 0;
    if (ctx->frontend_commit_pending && !self_map && !ext_map) {
        ctx->pre_sync_hook = [&]() -> int { in_hook = ctx->hook_box_words_zero; const int r = enqueue_query_filter(); in_hook = false; return r; };
        rc = lili_localmap_commit(ctx, LILI_KIND_SURF, opt->leaf_map, match->kd_max_radius, &n_raw, &n_map);
        ctx->pre_sync_hook = nullptr;      // (a commit that built no index has not called it)
        if (rc != LILI_OK) return rc;
        ctx->frontend_commit_pending = false;
    }
    stamp(0);
    if (!filter_enqueued) {
        rc = enqueue_query_filter();
        if (rc != LILI_OK) return rc;
        if (filter_pending) { rc = lili_readback_finish(ctx); if (rc != LILI_OK) return rc; }
    }
    res->n_edge = (int32_t)d_edge.n; res->n_surf = (int32_t)d_surf.n;
    // ---- the centroids become the frame's queries AND its keyframe
    const float4* d_q = static_cast<const float4*>(d_surf.data); int n_q = (int)d_surf.n;
    if (opt->leaf_query > 0) {
        rc = lili_voxel_filter_dev_complete(ctx, &d_q, &n_q);
        if (rc != LILI_OK) return rc;
    }
    res->n_query = n_q;
    stamp(1);
    if (self_map) {
        // buildLocalMap's initialisation branch (L:283-289): the map is this frame's own surf features; downSampleCloud filters them with the MAP filter (L:314-318)
        const float4* d_m = d_q; int n_m = n_q;
        if (opt->leaf_map != opt->leaf_query) {
            rc = lili_voxel_filter_dev(ctx, static_cast<const float4*>(d_surf.data), (int)d_surf.n, opt->leaf_map, &d_m, &n_m);      // (the filter's first output: the queries stay)
            if (rc != LILI_OK) return rc;
        }
        const lili_cloud mc{d_m, (size_t)n_m, 16, 12, LILI_MEM_DEVICE};
        const bool srows = ctx->super_rows;
        if (!ctx->localmap_super_rows && n_m < 400000 && !(ctx->focus_radius > 0)) ctx->super_rows = false;      // as lili_localmap_commit: small maps go without the 9x copy
        rc = lili_map_set(ctx, LILI_KIND_SURF, &mc, match->kd_max_radius);
        ctx->super_rows = srows;
        if (rc != LILI_OK) return rc;
    }
    const int mask = LILI_MASK_SURF | (edges ? LILI_MASK_EDGE : 0);
    PoseVal pv{};
    for (int i = 0; i < 3; i++) pv.v[i] = t_pred[i];
    for (int i = 0; i < 4; i++) pv.v[3 + i] = q_pred[i];
    const int n_e = edges ? (int)d_edge.n : 0;
    if (n_q > 0 && (!edges || n_e > 0)) {      // the slot sized (lili_s2m_set_queries' bookkeeping), then queries (the filter's second output is rewritten by the next frame's filter) and pose in one launch
        rc = lili_s2m_set_queries_counted(ctx, slot, LILI_KIND_SURF, n_q);
        if (rc == LILI_OK && edges) rc = lili_s2m_set_queries_counted(ctx, slot, LILI_KIND_EDGE, n_e);
        if (rc != LILI_OK) return rc;
        hipLaunchKernelGGL(k_frame_queries, dim3(nblocks(n_q + n_e, 256)), dim3(256), 0, ctx->stream, d_q, n_q, ctx->slots[slot].k[LILI_KIND_SURF].q.as<float4>(),
                           static_cast<const float4*>(d_edge.data), n_e, edges ? ctx->slots[slot].k[LILI_KIND_EDGE].q.as<float4>() : nullptr, ctx->state(slot), pv);
    } else {
        const lili_cloud qc{d_q, (size_t)n_q, 16, 12, LILI_MEM_DEVICE};
        rc = lili_s2m_set_queries(ctx, slot, LILI_KIND_SURF, &qc);
        if (rc != LILI_OK) return rc;
        if (edges) { rc = lili_s2m_set_queries(ctx, slot, LILI_KIND_EDGE, &d_edge); if (rc != LILI_OK) return rc; }
        hipLaunchKernelGGL(k_pose_set, dim3(1), dim3(8), 0, ctx->stream, ctx->state(slot), pv);
    }
    HIPCHK(hipGetLastError());
    ctx->slots[slot].assoc_since_pose = 0;
    // ---- updateTransformationWithCeres (L:483-561): needs a map of at least 10 points (L:485-488); the first frame of a sequence has none
    const bool matched = (self_map || ext_map || lili_localmap_ring_size(ctx, LILI_KIND_SURF) > 0) && ctx->map[LILI_KIND_SURF].valid && ctx->map[LILI_KIND_SURF].n >= 10 && n_q > 0 &&
                         opt->n_iters > 0 && (!edges || ctx->map[LILI_KIND_EDGE].valid);
    if (matched) {
        rc = arm_pose_mirror(ctx, slot);
        if (rc != LILI_OK) return rc;
        rc = lili_s2m_iterate(ctx, slot, mask, match, opt->n_iters);
        if (rc != LILI_OK) return rc;
    }
    res->matched = matched ? 1 : 0;
    stamp(2);
    // ---- ONE synchronisation: the pose.  The frame then joins the ring at that pose (read by the push kernel from the slot's device state) BEHIND the read-back — the
    //      caller has its pose while the push is still on the stream; the local map with the new keyframe is built at the start of the next frame (or by
    //      lili_frontend_flush).
    SlotState st{};
    if (matched) rc = read_pose_after_iterate(ctx, slot, &st);
    else { rc = lili_readback_add(ctx, &st, ctx->state(slot), sizeof(st)); if (rc == LILI_OK) rc = lili_readback_finish(ctx); }
    if (rc != LILI_OK) return rc;
    if (!ext_map) {
        rc = lili_localmap_push_dev(ctx, LILI_KIND_SURF, ctx->slots[slot].k[LILI_KIND_SURF].q.as<float4>(), (opt->flags & LILI_FRAME_PUSH_EMPTY) ? 0 : n_q, ctx->state(slot), opt->width);
        if (rc != LILI_OK) return rc;
        ctx->frontend_commit_pending = true;
    }
    stamp(3);
    for (int i = 0; i < 3; i++) res->t[i] = st.pose[i];
    for (int i = 0; i < 4; i++) res->q[i] = st.pose[3 + i];
    res->gn_status = matched ? st.gn_status : 0;
    // the map the frame was matched against (also when it was not built in this call — a self-map frame, a caller's map, a map built by lili_frontend_flush)
    res->n_map = matched ? (int32_t)ctx->map[LILI_KIND_SURF].n : (int32_t)n_map;
    if (n_raw) ctx->frontend_map_raw = (int32_t)n_raw; else if (self_map) ctx->frontend_map_raw = (int32_t)d_surf.n;
    res->n_map_raw = ext_map ? res->n_map : ctx->frontend_map_raw;
    return LILI_OK;
}

extern "C" {

int lili_frontend_frame(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* livox,
                        const lili_s2m_params* match, const lili_frontend_options* opt, const double t_pred[3], const double q_pred[4],
                        lili_frontend_result* res) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(scan && q_imu && livox, "frontend_frame: null argument");
    FrameExtractor ex;
    ex.enqueue = [&]() { return lili_extract_livox_enqueue(ctx, scan, curvature_offset, q_imu, livox); };
    ex.complete = [&]() { return lili_extract_livox_complete(ctx); };
    const bool want_edges = opt && (opt->flags & LILI_FRAME_EDGES) != 0;
    ex.lists = [&](lili_cloud* e, lili_cloud* s) { return lili_extract_livox_device_ex(ctx, e, s, want_edges); };
    return frame_impl(ctx, ex, match, opt, t_pred, q_pred, res);
}

// The same chain behind the LOAM-style extractor of LiLi-OM-ROT (R/src/Preprocessing.cpp:248-535 in front of R/src/LidarOdometry.cpp:638-693 — the ROT package's odometry
// node is the Livox package's but for the point type: no normals to rotate in transformCloud).
int lili_frontend_frame_rot(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* rot,
                            const lili_s2m_params* match, const lili_frontend_options* opt, const double t_pred[3], const double q_pred[4],
                            lili_frontend_result* res) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(scan && q_imu && q_lb && rot, "frontend_frame_rot: null argument");
    FrameExtractor ex;
    ex.enqueue = [&]() { return lili_extract_rot_enqueue(ctx, scan, q_imu, q_lb, rot); };
    ex.complete = [&]() { return lili_extract_rot_complete(ctx); };
    ex.lists = [&](lili_cloud* e, lili_cloud* s) { return lili_extract_rot_device(ctx, nullptr, e, s); };
    ex.prev = [&](int* pe, int* ps) { lili_extract_rot_prev(ctx, pe, ps); };
    ex.enqueue_sink = [&](const lili_query_sink* sink) { return scan->n == 0 ? 1 : lili_extract_rot_enqueue(ctx, scan, q_imu, q_lb, rot, sink); };
    ex.redone = [&]() { return lili_extract_rot_redone(ctx); };
    return frame_impl(ctx, ex, match, opt, t_pred, q_pred, res);
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// Back-end keyframe preparation (SURVEY §8 f-1 + a-13 / a-14 as ONE call): what BackendFusion does between "a new keyframe has arrived" and
// "ceres::Solve" (L/src/BackendFusion.cpp:830-980 with buildLocalMapWithLandMark :1387-1484 and downSampleCloud :1486-1528):
//   1. the keyframe whose pose the last solve fixed joins both local-map rings at that pose (recent_{surf,edge}_keyframes.push_back(transformCloud(...)), the oldest
//      leaves beyond local_map_width),
//   2. both rings -> VoxelGrid(surf_ds / edge_ds) -> kd_tree_*_local_map->setInputCloud (:1488-1492, 839-840): lili_localmap_commit per kind,
//   3. the NEW keyframe's features -> ds_filter_surf / ds_filter_edge (:1505-1519) -> the queries of the window's newest slot,
//   4. findCorrespondingCornerFeatures / findCorrespondingSurfFeatures of every keyframe of the window at its association pose (:919-936).
// All of it on the device: the features arrive once (host or device clouds), rings, maps, indices, down-sampled queries and correspondence records stay in HBM.  The
// two feature filters ride on the index builds of the two commits (behind their kernels, before their read-backs: lili_ctx::pre_sync_hook), so a keyframe costs the
// synchronisations of the two commits and the one of the association — five — where the calls one by one take eleven and move every cloud through host buffers.
// ------------------------------------------------------------------------------------------------------------------------------------------------
int lili_backend_keyframe_prepare(lili_ctx* ctx, const lili_cloud* join_surf, const lili_cloud* join_edge, const double t_join[3], const double q_join[4],
                                  const lili_cloud* new_surf, const lili_cloud* new_edge, const int* slots, int n_slots, const double* t_assoc, const double* q_assoc,
                                  const lili_s2m_params* match, const lili_backend_options* opt, int32_t* n_res, lili_backend_result* res) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(new_surf && new_edge && slots && n_slots >= 1 && n_slots <= LILI_MAX_SLOTS && t_assoc && q_assoc && match && opt, "backend_keyframe_prepare: null argument");
    ARGCHK(opt->leaf_surf > 0 && opt->leaf_edge > 0 && opt->leaf_surf_map > 0 && opt->leaf_edge_map > 0 && opt->width >= 1, "backend_keyframe_prepare: bad options");
    ARGCHK((join_surf == nullptr) == (join_edge == nullptr) && (!join_surf || (t_join && q_join)), "backend_keyframe_prepare: the joining keyframe needs both feature kinds and its pose");
    ARGCHK(opt->join_slot < LILI_MAX_SLOTS && (opt->join_slot < 0 || (!join_surf && t_join && q_join)), "backend_keyframe_prepare: join_slot excludes join_surf / join_edge and needs the pose");
    for (int i = 0; i < n_slots; i++) {
        ARGCHK(slots[i] >= 0 && slots[i] < LILI_MAX_SLOTS, "backend_keyframe_prepare: bad slot");
        for (int k = 0; k < i; k++) ARGCHK(slots[k] != slots[i], "backend_keyframe_prepare: duplicate slot");
    }
    HIPCHK(hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    lili_backend_result R{};
    auto stamp = [&](int k) { if (opt->want_timing) R.stage_us[k] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
    int rc;
    // 1. the keyframe of the previous solve joins the rings (asynchronous: ingestion + transform on the stream)
    if (join_surf) {
        if ((rc = lili_localmap_push(ctx, LILI_KIND_SURF, join_surf, t_join, q_join, opt->width)) != LILI_OK) return rc;
        if ((rc = lili_localmap_push(ctx, LILI_KIND_EDGE, join_edge, t_join, q_join, opt->width)) != LILI_OK) return rc;
    } else if (opt->join_slot >= 0) {      // the down-sampled features an earlier call left in that slot (device to device)
        for (int kind = 0; kind < 2; kind++) {
            const KindSlot& ks = ctx->slots[opt->join_slot].k[kind];
            if (!ks.has_queries) return ctx->fail(LILI_E_STATE, "backend_keyframe_prepare: join_slot holds no queries");
            const lili_cloud jc{ks.q.p, (size_t)ks.n_q, 16, 12, LILI_MEM_DEVICE};
            if ((rc = lili_localmap_push(ctx, kind, &jc, t_join, q_join, opt->width)) != LILI_OK) return rc;
        }
    }
    // the new keyframe's features as device float4 rows (a device cloud in that layout is read where it lies)
    const lili_cloud* feats[2] = {new_surf, new_edge};
    const float4* d_feat[2] = {nullptr, nullptr};
    for (int kind = 0; kind < 2; kind++) {
        const lili_cloud* c = feats[kind];
        const bool in_place = c->mem == LILI_MEM_DEVICE && c->stride == sizeof(float4) && c->aux_offset == 12 && (reinterpret_cast<uintptr_t>(c->data) & 15) == 0;
        if (in_place || c->n == 0) { d_feat[kind] = static_cast<const float4*>(c->data); continue; }
        if ((rc = lili_ingest_cloud(ctx, c, ctx->kf_in[kind])) != LILI_OK) return rc;
        d_feat[kind] = ctx->kf_in[kind].as<float4>();
    }
    const int slot_new = slots[n_slots - 1];
    ARGCHK(slot_new >= 0 && slot_new < LILI_MAX_SLOTS, "backend_keyframe_prepare: bad slot");
    // 2 + 3. per kind: commit (ring -> VoxelGrid -> index); the new keyframe's filter of that kind is enqueued behind the index build's kernels
    const float leaf_q[2] = {opt->leaf_surf, opt->leaf_edge}, leaf_m[2] = {opt->leaf_surf_map, opt->leaf_edge_map};
    const double gate[2] = {match->kd_max_radius, match->edge_gate};
    for (int kind = 0; kind < 2; kind++) {
        bool enq = false, pending = false, in_hook = false;
        const int n_f = (int)feats[kind]->n;
        auto enqueue_filter = [&]() -> int {
            const int r = n_f > 0 ? lili_voxel_filter_dev_enqueue(ctx, d_feat[kind], n_f, leaf_q[kind], in_hook, &pending) : LILI_OK;
            enq = r == LILI_OK;
            return r;
        };
        int64_t a = 0, b = 0;
        if (lili_localmap_ring_size(ctx, kind) > 0) {
            ctx->pre_sync_hook = [&]() -> int { in_hook = ctx->hook_box_words_zero; const int r = enqueue_filter(); in_hook = false; return r; };
            rc = lili_localmap_commit(ctx, kind, leaf_m[kind], gate[kind], &a, &b);
            ctx->pre_sync_hook = nullptr;
            if (rc != LILI_OK) return rc;
        }
        R.n_map_raw[kind] = (int32_t)a; R.n_map[kind] = (int32_t)b;
        if (!enq) {
            if ((rc = enqueue_filter()) != LILI_OK) return rc;
            if (pending) { rc = lili_readback_finish(ctx); if (rc != LILI_OK) return rc; }
        }
        const float4* d_q = d_feat[kind]; int n_q = 0;
        if (n_f > 0) { if ((rc = lili_voxel_filter_dev_complete(ctx, &d_q, &n_q)) != LILI_OK) return rc; }
        R.n_query[kind] = n_q;
        const lili_cloud qc{d_q, (size_t)n_q, 16, 12, LILI_MEM_DEVICE};
        if ((rc = lili_s2m_set_queries(ctx, slot_new, kind, &qc)) != LILI_OK) return rc;      // device-to-device: the filter's buffer is free for the other kind
        stamp(kind);
    }
    // 4. the window's associations (one launch for all keyframes where the sizes allow it), ONE synchronisation for the 2 n counts
    const int mask = LILI_MASK_SURF | LILI_MASK_EDGE;
    const bool maps = ctx->map[LILI_KIND_SURF].valid && ctx->map[LILI_KIND_EDGE].valid;
    if (maps) {
        if ((rc = lili_s2m_associate_window(ctx, slots, n_slots, mask, t_assoc, q_assoc, match, n_res)) != LILI_OK) return rc;
    } else if (n_res) for (int i = 0; i < 2 * n_slots; i++) n_res[i] = 0;
    R.associated = maps ? 1 : 0;
    stamp(2);
    if (res) *res = R;
    return LILI_OK;
}

}  // extern "C"
