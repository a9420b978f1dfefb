// Internal: context and buffer types shared by the translation units of liblili_hip.so (not part of the ABI).
#pragma once
#include "../../include/lili_hip.h"
#include "lili_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <utility>
#include <vector>

using namespace lili;

namespace lili_detail {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes + 256 <= cap) return hipSuccess;      // (every buffer keeps >= 256 bytes of slack behind what was asked for: kernels may read whole 16-byte vectors at the end)
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    void swap(DevBuf& o) { void* tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }   // lili_ctx_destroy deletes the context with its device current
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct MapIndex {
    bool valid = false;
    int64_t n = 0, n_cells = 0;
    double cell = 0;
    GridView view{};
    DevBuf sorted, aux_sorted, cell_start, cell_tmp, pt_cell /* rank of every point inside its cell (int), between the count and the scatter pass */, block_sums, cell_start9, row9;
    bool has_aux = false;
    // density adaptation: a second index with cells sized from the measured density (dense maps only) — the ONLY one the association of a dense map searches (lili_s2m_dense.hip)
    bool has_fine = false;
    GridView fview{};
    DevBuf sorted_f, aux_sorted_f, cell_start_f, cell_start9_f;
    float fbound = 0.f;          // squared radius the fine index covers completely (rounded down)
    double fine_cell = 0, mean_occupancy = 0;
    void swap(MapIndex& o) {     // lili_map_set_begin / _end: the index under construction and the current one trade places
        std::swap(valid, o.valid); std::swap(n, o.n); std::swap(n_cells, o.n_cells); std::swap(cell, o.cell); std::swap(view, o.view);
        sorted.swap(o.sorted); aux_sorted.swap(o.aux_sorted); cell_start.swap(o.cell_start); cell_tmp.swap(o.cell_tmp); pt_cell.swap(o.pt_cell);
        block_sums.swap(o.block_sums); cell_start9.swap(o.cell_start9); row9.swap(o.row9);
        std::swap(has_aux, o.has_aux); std::swap(has_fine, o.has_fine); std::swap(fview, o.fview);
        sorted_f.swap(o.sorted_f); aux_sorted_f.swap(o.aux_sorted_f); cell_start_f.swap(o.cell_start_f); cell_start9_f.swap(o.cell_start9_f);
        std::swap(fbound, o.fbound); std::swap(fine_cell, o.fine_cell); std::swap(mean_occupancy, o.mean_occupancy);
    }
};

// What the next index build of a kind may start from (lili_map_set): the previous build's TRUE bounding box (ordered-uint words as the
// device reduces them), the size and gate radius of that cloud, and the margin (in cells) a guessed box adds on every side (doubles whenever a guess fails).
struct SpecBox {
    bool valid = false;
    unsigned mm[6] = {0, 0, 0, 0, 0, 0};
    int64_t n = 0;
    double max_sq_radius = 0;
    int margin_cells = 1;
    int hits = 0, misses_in_a_row = 0, skip_guesses = 0;      // guesses that held since the margin last changed / consecutive misses / builds that measure before guessing resumes
    bool dense = false;            // the last build of this kind found a dense map and gave it the fine index (lili_map_set then skips the super-row copy of the gate-sized index)
    bool wide_counts = false;      // a build of this kind met a cell of more than 255 points: 32-bit cell counters from then on (k_cell_count instead of k_cell_count_narrow)
};

struct KindSlot {
    int64_t n_q = 0;
    bool has_queries = false, has_records = false;
    bool has_aux = false;    // the query cloud carried the auxiliary float (Livox: reflectivity)
    DevBuf q, rec0, rec1, valid, dbg_idx, dbg_d2, partials, partials_wave, block_counts;
    int launches = 0;        // association launches since set_queries
    int n_assoc_blocks = 0;  // grid of the last association launch (= number of per-block counts)
    int n_blocks = 0;      // association grid (one thread per query)
    int n_lin_blocks = 0;  // linearisation grid (grid-stride, <= kMaxLinBlocks partials)
};

struct Slot {
    KindSlot k[2];
    bool use_global_counts = false;   // next linearize_dev scales with the caller's (all-reduced) counts instead of its own
    const int32_t* global_counts = nullptr;
    bool sticky_global_counts = false;   // lili_s2m_counts_window_sharded: the global counts stay in force until the slot's next association
    DevBuf lm_cnt;                            // k_iterate_coop: the correspondence counts of the workgroups as granules
    DevBuf lm_part, lm_gsum, lm_summary;      // lili_s2m_solve_lm: granule-tagged block partials / group sums (two parities each), device copy of the summary
    int assoc_since_pose = 0;         // association launches since the slot's pose was (re)set: the first one is the far-from-converged launch (coop_lanes)
};

constexpr int kLinBlock = 1024;      // must match lili_s2m.hip

constexpr int kMaxLinBlocks = 256;
inline size_t lds_linearize(int threads) { return (size_t)threads * 12 * sizeof(double); }   // rows [J r | 1 cost 0 0]; reused for the per-wave result blocks

}  // namespace lili_detail
using namespace lili_detail;

constexpr size_t kMiscAlloc = 2 * 64 * 128 + 256 + 2 * ((8192 + 2) * 8 + 112);      // ctx->misc: scratch words of a map build, laid out in lili_map.hip
struct lili_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side[LILI_MAX_SLOTS] = {};   // lili_s2m_iterate_window: one extra stream per concurrently iterated slot (lazy)
    hipEvent_t fork_ev = nullptr, join_ev[LILI_MAX_SLOTS] = {};
    bool keep_nn = false;
    std::string err;
    MapIndex map[2];
    // lili_map_set_begin / _end: the NEXT index of a kind is built on its own stream (own scratch: staging, bbox / density words) while work
    // already enqueued keeps using the current one
    MapIndex map_next[2];
    hipStream_t build_stream = nullptr;
    hipEvent_t build_done[2] = {}, main_mark[2] = {};
    hipEvent_t cloud_ready = nullptr;      // lili_map_set_begin with a device cloud: the build stream waits for what the context's stream has enqueued so far
    unsigned long long lm_launches = 0;    // lili_s2m_solve_lm launches of this context: part of the granule keys, so no launch ever sees an older one's partials as its own
    SpecBox spec_box[2];
    bool map_guess_box = true;             // lili_map_set without a box from its caller starts from the previous build's box (+ margin) and checks it at its final read-back (0: measure first, A/B)
    int box_guesses = 0, box_guess_misses = 0, narrow_overflows = 0;
    bool map_narrow_counts = true;         // 8-bit cell counters in the index build (a quarter of the table to clear, fill and scan; 0: 32-bit always, A/B)
    int scan_fallbacks = 0;                // map builds repeated with the three-kernel scan because a look-back scan gave up (never expected; lili_map_info reports it)
    bool build_pending[2] = {false, false}, main_marked[2] = {false, false};
    DevBuf staging_build, misc_build;
    Slot slots[LILI_MAX_SLOTS];
    DevBuf states;       // SlotState[LILI_MAX_SLOTS]
    DevBuf staging;      // raw host clouds
    DevBuf kf_in[2];     // lili_backend_keyframe_prepare: the new keyframe's surf / edge features as float4 rows
    DevBuf fmt_out;      // lili_livox_custom_to_cloud output when the caller wants it on the host
    DevBuf gram;         // LILI_GRAM_DOUBLES per slot
    DevBuf pose_pub;     // kPubReplicas x kPubStride doubles per slot: the pose a Gauss-Newton kernel publishes for the association launched behind it without a barrier (overlap_gn)
    DevBuf win_rec;      // lili_s2m_linearize_window: the n x LILI_GRAM_DOUBLES records of one evaluation of the window (k_window_reduce's output)
    DevBuf win_counts;   // lili_s2m_associate_window: [surf, edge] counts of every slot (k_window_counts' output)
    DevBuf misc;         // bbox words etc.
    // Small device-to-host reads (counts, boxes, states) land in a page-locked scratch and are copied out after the synchronisation: a D2H into
    // pageable memory is staged by the runtime and blocks, which costs ~10 us more per read (64 KB; lili_readback_* in lili_api.hip).
    unsigned char* h_pin = nullptr;
    size_t h_pin_used = 0;
    unsigned long long readback_gen = 0;      // number of lili_readback_finish calls so far: a deferred reader knows whether its items have been delivered
    struct PinItem { void* dst; size_t off, bytes; };
    std::vector<PinItem> h_pin_items;
    struct LazyItem { const void* src; size_t off, bytes; };
    std::vector<LazyItem> h_pin_lazy;          // items whose device-side read happens in lili_readback_finish's gather launch (k_readback_gather)
    unsigned char* h_pin_dev = nullptr;        // the scratch as the device sees it
    bool readback_gather = true;
    double* h_records = nullptr;   // page-locked landing area for LILI_MAX_SLOTS Gram records (+ 2 x LILI_MAX_SLOTS counts behind them)
    double* h_records_dev = nullptr;   // the same memory as the device sees it: the blocking calls' reduction kernels write their records THERE (round 4: no copy launch between the kernel and the host)
    int max_cells = 1 << 27;
    int grid_reach = 2;          // 2: cells smaller than the gate radius, inner 3x3x3 block first, shell on demand (knn5_grid)
    bool fuse_tail = false;      // reduce (+ GN) inside the linearisation launch (its last block sweeps the other blocks' granule-tagged partials):
                                 // bit-identical, measured SLOWER than the separate k_reduce_partials launch (41.1 vs 38.9 us per iteration: reading 125 KB
                                 // of freshly published partials through sc1 loads costs one block 3.4 us, a kernel boundary 1.5 us) — A/B only
    bool no_p2p_fusion = false;  // A/B: lili_s2m_iterate_sharded with lili_p2p_allreduce as separate launches (the generic path) instead of inside the count / reduce kernels
    bool merge_kinds = true;     // surf and edge of one keyframe in ONE association launch / ONE linearisation launch
    int cell_pct = 65;           // reach 2: cell edge in % of 1.01 * gate radius (>= 50)
    bool fine_grid = true;       // measure the map density in lili_map_set and build the fine index when a gate-sized cell holds more than fine_occupancy points
    int fine_occupancy = 12;
    double focus[3] = {0, 0, 0}, focus_radius = 0;   // lili_map_focus: where the super-row copy is built (radius 0: everywhere)
    int fuse_lin_block = 0;      // 0 = by scan size, else 64 / 256 (A/B, tests)
    int assoc_lpq = 0;           // lanes per query of the association: 0 = by launch size (coop_lanes), 1 = one lane per query always, 2 / 4 / 8 / 16 forced (A/B, tests)
    bool count_barrier = false;  // (measured: 19.4 vs 19.8 us per iteration at 2 k queries — the wait for the slowest workgroup costs what the launch saved; off by default)

    bool persistent_off_now = false;  // set by lili_s2m_iterate_window for more than four slots
    bool persistent_iterate = false;  // lili_s2m_iterate* of small scans (<= 128 cooperative workgroups): one persistent launch per registration (k_iterate_coop).
                                      // Measured (tools/iter_time.py, profiles/r03_iter_time.json): 21.3 vs 19.6 us per outer iteration at 2 k queries (ROT), 17.2 vs 14.9
                                      // (front end) — two exchange hops through memory across the XCDs cost more than the launch boundaries they replace; off by default
    bool frontend_commit_pending = false;   // lili_frontend_frame: the ring has a keyframe the local map does not hold yet (the commit runs at the start of the next frame, under its extraction)
    int32_t frontend_map_raw = 0;          // ring points of the local map the front-end frames are matched against (the last commit's)
    bool voxel_guess_bits = true;      // a VoxelGrid of more than 8192 points keeps its bounding box on the device and guesses its key bits from the previous filter (k_vox_key_dev; 0: measured, A/B)
    bool hook_box_words_zero = false;      // valid inside pre_sync_hook: the first words of ctx->misc are still zero from the build's scratch fill
    std::function<int()> pre_sync_hook;      // one-shot: called by the next map build right before its read-back synchronises, so that the caller's launches and read-backs share that synchronisation (lili_pipeline.hip)
    bool voxel_small = true;     // lili_voxel_filter / lili_frontend_frame: clouds of <= 8192 points are filtered by ONE single-workgroup launch (k_voxel_small; 0: the general chain, A/B)
    bool overlap_gn = false;     // lili_s2m_iterate*, three-launch path: the reduction + GN kernel publishes the pose as keyed granules and the next association is launched without a
                                 // barrier against it (hipExtAnyOrderLaunch): its waves are dispatched and poll for the pose while the reduction still runs
    unsigned long long gn_seq = 0;   // published poses of this context so far (their keys never repeat)
    bool fuse_lin = true;        // lili_s2m_iterate*: flavours without count scaling linearise inside the association launch (k_associate_lin)
    bool super_rows = true;      // lili_map_set also stores the super-row copy of the map (9x the points): the inner 27-cell block of a query is one run
    bool scan_lookback = true;   // map index: single-pass (decoupled look-back) scan of the cell array; 0 = the three-kernel scan (A/B)
    bool localmap_super_rows = false;   // lili_localmap_commit builds the super-row copy also for small maps (< 400 k points)
    bool localmap_incremental = true;   // lili_localmap_commit keeps the ring sorted by voxel and merges one keyframe per step (0: rebuild every time, A/B)
    int sort_fused_max_tiles = 256;   // (measured: 200 k keys 212 -> 190 us per filter, 1 M keys 301 -> 296; 489 tiles of a 2 M-key sort: 290 -> 475 us)
    bool sort_fused_scan = true;      // radix passes of at most sort_fused_max_tiles tiles: the scatter kernel derives its offsets from the count table itself (no scan launch)
    // A frame's extraction on a stream of its own (lili_pipeline.hip, option "frame_extract_stream"): the extractor's kernels run next to the ring merge of the local map the
    // previous frame left pending — they share nothing — instead of in front of it.  `extract_side_next`: the next deferred extraction may take side[kExtractSide] (one shot,
    // set by the frame); `extract_join_pending`: the context's stream has not yet waited for that extraction — lili_readback_finish does, before it gathers.
    static constexpr int kExtractSide = 2;
    hipEvent_t extract_fork_ev = nullptr;
    bool frame_extract_stream = false, extract_side_next = false, extract_join_pending = false;      // (default OFF: measured -15..30 us per frame in a process with one context, +100 us in one that holds a dozen streams — the whole bench —)
    bool rot_fold = true, rot_segment_wait = true;      // lili_extract_rot: k_rot_ring writes the scan's lists itself (0: k_rot_compact behind it) / a segment whose pick may lie under its predecessor's marks waits for them in k_rot_segments (0: k_rot_ring redoes it) — A/B and fallback paths
    int frame_guess_misses = 0;         // frames whose guessed feature counts were too small (matched again the plain way)
    bool frame_guess_counts = true;     // lili_frontend_frame_rot on a caller's maps: the matcher is enqueued behind the extractor with GUESSED feature counts (one synchronisation per scan; 0: wait for the counts first, A/B)
    bool sort_ride_hist = true;         // the radix sort's digit histograms ride on the key kernel and on the scatter passes (a pass = one launch; 0: a histogram launch per pass, A/B)
    int sort_digit_bits = 8;     // radix sort of the voxel filter: 8-bit digits (4 = the round-2 passes, A/B)
    int rot_atan = 2;            // ROT extractor: 2 = glibc fdlibm float atan / atan2 (the reference build's bits), 1 = f64 functions rounded to f32
    int n_simd = 0;              // SIMDs of the device (CUs x 4)
    void* ext_rot = nullptr;                 // extractor state (lili_extract_rot.hip), freed through ext_rot_free
    void (*ext_rot_free)(void*) = nullptr;
    void* ext_livox = nullptr;
    void (*ext_livox_free)(void*) = nullptr;
    void* ext_voxel = nullptr;               // voxel filter / local-map ring buffer (lili_voxel.hip)
    void (*ext_voxel_free)(void*) = nullptr;

    int fail(int code, const std::string& m) { err = m; return code; }
    SlotState* state(int slot) { return states.as<SlotState>() + slot; }
    double* gram_of(int slot) { return gram.as<double>() + (size_t)slot * LILI_GRAM_DOUBLES; }
    // Page-locked copy of the slots' states for "iterate, then read the pose" without a copy launch (lili_pipeline.hip): `state_mirror_want` asks the next lili_s2m_iterate to
    // hand the mirror to the reduction + GN kernel of its LAST iteration (take_state_mirror: one shot); a path without such a kernel leaves the host's sentinel in place.
    lili::SlotState* h_state_mirror = nullptr; lili::SlotState* h_state_mirror_dev = nullptr;
    bool state_mirror_want = false, state_mirror_armed = false;
    lili::SlotState* take_state_mirror(int slot) { if (!state_mirror_armed || !h_state_mirror_dev) return nullptr; state_mirror_armed = false; return h_state_mirror_dev + slot; }
    double* pub_of(int slot) { return pose_pub.as<double>() + (size_t)slot * kPubReplicas * kPubStride; }      // option overlap_gn: the slot's published-pose copies
};

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return ctx->fail(LILI_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define ARGCHK(cond, msg) do { if (!(cond)) return ctx->fail(LILI_E_ARG, msg); } while (0)

static inline int nblocks(int64_t n, int per) { return (int)((n + per - 1) / per); }

// enqueue a small read of device memory on `stream` (default: the context's) into the page-locked scratch; lili_readback_finish synchronises that stream and
// copies every pending item to its destination.  One thread per context, items of one stream per finish.
// lili_map_set for clouds the library itself produced (the local map): `box6` = the points' bounding box as the six ordered-uint words k_bbox leaves
// (min xyz, max xyz) already on the host — no read-back before the grid —, `in_place` = a device float4 array (16-byte rows, aux in w or none) that stays
// untouched until the build is through — no ingestion copy.  Either may be off (nullptr / false).
int lili_map_set_hinted(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius, const unsigned* box6, bool in_place);
int lili_readback_add(lili_ctx* ctx, void* dst, const void* d_src, size_t bytes, hipStream_t stream = nullptr);
int lili_readback_finish(lili_ctx* ctx, hipStream_t stream = nullptr);
// a lazy read-back reads its device source when the FINISH launches its gather kernel, so nothing enqueued in between may rewrite that source.  Every fill of
// the shared scratch words (ctx->misc) asks here first: LILI_E_STATE if [p, p + bytes) overlaps the source of a read-back that is still pending (a programming error in
// the library — a hook or a commit that clears words another stage has yet to deliver), LILI_OK otherwise.  Host-side, a loop over at most twelve items.
int lili_lazy_sources_clear_of(lili_ctx* ctx, const void* p, size_t bytes);
void* lili_pinned_dev_ptr(const void* host, size_t align = 4);      // device-side address of a PAGE-LOCKED host buffer, or nullptr (pageable / misaligned); asked per call: ~0.1 us
int lili_ingest_cloud(lili_ctx* ctx, const lili_cloud* c, lili_detail::DevBuf& out_f4, unsigned* d_bbox = nullptr);   // d_bbox: also reduce the bounding box (6 ordered-uint words, initialised by the caller)
// lili_p2p.hip: the view of the NEXT exchange of a communicator (advances its sequence number); usable = connected and on this context
lili::P2PView lili_p2p_next_view(lili_p2p* c);
bool lili_p2p_usable(const lili_p2p* c, const lili_ctx* ctx);
// lili_voxel.hip -> lili_pipeline.hip: the voxel filter and the keyframe ring on device clouds (no host copies, the pose read on the device)
int lili_voxel_filter_dev(lili_ctx* ctx, const float4* d_pts, int n, float leaf, const float4** d_out, int* n_out);
int lili_voxel_filter_dev_enqueue(lili_ctx* ctx, const float4* d_pts, int n, float leaf, bool box_zeroed, bool* pending);
int lili_voxel_filter_dev_complete(lili_ctx* ctx, const float4** d_out, int* n_out);
int lili_localmap_push_dev(lili_ctx* ctx, int kind, const float4* d_pts, int n, const lili::SlotState* d_state, int width);
int lili_localmap_ring_size(lili_ctx* ctx, int kind);
// lili_extract_livox.hip -> lili_pipeline.hip: the extraction enqueued without its synchronisation, and the counts taken afterwards
int lili_extract_livox_enqueue(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params);
int lili_extract_livox_complete(lili_ctx* ctx);
int lili_extract_livox_device_ex(lili_ctx* ctx, lili_cloud* edge, lili_cloud* surf, bool convert_edge);      // lili_extract_livox_device; convert_edge = false: the edge cloud's count only
// Where an extraction may put its feature lists IN ADDITION to its own device lists (lili_pipeline.hip: a frame with guessed feature counts): the query arrays of a matcher
// slot — rows behind the lists' ends are filled with NaN up to the capacities (lili_s2m_set_queries_counted) —, and the slot's state, set to `pose` like lili_s2m_pose_set.
struct lili_query_sink { float4* q_surf = nullptr; int cap_surf = 0; float4* q_edge = nullptr; int cap_edge = 0; lili::SlotState* state = nullptr; double pose[7] = {0, 0, 0, 1, 0, 0, 0}; };
int lili_extract_rot_enqueue(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params, const lili_query_sink* sink = nullptr);
// list lengths of the previous completed extraction on this context (0 / 0: none)
void lili_extract_rot_prev(lili_ctx* ctx, int* prev_edge, int* prev_surf);
int lili_extract_rot_complete(lili_ctx* ctx);
// did the last lili_extract_rot_complete rewrite the lists (second passes)?
bool lili_extract_rot_redone(lili_ctx* ctx);
// lili_match.hip -> lili_pipeline.hip: a slot sized for a GUESSED number of queries (the producer pads with NaN rows), trimmed once the number is known
int lili_s2m_set_queries_counted(lili_ctx* ctx, int slot, int kind, int n_guess);
int lili_s2m_trim_queries(lili_ctx* ctx, int slot, int kind, int n);
