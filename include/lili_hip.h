/*
 * lili_hip.h — C ABI of the MI355X (gfx950) hot path of LiLi-OM.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, no
 * exceptions.  Every entry point names the reference code it replaces
 * (L/ = LiLi-OM/, R/ = LiLi-OM-ROT/ of KIT-ISAS/lili-om).  INTEGRATION.md shows the ROS-node /
 * ceres::CostFunction side binding.
 *
 * Conventions
 *   - quaternions are double[4] in (w, x, y, z) order — the order of the reference's Ceres parameter
 *     blocks (tmpQuat, L/src/BackendFusion.cpp:853-856); translations double[3].
 *   - point clouds are described by (pointer, count, stride, memory space): x,y,z are 3 floats at byte
 *     offset 0 of every point — true for pcl::PointXYZI (32 B) and pcl::PointXYZINormal (48 B,
 *     L/include/utils/common.h:71-73); `aux_offset` is the byte offset of one extra float per point
 *     (Livox: curvature = 0.1*reflectivity at 36; ROT: intensity at 16) or -1.
 *   - every function returns LILI_OK (0) or a negative error code and never throws;
 *     lili_last_error() gives the message of the last failure on that context.
 *   - a context owns one HIP stream (or borrows the caller's); it is NOT thread-safe; all work of a
 *     context is ordered on that stream.  Functions documented "async" only enqueue.
 *   - results are deterministic: no floating-point atomics anywhere; reductions have a fixed order.
 */
#ifndef LILI_HIP_H
#define LILI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LILI_ABI_VERSION 1

typedef struct lili_ctx lili_ctx;

enum {
    LILI_OK = 0,
    LILI_E_ARG = -1,      /* bad argument */
    LILI_E_HIP = -2,      /* HIP runtime error (message in lili_last_error) */
    LILI_E_STATE = -3,    /* call order violated (e.g. associate before map_set) */
    LILI_E_NOMEM = -4,
    LILI_E_NODEVICE = -5  /* no usable gfx950 device: there is no CPU fallback */
};
enum { LILI_KIND_SURF = 0, LILI_KIND_EDGE = 1 };
enum { LILI_MASK_SURF = 1, LILI_MASK_EDGE = 2 };
enum { LILI_VARIANT_LIVOX = 0, LILI_VARIANT_ROT = 1, LILI_VARIANT_FRONTEND = 2 };
enum { LILI_LOSS_NONE = 0, LILI_LOSS_CAUCHY = 1, LILI_LOSS_HUBER = 2 };
enum { LILI_MEM_HOST = 0, LILI_MEM_DEVICE = 1 };

#define LILI_MAX_SLOTS 8       /* sliding-window keyframes handled per context (slide_window_width = 3) */
#define LILI_GRAM_DOUBLES 72   /* device-side reduction record: 64 Gram + cost + 7 spare */

typedef struct lili_cloud {
    const void* data;   /* first point */
    size_t n;           /* number of points */
    size_t stride;      /* bytes between points (>= 12) */
    int aux_offset;     /* byte offset of the per-point auxiliary float, or -1 */
    int mem;            /* LILI_MEM_HOST or LILI_MEM_DEVICE */
} lili_cloud;

/* Matcher parameters — the ★ rows of SURVEY.md App. C. */
typedef struct lili_s2m_params {
    int variant;             /* LILI_VARIANT_*: which findCorresponding* / factor flavour */
    int loss;                /* LILI_LOSS_*: CauchyLoss(1.0) L/src/BackendFusion.cpp:845; HuberLoss(0.1) L/src/LidarOdometry.cpp:507 */
    double loss_a;
    double lidar_const;      /* L/config/config_fr_iosb.yaml:18 (20), R/config/config_fr_iosb.yaml:30 (7.5) */
    double kd_max_radius;    /* compared with a SQUARED distance, L/src/BackendFusion.cpp:1615 */
    double edge_gate;        /* 1.0, L/src/BackendFusion.cpp:1543 (squared distance too) */
    double surf_dist_thres;  /* L:1651 */
    double reflect_thres;    /* Livox only, L:1628 */
    double surf_weight_min;  /* 0.2 L:1665 / 0.3 R/src/BackendFusion.cpp:1504 / 0.4 L/src/LidarOdometry.cpp:400 */
    double edge_dist_max;    /* ROT only, R/src/BackendFusion.cpp:1443 (0.1); <= 0 disables */
    double q_lb[4];          /* extrinsic rotation, (w,x,y,z), L/config/config_fr_iosb.yaml:35-38 */
    double t_lb[3];          /* extrinsic translation */
    double scale_surf_num;   /* ROT: residual scale = num / N_surf (1000, R/src/BackendFusion.cpp:861); 0 = no count scaling */
    double scale_edge_num;   /* ROT: 200 / N_edge (R/src/BackendFusion.cpp:843); 0 = none */
} lili_s2m_params;

/* ---- context -------------------------------------------------------------------------------- */

/* Creates a context on HIP device `device`.  `stream` is a hipStream_t to enqueue on (e.g. the
 * caller's current stream so that RCCL collectives interleave correctly), or NULL to create one.
 * Fails with LILI_E_NODEVICE when no gfx950 device is usable — there is no CPU path. */
int lili_ctx_create(lili_ctx** out, int device, void* stream);
void lili_ctx_destroy(lili_ctx* ctx);
const char* lili_last_error(const lili_ctx* ctx);
int lili_abi_version(void);
/* Blocks until everything enqueued on the context's stream has finished. */
int lili_sync(lili_ctx* ctx);
/* When enabled, associate also stores the 5 neighbour indices / squared distances per query so that
 * lili_s2m_get_neighbors can return them (parity tests).  Off by default (extra HBM writes). */
int lili_set_debug(lili_ctx* ctx, int keep_neighbors);
/* Tuning knobs that never change any result bit (the closed experiments "bin_queries", "tiled", "nn_cache" and "balance" were removed in round 5:
 * profiles/EXPERIMENTS.md; an unknown name returns LILI_E_ARG):
 *   map index     "max_cells" (cap on grid cells; coarser cells stay exact), "grid_reach" (1 = cells of the gate radius, 27-cell search; 2 = smaller
 *                 cells, inner 27 first and the shell of the 125-cell block on demand; default 2), "cell_pct" (reach 2: cell edge in % of the gate radius,
 *                 50..100, default 65), "super_rows" (1 = the map is also stored in the super-row layout, see lili_map_focus; default 1), "fine_grid" (1 = a
 *                 map with more than "fine_occupancy" (default 12) points per gate-sized cell gets a second index with density-sized cells that the
 *                 association searches first — exact; default 1), "scan_lookback" (1 = single-pass decoupled look-back scans; default 1),
 *                 "map_narrow_counts" (1 = byte-wide cell counts while they fit; default 1), "map_guess_box" (1 = a rebuild guesses its bounding box from the
 *                 previous map of the kind and verifies it on the device; default 1; setting it resets the guess state).  All take effect at the next
 *                 lili_map_set;
 *   iteration     "fuse_tail" (1 = reduction + GN update run in the last block of the linearisation launch; default 0 = separate launch, faster on
 *                 MI355X), "merge_kinds" (1 = surf and edge of a keyframe share ONE association and ONE linearisation launch; default 1), "fuse_lin"
 *                 (1 = flavours without count scaling run two launches per iteration below ~100 k queries, the association linearises on the fly;
 *                 default 1; "fuse_lin_block" its workgroup size), "assoc_lpq" (lanes per query in the association of small scans: 0 = by size,
 *                 1 / 2 / 4 / 8 / 16 forces a value), "count_barrier" (default 0, measured no gain), "persistent_iterate" (1 = scans of <= 128
 *                 cooperative workgroups run a whole registration as ONE persistent launch; default 0, measured slower; poses then differ by the
 *                 partition of the Gram sum, <= 1e-10), "p2p_fusion" (0 = lili_s2m_iterate_sharded runs lili_p2p_allreduce as its own launches;
 *                 default 1);
 *   local map     "localmap_incremental" (default 1, see lili_localmap_commit), "localmap_super_rows" (1 = ring maps below 400 k points get the
 *                 super-row copy too; default 0), "sort_digit_bits" (8, or 4 = the round-2 radix passes), "sort_fused_scan" (1 = radix passes of at most
 *                 "sort_fused_max_tiles" (256) tiles derive their offsets inside the scatter kernel; default 1), "sort_ride_hist" (1 = in the frame pipeline's query filter the
 *                 digit histograms ride on the key kernel and the scatter passes, one launch per pass; default 1), "voxel_small" (1 = clouds of <= 8192
 *                 points are voxel-filtered / keyframe-sorted by ONE workgroup in LDS; default 1), "voxel_guess_bits" (see lili_voxel_filter_stats; default 1);
 *   host          "readback_gather" (1 = the small reads of a synchronisation are gathered by one kernel writing into page-locked memory instead of
 *                 one copy launch each; default 1), "frame_guess_counts" (1 = lili_frontend_frame_rot with LILI_FRAME_EXTERNAL_MAP and leaf_query 0 enqueues the
 *                 matcher behind the extractor for GUESSED feature counts — the previous scan's plus a margin, padding rows select nothing — and synchronises once;
 *                 a scan with more features than guessed is matched again the plain way; results identical either way; default 1), "frame_extract_stream" (1 = a frame whose
 *                 local map is still to be built runs its extraction on a stream of its own, next to the ring merge instead of in front of it: 15-30 us per frame faster in a
 *                 process with one context, ~100 us SLOWER in a process that holds many streams (they share the runtime's few hardware queues); default 0).
 *   extraction    "rot_fold" (1 = the ring stage of lili_extract_rot writes the scan's feature lists itself, four launches; 0 = per-ring lists and a concatenation
 *                 launch, also the fallback of a look-back that gave up; default 1), "rot_segment_wait" (1 = a segment whose pick may lie under marks of the
 *                 segment before it waits for them inside the segment stage; 0 = the ring stage repeats such a segment; default 1).
 * One knob that DOES choose between two definitions of a result: "rot_atan" — lili_extract_rot's atan / atan2 on float arguments
 * (R/src/Preprocessing.cpp:285-288,315,349): 2 (default) = glibc's float routines statement for statement (atanf / atan2f of
 * every glibc up to 2.40 — the bits a build of the reference produces), 1 = the f64 functions rounded to f32 (libm-independent). */
int lili_set_option(lili_ctx* ctx, const char* name, int value);

/* Page-locked host memory for clouds handed over with LILI_MEM_HOST: a cloud in such a buffer is DMA'd straight into HBM (an 80 MB
 * map: ~1.5 ms), a pageable one is staged by the runtime page by page (~8 ms).  Plain hipHostMalloc / hipHostFree — a caller that
 * already owns pinned memory (hipHostRegister on a PCL cloud's buffer) needs neither. */
void* lili_host_alloc(size_t bytes);
void lili_host_free(void* p);

/* ---- local map index ------------------------------------------------------------------------ */

/* Replaces kd_tree_{surf,edge}_local_map->setInputCloud(...) (L/src/BackendFusion.cpp:839-840,
 * L/src/LidarOdometry.cpp:490).  Copies the cloud (if on the host), bins it into a uniform grid whose
 * cell edge covers sqrt(max_sq_radius) so that the 27-cell neighbourhood of a query contains every
 * point the reference's gate `d2[4] < max_sq_radius` can accept — the search is EXACT for all queries
 * the reference keeps.  Blocking.  The points are read where they lie (a device cloud is not copied; it must stay valid until the
 * call returns).  The first build of a kind measures the cloud's bounding box and reads it back before the grid exists; later builds of a cloud
 * of about the same size and gate start from the previous build's box grown by a margin of cells, and check at their final read-back — the one with
 * the density — that no point of THIS cloud lay outside it; if one did, the index is rebuilt with the true box before the call returns and the margin
 * doubles (option "map_guess_box" = 0: always measure first; lili_map_build_stats counts both).  Search results do not depend on where the grid's
 * origin lies. */
int lili_map_set(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius);
/* Diagnostics of the index builds of this context: builds that started from a guessed box, those of them that had to be repeated with the true
 * box, builds repeated with the three-kernel scan because a single-pass scan gave up (never expected).  Any pointer may be NULL. */
int lili_map_build_stats(lili_ctx* ctx, int32_t* box_guesses, int32_t* box_guess_misses, int32_t* scan_fallbacks);
/* The same in two steps for pipelines that rebuild the local map every keyframe (L/src/BackendFusion.cpp:839-840 runs once per keyframe, before
 * the window's iterations): _begin builds the NEXT index of `kind` on a side stream into a second set of buffers — work already enqueued on the
 * context's stream (the previous keyframe's iterations, lili_s2m_iterate*) keeps the current index and overlaps with the build —, _end makes the
 * new index the current one for everything enqueued after it.  _begin waits for the build's small read-backs only (density and box check; the box itself on a first build).
 * The cloud must stay valid until _end.  Results are those of lili_map_set. */
int lili_map_set_begin(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius);
int lili_map_set_end(lili_ctx* ctx, int kind);
/* Number of points / grid cells of the current index (diagnostics). */
int lili_map_info(lili_ctx* ctx, int kind, int64_t* n_points, int64_t* n_cells, double* cell_edge);
/* Density adaptation of the current index (diagnostics): the point-weighted mean number of points per gate-sized cell measured by
 * lili_map_set (0 with option "fine_grid" = 0), and — if the map got the second, density-sized index — its cell edge and the squared
 * radius it covers completely (both 0 otherwise).  Reference call sites: L/src/BackendFusion.cpp:839-840 (setInputCloud on maps
 * down-sampled at L:1488-1511 — leaf sizes down to centimetres make a gate-sized cell hold hundreds of points). */
int lili_map_density(lili_ctx* ctx, int kind, double* mean_occupancy, double* fine_cell_edge, double* fine_sq_radius);

/* Performance hint (no reference counterpart: pcl::KdTreeFLANN indexes the whole cloud it is given, L/src/BackendFusion.cpp:839-840): a local
 * map may cover far more ground than one scan reaches.  lili_map_set stores, next to the cell-sorted points, a "super-row" copy in which the
 * 27-cell neighbourhood of a query is one contiguous run (option "super_rows", 9x the points); with a focus, that copy is built only for the
 * cells within `radius` of `center` (map frame; e.g. the sensor position and its maximum range + the pose uncertainty) at the following
 * lili_map_set calls.  Queries elsewhere take the nine-row walk over the base index: results never depend on the hint.  radius <= 0 or
 * center == NULL: the whole map again. */
int lili_map_focus(lili_ctx* ctx, const double center[3], double radius);

/* ---- feature extraction ----------------------------------------------------------------------- */

/* Caller-owned output cloud: `capacity` points of `stride` bytes (>= 16; 32 for pcl::PointXYZI, whose x,y,z
 * are floats at 0/4/8 — the 4th float written at offset 12 is the intensity, copy it to offset 16 for PCL or
 * pass stride 16 and repack); `count` receives the number of points available (may exceed capacity); entries of the buffer behind
 * `count` are unspecified (an extractor may have written there). */
typedef struct lili_feature_out {
    void* data;
    size_t capacity;
    size_t stride;
    int mem;          /* LILI_MEM_HOST or LILI_MEM_DEVICE */
    size_t count;     /* out */
} lili_feature_out;

typedef struct lili_rot_params {
    int n_scans;       /* 16 / 32 / 64  (R/src/Preprocessing.cpp:46, config line_num) */
    int ds_rate;       /* R/config/config_fr_iosb.yaml:13 */
    float ds_v;        /* 0.6, R/src/Preprocessing.cpp:14 */
    float near_range;  /* 3.0, R/src/Preprocessing.cpp:281 */
} lili_rot_params;

/* Replaces the body of Preprocessing::cloudHandler of LiLi-OM-ROT (R/src/Preprocessing.cpp:277-527) for one
 * scan: NaN / near-range filter, ring id + relative time, IMU deskew (q_imu = the integrated gyro quaternion of
 * R:179-223, computed by the caller; q_lb the extrinsic), ring concatenation, 11-tap curvature, per-segment
 * sharp / less-sharp / flat selection, per-ring VoxelGrid of the less-flat points.
 *   scan  : raw points in firing order, aux_offset = byte offset of the intensity field;
 *   full  : the ring-concatenated, deskewed cloud          (/lidar_cloud_cutted)
 *   edge  : cornerPointsLessSharp, in push order            (/edge_features)
 *   surf  : voxel-filtered less-flat points, ring by ring   (/surf_features)
 * Points are written as (x, y, z, intensity = ring + 0.1 * relTime).  Blocking.  A LILI_MEM_DEVICE scan of 16-byte rows (x, y, z, intensity)
 * is read in place; a scan in PAGE-LOCKED host memory (lili_host_alloc) is read by the first kernel where it lies instead of being uploaded first; a
 * `full` buffer in page-locked host memory receives min(scan->n, capacity) records while the features are selected — the records behind full->count
 * are unspecified —, `edge` / `surf` buffers in page-locked memory (16-byte aligned, stride a multiple of 16) are written by a kernel behind the
 * selection, `count` records each (only the first 16 bytes of a row), and the call synchronises once; pageable buffers are served by copies after the
 * counts have arrived. */
int lili_extract_rot(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params,
                     lili_feature_out* full, lili_feature_out* edge, lili_feature_out* surf);
/* Intermediate products of the last lili_extract_rot for parity tests (any pointer may be NULL):
 * counts = {n_full, n_edge, n_sharp, n_flat, n_lessflat, n_surf, halfPassed index, first valid index};
 * ring_start/ring_end: 64 ints (scanStartInd / scanEndInd); per-point arrays are n_full long; index lists point
 * into the full cloud. */
int lili_extract_rot_debug(lili_ctx* ctx, int32_t counts[8], int32_t* ring_start, int32_t* ring_end, int32_t* full_src, float* curvature,
                           int32_t* label, int32_t* edge_idx, int32_t* sharp_idx, int32_t* flat_idx, int32_t* lessflat_idx, int32_t* surf_cnt);

typedef struct lili_livox_params {
    double surf_thres;   /* L/config/config_fr_iosb.yaml:5 (0.28) */
    double edge_thres;   /* L/config/config_fr_iosb.yaml:6 (4)    */
    float near_range;    /* 0.1, L/src/Preprocessing.cpp:226      */
} lili_livox_params;

/* Replaces the body of Preprocessing::cloudHandler of LiLi-OM (L/src/Preprocessing.cpp:219-401) for one Livox
 * Horizon scan in FormatConvert's layout (L/src/FormatConvert.cpp:14-23): scan->aux_offset = byte offset of
 * `intensity` (line + 0.1 * t), curvature_offset = byte offset of `curvature` (0.1 * reflectivity) — 32 and 36 for
 * pcl::PointXYZINormal.  q_imu = the gyro quaternion integrated over the scan (L:129-171, caller side).
 * Outputs are records of 32 B (x,y,z,nx,ny,nz,intensity,curvature; stride 32) or pcl::PointXYZINormal (stride 48):
 *   cutted: every deskewed point with a valid line (/lidar_cloud_cutted), edge: /edge_features (normal = line
 *   direction), surf: /surf_features (normal = plane normal).  Blocking.  The rows are read as they are (one transfer for host memory);
 *   a `cutted` buffer in host memory receives min(scan->n, capacity) records while the features are selected — the records behind
 *   cutted->count are unspecified.  `edge` and `surf` buffers in PAGE-LOCKED host memory (lili_host_alloc, 16-byte aligned) are written by the
 *   packing kernel itself, `count` records each, and the call synchronises once; pageable buffers are served by copies after the counts have arrived. */
int lili_extract_livox(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params,
                       lili_feature_out* cutted, lili_feature_out* edge, lili_feature_out* surf);
/* counts = {n_cutted, n_edge, n_surf}; cut_src[n_cutted]; cell_src[24000] (input index owning each grid cell or -1);
 * edge_cell / surf_cell: cell (line * 4000 + column) of each emitted feature.  Any pointer may be NULL. */
int lili_extract_livox_debug(lili_ctx* ctx, int32_t counts[3], int32_t* cut_src, int32_t* cell_src, int32_t* edge_cell, int32_t* surf_cell);

/* Device views (float4 x,y,z,aux; LILI_MEM_DEVICE clouds, valid until the next extract call on the context) of the last
 * extraction, so that features go from the extractor to lili_voxel_filter / lili_s2m_set_queries / lili_localmap_push
 * without leaving HBM.  ROT: aux = intensity (ring + 0.1 relTime); Livox: aux = curvature (0.1 * reflectivity). */
int lili_extract_rot_device(lili_ctx* ctx, lili_cloud* full, lili_cloud* edge, lili_cloud* surf);
int lili_extract_livox_device(lili_ctx* ctx, lili_cloud* edge, lili_cloud* surf);

/* ---- local map: keyframe ring buffer + VoxelGrid + index ------------------------------------ */

/* pcl::VoxelGrid<PointT>::filter with leaf (leaf, leaf, leaf) on (x, y, z, aux) points: one centroid per occupied
 * voxel (aux averaged too), output ordered by ascending voxel index, f32 accumulation in input order
 * (L/src/BackendFusion.cpp:1488-1511, R/src/Preprocessing.cpp:502-506).  `counts` (optional, capacity ints) receives
 * the number of points merged into each output point.  Blocking. */
int lili_voxel_filter(lili_ctx* ctx, const lili_cloud* cloud, float leaf, lili_feature_out* out, int32_t* counts);
/* Keyframe ring buffer of one map kind (recent_surf_keyframes / recent_edge_keyframes, L:1407-1477):
 * push = transformCloud(features, pose) appended, oldest dropped beyond `width` (local_map_width);
 * commit = concatenate (L:1479-1483) + VoxelGrid(leaf) (L:1488-1492) + lili_map_set on the result (L:839-840),
 * all on the device.  n_raw / n_map (optional) receive the point counts before / after the filter. */
int lili_localmap_reset(lili_ctx* ctx, int kind);
int lili_localmap_push(lili_ctx* ctx, int kind, const lili_cloud* features, const double t[3], const double q[4], int width);
int lili_localmap_commit(lili_ctx* ctx, int kind, float leaf, double max_sq_radius, int64_t* n_raw, int64_t* n_map);
/* How the commits of this context were served: steps on the ring kept sorted by voxel (one keyframe popped / pushed since the last commit, the
 * reference's steady state, L/src/BackendFusion.cpp:1407-1477) against full rebuilds (first commit, several keyframes pending, another leaf).
 * The map is the same bit for bit either way (option "localmap_incremental" = 0 forces rebuilds). */
int lili_localmap_stats(lili_ctx* ctx, int32_t* incremental_commits, int32_t* full_commits);
/* How the VoxelGrid filters of more than 8192 points (lili_voxel_filter, lili_localmap_commit's rebuild, lili_frontend_frame's query filter) were served: a filter
 * whose leaf size has been seen before keeps its bounding box on the device and sorts by as many key bits as the previous one needed ("voxel_guess_bits", default 1);
 * the device checks the guess, a filter it does not hold for is repeated with the box measured (key_guess_misses).  The output never depends on it. */
int lili_voxel_filter_stats(lili_ctx* ctx, int32_t* key_guesses, int32_t* key_guess_misses);
/* The down-sampled local map of the last lili_localmap_commit as (x, y, z, aux) rows in map order (out->count = its size; at most
 * out->capacity rows are copied).  Blocking. */
int lili_localmap_get(lili_ctx* ctx, lili_feature_out* out);

/* ---- front-end frame: the whole per-scan chain of the odometry node in ONE call ------------------ */

/* Replaces the body of LidarOdometry::run for one Livox scan (L/src/LidarOdometry.cpp:652-707) together with the extraction node in front of
 * it (L/src/Preprocessing.cpp:219-401): extraction -> down_size_filter_surf (L:155, 320-322) -> scan-to-map iterations against the local map
 * of the last `width` frames (buildLocalMap L:280-303, downSampleCloud L:314-318, kd_tree_surf_last->setInputCloud L:490,
 * findCorrespondingSurfFeatures + LidarPlaneNormIncreFactor + HuberLoss L:352-413, 483-561) -> the frame joins the ring at the pose found
 * (transformCloud, L:292-297); the local map that includes it (concatenate + VoxelGrid + index) is built at the start of the next call, under that frame's extraction.
 * Everything between the scan and the returned pose stays in HBM: features, down-sampled queries, ring, map and pose never visit the host;
 * the host synchronises three times per frame, for the counts that size the next launches (ring merge; index build + query filter, whose launches are enqueued
 * behind the build's so that both outcomes come back together) and for the result.
 *   scan / curvature_offset / q_imu / livox : as lili_extract_livox
 *   match   : matcher parameters (LILI_VARIANT_FRONTEND for the reference's front end)
 *   t_pred, q_pred : poseInitialization's guess (L:415-441: constant-velocity extrapolation, caller side); a frame that is not matched
 *                    (n_iters = 0, or a map of fewer than 10 points, L:485-488) keeps it.
 * `res` receives the pose, the solver status of the last update and the sizes; res->stage_us (opt->want_timing) the host-side time stamps
 * of the stages in microseconds since the call began ([0] pending local map built and query filter enqueued, [1] queries known, [2] iterations enqueued,
 * [3] pose known).  lili_frontend_reset empties the ring (a new sequence). */
typedef struct lili_frontend_options {
    float leaf_query;   /* down_size_filter_surf: 0.4 (L/src/LidarOdometry.cpp:155) */
    float leaf_map;     /* down_size_filter_surf_map: 0.4 (L:156) */
    int width;          /* frames in the local map: 20 (L:290) */
    int n_iters;        /* outer iterations (association + Gauss-Newton update) per frame */
    int slot;           /* matcher slot used for the frame's queries and pose */
    int want_timing;
    int flags;          /* LILI_FRAME_* */
} lili_frontend_options;
/* The reference node's start-up (L/src/LidarOdometry.cpp:659-663, 283-289, 501-504), expressed per call so that the caller's frame counter decides:
 *   frame 0 (system not initialised: savePoses + checkInitialization only): n_iters = 0 and LILI_FRAME_PUSH_EMPTY — the pose is the given one and the ring
 *            receives an EMPTY keyframe (surf_frames[0] is the not-yet-filled surf_last_ds);
 *   frame 1 (pose_cloud_frame holds one pose): LILI_FRAME_SELF_MAP — the local map is the frame's own surf features (L:286-289), n_iters = 8 (L:501-502);
 *   later frames: flags = 0, n_iters = scan_match_cnt. */
enum { LILI_FRAME_SELF_MAP = 1, LILI_FRAME_PUSH_EMPTY = 2,
       /* round 6 — the frame call for a caller that keeps its own maps (the BACK end's matcher on one scan: BASELINE configs[0]):
        *   LILI_FRAME_EXTERNAL_MAP  match against the index(es) the caller set with lili_map_set; nothing joins the ring, no local map is built;
        *   LILI_FRAME_EDGES         (with EXTERNAL_MAP) the edge features are queries too (kind mask surf | edge, findCorrespondingCornerFeatures);
        *   leaf_query = 0           (with EXTERNAL_MAP) the surf features themselves are the queries (no down_size_filter_surf). */
       LILI_FRAME_EXTERNAL_MAP = 4, LILI_FRAME_EDGES = 8 };
typedef struct lili_frontend_result {
    double t[3], q[4];
    int gn_status;      /* of the last update; 0 when the frame was not matched */
    int matched;        /* 0: first frame of a sequence, or a map of fewer than 10 points (L:485-488): pose = prediction */
    int32_t n_edge, n_surf, n_query, n_map_raw, n_map;   /* features, down-sampled queries; ring points and map points of the map the frame was MATCHED AGAINST (whichever call built it;
                                                          * a LILI_FRAME_SELF_MAP frame: its own features and their filtered cloud; LILI_FRAME_EXTERNAL_MAP: the caller's map, twice) */
    double stage_us[8];
} lili_frontend_result;
int lili_frontend_frame(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* livox,
                        const lili_s2m_params* match, const lili_frontend_options* opt, const double t_pred[3], const double q_pred[4],
                        lili_frontend_result* res);
/* The same chain behind the LOAM-style extractor of the LiLi-OM-ROT package (scan / q_imu / q_lb / rot: as lili_extract_rot; R/src/Preprocessing.cpp:248-535 in front of
 * R/src/LidarOdometry.cpp:638-693 — that node is the Livox package's but for the point type).  One call per spinning-LiDAR scan. */
int lili_frontend_frame_rot(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* rot,
                            const lili_s2m_params* match, const lili_frontend_options* opt, const double t_pred[3], const double q_pred[4],
                            lili_frontend_result* res);
int lili_frontend_reset(lili_ctx* ctx);
/* lili_frontend_frame leaves the local map WITH the frame it has just pushed to the next call, which builds it under its own extraction (res->n_map_raw / n_map
 * describe the map the frame was matched against).  lili_frontend_flush builds it now — for a caller that reads the map between frames (lili_localmap_get,
 * lili_map_info) or ends a sequence.  Blocking; n_map_raw / n_map optional. */
int lili_frontend_flush(lili_ctx* ctx, const lili_s2m_params* match, const lili_frontend_options* opt, int32_t* n_map_raw, int32_t* n_map);

/* ---- back-end keyframe: everything between "a keyframe has arrived" and ceres::Solve in ONE call ---------------- */

/* Replaces, for one keyframe of the reference's back end (L/src/BackendFusion.cpp:830-980):
 *   buildLocalMapWithLandMark (L:1387-1484, steady state :1444-1476) — the keyframe whose pose the previous solve fixed (`join_*`, its LiDAR pose in the map frame
 *       t_join / q_join = q_po * q_bl, q_po * t_bl + t_po; NULL: nothing joins, e.g. the first call) is transformed and appended to both rings, the oldest leaves
 *       beyond opt->width;
 *   downSampleCloud (L:1486-1519) — both rings -> VoxelGrid(leaf_*_map) -> kd_tree_*_local_map->setInputCloud (L:839-840); the NEW keyframe's features
 *       (`new_surf`, `new_edge`, LiDAR frame) -> VoxelGrid(leaf_surf / leaf_edge) = surf_lasts_ds / edge_lasts_ds -> the queries of slots[n_slots - 1];
 *   findCorrespondingCornerFeatures / findCorrespondingSurfFeatures (L:919-936) — every keyframe of the window (slots[i], oldest first; the older slots keep the
 *       queries earlier calls gave them) at its association pose (t_assoc 3, q_assoc 4 values per slot: q * q_lb^-1, t - Q2 t_lb, L:929-930).
 * Rings, maps, indices, queries and correspondence records stay in HBM; afterwards the window is ready for lili_s2m_linearize_window (LidarWindowFactor),
 * lili_s2m_solve_lm_window or lili_marg_add_lidar.  n_res (optional): 2 per slot {surf, edge}.  Records and counts equal the calls one by one
 * (lili_localmap_push x 2, lili_localmap_commit x 2, lili_voxel_filter x 2, lili_s2m_set_queries x 2, lili_s2m_associate_window) bit for bit. */
typedef struct lili_backend_options {
    float leaf_surf, leaf_edge;           /* ds_filter_surf / ds_filter_edge: surf_ds 0.4, edge_ds 0.2 (L/config/config_fr_iosb.yaml:22-23) */
    float leaf_surf_map, leaf_edge_map;   /* ds_filter_surf_map / ds_filter_edge_map: the same values in the reference (L:491-494) */
    int width;                            /* local_map_width: 40 (L/config :16) */
    int want_timing;
    int join_slot;                        /* >= 0 (and join_surf = join_edge = NULL): the joining keyframe's features are the QUERIES of that matcher slot — the reference stores
                                           * surf_lasts_ds / edge_lasts_ds of a keyframe as its surf_frames / edge_frames entry (L:1505-1519, saveKeyFramesAndFactors), i.e. what an
                                           * earlier call left in the slot that was then the newest: the keyframe never leaves HBM.  < 0: not used */
} lili_backend_options;
typedef struct lili_backend_result {
    int32_t n_map_raw[2], n_map[2];       /* ring points and map points per kind {surf, edge} */
    int32_t n_query[2];                   /* the new keyframe's down-sampled features per kind */
    int32_t associated;                   /* 0: a map was missing (first keyframes), nothing associated */
    double stage_us[8];                   /* opt->want_timing: host time since the call began after [0] the surf side, [1] the edge side, [2] the associations */
} lili_backend_result;
int lili_backend_keyframe_prepare(lili_ctx* ctx, const lili_cloud* join_surf, const lili_cloud* join_edge, const double t_join[3], const double q_join[4],
                                  const lili_cloud* new_surf, const lili_cloud* new_edge, const int* slots, int n_slots, const double* t_assoc, const double* q_assoc,
                                  const lili_s2m_params* match, const lili_backend_options* opt, int32_t* n_res, lili_backend_result* res);

/* ---- scan-to-map matcher -------------------------------------------------------------------- */

/* Uploads the feature points of keyframe `slot` (surf_lasts_ds[idx] / edge_lasts_ds[idx],
 * L/src/BackendFusion.cpp:1536,1606).  Async when the cloud is already on the device. */
int lili_s2m_set_queries(lili_ctx* ctx, int slot, int kind, const lili_cloud* cloud);

/* Replaces findCorresponding{Surf,Corner}Features(idx, q, t) (L/src/BackendFusion.cpp:1531-1681,
 * R/src/BackendFusion.cpp:1394-1520, L/src/LidarOdometry.cpp:352-413): transform with (q_assoc,
 * t_assoc), exact 5-NN, line / plane fit, gates, weights.  Correspondence records stay on the device.
 * n_res (optional) receives the number of correspondences — passing it makes the call blocking. */
int lili_s2m_associate(lili_ctx* ctx, int slot, int kind, const double t_assoc[3], const double q_assoc[4],
                       const lili_s2m_params* params, int* n_res);
/* The same for the keyframes of the sliding window in one call (L/src/BackendFusion.cpp:919-936 calls both finders per keyframe): t_assoc = 3,
 * q_assoc = 4 values per slot; n_res (optional) = 2 per slot, {surf, edge} (0 for a kind outside kind_mask).  Both kinds of a slot share a launch,
 * the slots run concurrently, and with n_res the call synchronises once.  Records equal lili_s2m_associate per slot and kind. */
int lili_s2m_associate_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t_assoc, const double* q_assoc,
                              const lili_s2m_params* params, int* n_res);

/* Replaces N x (AutoDiffCostFunction::Evaluate + loss corrector) and the J^T J / J^T r accumulation
 * (L/include/factors/LidarKeyframeFactor.h:12-139, L/src/MarginalizationFactor.cpp:3-29,44-70) for the
 * records of `slot` selected by `kind_mask`, at the body pose (t, q).
 *   gram[64] : row-major 8x8, sum over residuals of [J r]^T [J r] after robustification, with J the
 *              1x7 GLOBAL Jacobian ordered (t0,t1,t2,qw,qx,qy,qz) and r the residual;
 *   cost     : sum of 1/2 rho(r^2);  counts[0] = surf residuals, counts[1] = edge residuals.
 * Blocking. */
int lili_s2m_linearize(lili_ctx* ctx, int slot, int kind_mask, const double t[3], const double q[4],
                       const lili_s2m_params* params, double gram[64], double* cost, int counts[2]);
/* The same for several slots at once — ONE evaluation of the joint sliding window (a Gram per keyframe; ceres::Solve evaluates the window up to 15
 * times per keyframe, L/src/BackendFusion.cpp:919-992): t = 3, q = 4, gram = 64, cost = 1 (optional), counts = 2 (optional) values per slot, in the
 * order of `slots`.  The slots run concurrently and the call synchronises once; results equal lili_s2m_linearize per slot bit for bit.  Blocking. */
int lili_s2m_linearize_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q,
                              const lili_s2m_params* params, double* gram, double* cost, int* counts);

/* Copy-out of the ordered correspondence lists (the vec_surf_cur_pts / vec_surf_normal /
 * vec_surf_scores and vec_edge_cur_pts / vec_edge_match_j / vec_edge_match_l of the reference).
 * Arrays may be NULL; capacity in records; *n_out = number of records available.  Blocking. */
int lili_s2m_get_surf_records(lili_ctx* ctx, int slot, size_t capacity, int32_t* query_index, float* cur_pt /*3*/,
                              float* normal /*3, weight-scaled*/, float* neg_oa_dot_norm, double* score, size_t* n_out);
int lili_s2m_get_edge_records(lili_ctx* ctx, int slot, size_t capacity, int32_t* query_index, float* cur_pt /*3*/,
                              float* pt_a /*3*/, float* pt_b /*3*/, float* s, size_t* n_out);
/* Per-query neighbour lists of the last associate (requires lili_set_debug(ctx, 1)): idx/d2 are n_q x 5. */
int lili_s2m_get_neighbors(lili_ctx* ctx, int slot, int kind, size_t n_q, int32_t* idx, float* d2);

/* ---- device-resident outer iterations (no host round trip) ----------------------------------- */

/* Body pose of `slot` kept in device memory. */
int lili_s2m_pose_set(lili_ctx* ctx, int slot, const double t[3], const double q[4]);
int lili_s2m_pose_get(lili_ctx* ctx, int slot, double t[3], double q[4], int* gn_status /*0 ok, 1 singular*/);

/* The last Gauss-Newton step of `slot`: delta[6] = (dt[3], rotation vector[3]) of the local parameterisation, the number of updates since
 * lili_s2m_pose_set and the status of the last one (0 ok, 1 = normal matrix not positive definite: pose left unchanged, 2 = a multi-GPU
 * exchange gave up).  The device loops take plain Gauss-Newton steps (one per re-association, like one accepted step of the reference's
 * ceres::Solve, L/src/BackendFusion.cpp:984-992); callers that need Ceres' step acceptance drive lili_s2m_linearize from their own solver
 * (include/lili_ceres_adapter.h) — this call lets the others at least see how far a step went.  Blocking. */
int lili_s2m_last_step(lili_ctx* ctx, int slot, double delta[6], int* n_updates, int* gn_status);

/* ---- Levenberg-Marquardt on the device (fixed correspondences) -------------------------------------
 * The reference's inner loop: ceres::Solve on the residual blocks of the last association (L/src/BackendFusion.cpp:984-992: DENSE_QR,
 * max_num_iterations = max_num_iter, otherwise Ceres 2.0 defaults — TRUST_REGION / LEVENBERG_MARQUARDT, initial radius 1e4, Jacobi scaling,
 * monotonic steps, function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8).  lili_s2m_solve_lm runs that loop for
 * the lidar blocks of ONE slot — both kinds in kind_mask, robustified by params->loss — as ONE persistent launch: every iteration evaluates
 * all records at the candidate pose (robust cost + Gram), the workgroups exchange their partials inside the launch and each takes the same
 * accept / reject decision; the pose of the slot ends at the last accepted point — a candidate that triggers the parameter or the function
 * tolerance is NOT taken (Ceres returns before HandleSuccessfulStep).  Decisions and final pose equal the oracle's restatement
 * of Ceres' loop on per-residual rows (oracle/lo_window.py::ceres_lm; tests/test_lm_gpu.py).  Needs lili_s2m_associate* first.
 * options == NULL: the defaults above with max_iterations = 15.  summary (host memory, optional): filled after a synchronisation of the
 * context's stream; with summary == NULL the call is asynchronous.  Not to be overlapped with other persistent launches of the same
 * device beyond what lili_s2m_solve_lm_window does itself (every workgroup of the launch has to be resident; waits are bounded and end in
 * LILI_LM_STALLED rather than a hang). */
#define LILI_LM_MAX_LOG 32
enum { LILI_LM_MAX_ITERATIONS = 0, LILI_LM_GRADIENT_TOLERANCE = 1, LILI_LM_PARAMETER_TOLERANCE = 2, LILI_LM_FUNCTION_TOLERANCE = 3,
       LILI_LM_STALLED = 4, LILI_LM_NUMERICAL_FAILURE = 5 /* a non-positive pivot, or 5 consecutive invalid steps (Ceres: FAILURE) */,
       LILI_LM_MIN_RADIUS = 6 /* trust-region radius <= min_radius (Ceres: CONVERGENCE, "minimum trust region radius reached") */ };
typedef struct lili_lm_options {
    int32_t max_iterations;
    int32_t reserved_;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
} lili_lm_options;
typedef struct lili_lm_iteration {       /* one evaluated candidate */
    double cost, new_cost, rho, radius, step_norm;
    int32_t accepted, iteration;
} lili_lm_iteration;
typedef struct lili_lm_summary {
    int32_t iterations, successful_steps, termination, n_logged;
    int32_t n_surf, n_edge;              /* correspondences the rows were scaled with (ROT: num / N) */
    double initial_cost, final_cost, final_radius;
    lili_lm_iteration it[LILI_LM_MAX_LOG];
} lili_lm_summary;
void lili_lm_default_options(lili_lm_options* opt);
int lili_s2m_solve_lm(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, const lili_lm_options* options,
                      lili_lm_summary* summary);
/* The same for several slots (the keyframes of the sliding window) concurrently, one launch per slot on forked streams; summaries:
 * n_slots entries (optional).  The joint window of the reference couples the keyframes through the IMU factors, which stay with the
 * caller's solver (include/lili_ceres_adapter.h); this call is for per-keyframe refinements and for the front-end. */
int lili_s2m_solve_lm_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params,
                             const lili_lm_options* options, lili_lm_summary* summaries);

/* One outer iteration, first half (async): re-associate at the device pose (association transform
 * Q2 = Q*q_lb^-1, T2 = T - Q2*t_lb as in L/src/BackendFusion.cpp:929-930), linearise, and reduce this
 * rank's partial into d_gram (DEVICE pointer to LILI_GRAM_DOUBLES doubles owned by the caller:
 * [0..63] Gram, [64] cost, [65] n_surf, [66] n_edge).  Between the two halves a multi-GPU caller
 * all-reduces d_gram (sum) over ranks on the same stream. */
int lili_s2m_accumulate(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, double* d_gram);
/* The same in two steps, for callers that shard the queries of one scan over several GPUs AND use the
 * ROT residual scaling num / N (R/src/BackendFusion.cpp:843,861), where N must be the GLOBAL count:
 *   lili_s2m_associate_dev   -> per-block counts of this rank in device memory
 *   lili_s2m_counts_export   -> this rank's {n_surf, n_edge} into a caller-owned DEVICE int32[2]   (async)
 *   [caller: all-reduce(sum) that buffer over ranks on the same stream]
 *   lili_s2m_counts_import   -> the next linearize_dev scales with that buffer (read by its kernels, no copy:
 *                               keep it valid and unchanged until that call's work has run)            (async)
 *   lili_s2m_linearize_dev   -> d_gram as above.                                                          */
int lili_s2m_associate_dev(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params);
int lili_s2m_counts_export(lili_ctx* ctx, int slot, int32_t* d_counts);
int lili_s2m_counts_import(lili_ctx* ctx, int slot, const int32_t* d_counts);
int lili_s2m_linearize_dev(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, double* d_gram);
/* Second half (async): one Gauss-Newton step on the 6-dof local parameterisation
 * (ceres::QuaternionParameterization plus-Jacobian and Plus()), pose updated in device memory. */
int lili_s2m_gn_update(lili_ctx* ctx, int slot, const double* d_gram);

/* The sharded iteration loop in ONE host call (queries block-sharded over the GPUs of a node, map replicated):
 * n_iters x [ associate_dev -> (count-scaled flavours only: counts_export -> all-reduce(int32[2]) -> counts_import) ->
 *             linearize_dev(d_gram) -> all-reduce(f64[LILI_GRAM_DOUBLES]) -> gn_update ]
 * everything enqueued on the context's stream, so that no host code sits between the kernels and the two collectives.
 * `allreduce` has the signature of ncclAllReduce (RCCL: sendbuff, recvbuff, count, ncclDataType_t, ncclRedOp_t, ncclComm_t,
 * hipStream_t) and is called in place with datatype 2 (ncclInt32) / 8 (ncclFloat64) and op 0 (ncclSum); the library does not
 * link RCCL — the caller hands over the function and its communicator (lili_om_amd/rccl.py does it with the librccl.so that
 * PyTorch loaded).  allreduce == NULL runs the same staged launches without collectives (one rank).  restart_every /
 * restart_slot as in lili_s2m_iterate_restart.  d_counts: DEVICE int32[2], d_gram: DEVICE double[LILI_GRAM_DOUBLES],
 * both caller-owned and valid until the stream has drained.  Every rank then holds the same pose (same reduced Gram, same
 * GN step), no broadcast is needed.  With allreduce == lili_p2p_allreduce (below) the exchange is folded into the count kernel and
 * into the reduce + Gauss-Newton kernel: associate, counts + exchange, linearise, reduce + exchange + GN = 4 launches per iteration. */
typedef int (*lili_allreduce_fn)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, void* stream);
int lili_s2m_iterate_sharded(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int restart_every,
                             int restart_slot, lili_allreduce_fn allreduce, void* comm, int32_t* d_counts, double* d_gram);
/* The sliding window across ranks (BASELINE configs[4]: the reference evaluates the lidar blocks of ALL keyframes of the window per solver
 * evaluation, L/src/BackendFusion.cpp:919-992).  Every rank holds its shard of every slot's queries (set_queries with the shard) and the
 * whole map; per evaluation ONE exchange of n_slots x LILI_GRAM_DOUBLES doubles makes every rank hold the same n Gram records bit for bit
 * (lili_p2p: folded into the reduction launch; any other lili_allreduce_fn: one call on the n x 72 doubles; NULL: one rank).
 *   lili_s2m_counts_window_sharded     after the slots' associations (lili_s2m_associate / _associate_dev per slot): the GLOBAL correspondence
 *                                      counts of every slot in ONE exchange of 2 n int32 into d_counts (DEVICE, [surf, edge] per slot; kept valid
 *                                      by the caller) — the count-scaled ROT flavour needs them, the others may skip the call.  They stay
 *                                      in force for the slots' linearisations until the next association.  Async.
 *   lili_s2m_linearize_window_dev      the n records at the slots' DEVICE poses into d_gram (DEVICE, n x LILI_GRAM_DOUBLES).  Async.
 *   lili_s2m_linearize_window_sharded  the same at host poses t (3 per slot) / q (4 per slot), blocking, records copied out like
 *                                      lili_s2m_linearize_window: one evaluation of the caller's solver.
 *   lili_s2m_iterate_window_sharded    n_iters x [associate every slot, counts exchange (count-scaled flavours), linearise every slot, ONE
 *                                      reduction launch + exchange + Gauss-Newton update of every slot]: the lidar-only registration of
 *                                      every keyframe as in lili_s2m_iterate_window, 2 exchanges per iteration whatever n_slots is. */
int lili_s2m_counts_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, lili_allreduce_fn allreduce, void* comm, int32_t* d_counts);
int lili_s2m_linearize_window_dev(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params,
                                  lili_allreduce_fn allreduce, void* comm, double* d_gram);
int lili_s2m_linearize_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q,
                                      const lili_s2m_params* params, lili_allreduce_fn allreduce, void* comm, double* d_gram,
                                      double* gram, double* cost, int* counts);
int lili_s2m_iterate_window_sharded(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters,
                                    lili_allreduce_fn allreduce, void* comm, int32_t* d_counts, double* d_gram);
/* Peer-to-peer all-reduce for the two records of the sharded loop (the loop being sharded is
 * L/src/BackendFusion.cpp:1536,1606): every rank owns a mailbox in its HBM that all peers map through hipIpc; one small kernel per
 * all-reduce stores this rank's record into every peer's mailbox over xGMI, waits for the peers' records in its own, and adds them IN
 * RANK ORDER — so all ranks hold bit-identical sums.  lili_p2p_allreduce has the signature of ncclAllReduce (lili_allreduce_fn) with
 * comm = the lili_p2p*; records of at most 640 elements (a window of 8 Gram records is 576 doubles), datatype 2 (int32) or 8 (f64), op 0 (sum).
 *   lili_p2p_create   one per rank (rank, world <= 16), on the context's device
 *   lili_p2p_handle   this rank's mailbox handle (LILI_P2P_HANDLE_BYTES, opaque) — the caller all-gathers the handles (MPI,
 *                     torch.distributed, a file ...) and hands all of them, in rank order, to
 *   lili_p2p_connect  (maps the peers' mailboxes; one process per GPU, or several processes on one GPU)
 *   lili_p2p_status   0 = ok, 1 = a wait for a peer gave up (the record is then undefined, the slot's gn_status is 2).  The failure is
 *                     sticky and contagious: the rank publishes nothing any more and raises the failure word of every peer, whose
 *                     next exchange fails at its first look; lili_s2m_iterate_sharded then returns LILI_E_STATE on every rank.
 *   lili_p2p_set_timeout  how long an exchange may wait for a peer (device time, default 10 s).  The FIRST exchange absorbs the ranks'
 *                     start-up skew (code-object load, data loading): keep it generous, or barrier the control plane first. */
#define LILI_P2P_HANDLE_BYTES 64
/* Slot-per-rank window (the split of the sliding window that scales, SURVEY §8e / L/src/BackendFusion.cpp:919-980): keyframe slots[i] lives on rank owner[i] at FULL
 * size (all its queries; the other ranks hold only its pose slot) and every evaluation ends with ONE exchange of the n x LILI_GRAM_DOUBLES doubles in which each record has
 * a single non-zero contributor — an all-gather carried by the rank-order sum of `allreduce` (lili_p2p_allreduce: inside the reduction launch).  No count exchange: the
 * owner's count is the global one.  _linearize_: records of one evaluation at the slots' DEVICE poses into d_gram (device, n x LILI_GRAM_DOUBLES; identical on every
 * rank, slot i's record = what lili_s2m_linearize_window returns for it on the owner, a -0.0 entry read as +0.0); _iterate_: n_iters x (association of the owned
 * keyframes, their linearisation, exchange, Gauss-Newton update of EVERY slot on every rank).  Every rank must have set the same pose in every slot.  Async. */
int lili_s2m_linearize_window_gather(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, const int* owner, int rank,
                                     lili_allreduce_fn allreduce, void* comm, double* d_gram);
int lili_s2m_iterate_window_gather(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters, const int* owner, int rank,
                                   lili_allreduce_fn allreduce, void* comm, double* d_gram);
/* _linearize_ at the body poses (t[3 n], q[4 n]) of the call instead of the device poses, records copied out as lili_s2m_linearize_window returns them (gram 64 n, cost n,
 * counts 2 n; one synchronisation): one solver evaluation of the slot-per-rank window (ceres::CostFunction::Evaluate of the joint window, L/src/BackendFusion.cpp:919-992;
 * include/lili_ceres_adapter.h LidarWindowFactor::gather_over_ranks). */
int lili_s2m_linearize_window_gather_at(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const double* t, const double* q, const lili_s2m_params* params,
                                        const int* owner, int rank, lili_allreduce_fn allreduce, void* comm, double* d_gram, double* gram, double* cost, int32_t* counts);

typedef struct lili_p2p lili_p2p;
int lili_p2p_create(lili_ctx* ctx, int rank, int world, lili_p2p** out);
int lili_p2p_handle(lili_p2p* comm, void* handle);
int lili_p2p_connect(lili_p2p* comm, const void* all_handles);
int lili_p2p_allreduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, void* stream);
int lili_p2p_status(lili_p2p* comm);
int lili_p2p_set_timeout(lili_p2p* comm, double seconds);
void lili_p2p_destroy(lili_p2p* comm);
/* Convenience: n_iters x (accumulate + gn_update) on an internal buffer.  Async. */
int lili_s2m_iterate(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters);

/* The INNER iteration of the reference back-end (ceres::Solve on fixed correspondences, L/src/BackendFusion.cpp:984-992; the
 * front-end's L/src/LidarOdometry.cpp:533): n_iters x [linearise the records of the LAST association at the device pose, reduce,
 * Gauss-Newton update] — no re-association.  One launch per iteration.  want_cost != 0 also evaluates sum 1/2 rho (slot 64 of the
 * internal record), as an LM caller needs it.  Async. */
int lili_s2m_iterate_inner(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int want_cost);

/* lili_s2m_iterate for several slots at once (the keyframes of one sliding window, or several sensors): every slot
 * runs its own chain on its own stream, forked from / joined to the context's stream, so the latency-bound linearise /
 * reduce launches of one slot overlap the association of another.  Same results as iterating the slots one after the
 * other; all slots share the map index.  Async. */
int lili_s2m_iterate_window(lili_ctx* ctx, const int* slots, int n_slots, int kind_mask, const lili_s2m_params* params, int n_iters);
/* Profiling aid: with LILI_DEBUG bit 256 set, kernels stamp a 100 MHz device clock at their phases; returns the 16 stamps. */
int lili_s2m_debug_times(lili_ctx* ctx, int slot, long long out[16]);
/* Async device-to-device copy of the body pose of src_slot into dst_slot (no host round trip). */
int lili_s2m_pose_copy(lili_ctx* ctx, int dst_slot, int src_slot);
/* lili_s2m_iterate that re-initialises the pose from restart_slot before iterations 0, restart_every, 2*restart_every...
 * ("one scan registration = restart_every GN iterations"; restart_every = 0 disables).  If assoc_ms is non-NULL the
 * association launches are bracketed by HIP events on the context's stream, their total duration (ms) is returned and
 * the call synchronises; with NULL it is async like lili_s2m_iterate. */
int lili_s2m_iterate_restart(lili_ctx* ctx, int slot, int kind_mask, const lili_s2m_params* params, int n_iters, int restart_every,
                             int restart_slot, float* assoc_ms);

/* Host-side helper used by the ceres adapter and the host LM: Gauss-Newton step from a host Gram. */
int lili_gn_step_host(const double gram[64], double t[3], double q[4], double delta[6]);

/* Square-root form of a Gram record for ceres (see include/lili_ceres_adapter.h): a 9-residual block whose
 * J^T J, J^T r and cost equal those of the N robustified lidar residuals the Gram was reduced from. */
int lili_gram_to_factor(const double gram[64], double cost, double residuals[9], double jacobian[63]);

/* ---- callers / data formats either side of the path ------------------------- */

/* Livox CustomMsg points -> the PointXYZINormal cloud the Livox extractor consumes; replaces livoxLidarHandler
 * (L/src/FormatConvert.cpp:11-35).  `points` = point_num records of the serialised livox_ros_driver/CustomPoint
 * (little endian: uint32 offset_time, float32 x, y, z, uint8 reflectivity, tag, line; `stride` >= 19 bytes).
 * Output: pcl::PointXYZINormal records (48 B: x y z 1.0 | normal 0 0 0 0 | intensity curvature 0 0) with
 *   intensity = line + 0.1 * (float(offset_time) / float(offset_time of the LAST point)),  curvature = 0.1 * reflectivity
 * (float / double expression widths of the reference).  in_mem / out_mem: LILI_MEM_HOST or LILI_MEM_DEVICE; the output
 * can be handed to lili_extract_livox as {data = out, stride = 48, aux_offset = 32, curvature_offset = 36}.  Blocking for
 * host memory, asynchronous on the context's stream when both sides are device memory. */
int lili_livox_custom_to_cloud(lili_ctx* ctx, const void* points, size_t point_num, size_t stride, int in_mem, void* out, int out_mem);

/* Gyro integration over one scan, host side (L/src/Preprocessing.cpp:129-171 processIMU / solveRotation, 176-191
 * imuHandler state, 232-234 NaN reset, 403 identity reset; deltaQ = utils/math_tools.h:125-138): mid-point rule on the
 * un-normalised small-angle quaternion, linear interpolation of the rate at the scan boundary.  The caller keeps the IMU
 * samples (stamps[n] seconds, gyr[n*3] rad/s, in arrival order, n may grow between calls) and one lili_imu_state per
 * sensor; q_out (w,x,y,z) is the q_imu argument of lili_extract_livox / lili_extract_rot for the scan ending at
 * t_scan_next.  Zero-initialise the state (or lili_imu_reset) before the first call. */
typedef struct lili_imu_state {
    int64_t idx;        /* idx_imu: first sample not yet consumed */
    double t_cur;       /* current_time_imu (< 0: unset) */
    double gyr0[3];     /* gyr_0 */
    int32_t first;      /* first_imu seen */
    int32_t reserved;
} lili_imu_state;
void lili_imu_reset(lili_imu_state* st);
int lili_imu_integrate(lili_imu_state* st, const double* stamps, const double* gyr, size_t n, double t_scan_next, double q_out[4]);

/* Lidar contribution of one keyframe to MarginalizationInfo's A, b (L/src/MarginalizationFactor.cpp:3-29, 151-174):
 * the reference evaluates every lidar residual again and adds J_i^T J_j / J_i^T r per block with rightCols(3) of the
 * 1x4 quaternion Jacobian; the same sums are rows / columns {0,1,2} (t) and {4,5,6} (q: x,y,z) of the Gram record
 * and its column 7.  A is a dense pos x pos matrix with leading dimension ld (symmetric: both triangles are
 * updated, so row- and column-major callers are served alike), b has pos entries; idx_t / idx_q are
 * parameter_block_idx of the keyframe's translation / rotation block.  Host function. */
int lili_marg_add_lidar(const double gram[64], double* A, size_t ld, double* b, size_t pos, size_t idx_t, size_t idx_q);

#ifdef __cplusplus
}
#endif
#endif /* LILI_HIP_H */
