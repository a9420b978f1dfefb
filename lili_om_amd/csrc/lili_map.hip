// C-ABI of liblili_hip.so, map index part (include/lili_hip.h: lili_map_set*, lili_map_focus, lili_map_info, lili_map_density, lili_map_build_stats).
// Host code only; the kernels live in lili_s2m.hip.  Replaces pcl::KdTreeFLANN::setInputCloud at L/src/LidarOdometry.cpp:490, L/src/BackendFusion.cpp:1258-1259.
#include "lili_launch.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
// --------------------------------------------------------------------------------------------
// map index
// --------------------------------------------------------------------------------------------
// Uniform-grid index of the caller's cloud `src` (read where it lies: device memory, or the staging copy of a host cloud) with cells of edge `cell` (grown if the
// bounding box needs more than max_cells cells): count (one atomic per run of equal cells, the returned value = the point's rank), in-place single-pass scan,
// atomic-free scatter.
// scratch words of a map build (ctx->misc): [0, 8192) bounding-box banks, [8192, 16384) density banks, then the sticky error word of the look-back scans
// then the status words of the two single-pass scans of a build (cell table, super-rows; <= 8192 tiles each).
// ONE memset arms all of it (round 4: five small fills and a host-to-device copy per build were ~25 us of serialised launches).
constexpr size_t kMiscBytes = 2 * 64 * 128 + 256;
constexpr size_t kScanStatusTiles = 8192, kScanStatusBytes = (kScanStatusTiles + 2) * sizeof(unsigned long long) + 112 /* -> a multiple of 128 */;
constexpr size_t kMiscTotal = kMiscBytes + 2 * kScanStatusBytes;
static_assert(kMiscTotal == kMiscAlloc, "ctx->misc is allocated in lili_ctx_create");
static unsigned* scan_err_word(lili_ctx* ctx) { return reinterpret_cast<unsigned*>(ctx->misc.as<char>() + 2 * 64 * 128); }
static unsigned long long* scan_status(lili_ctx* ctx, int which) { return reinterpret_cast<unsigned long long*>(ctx->misc.as<char>() + kMiscBytes + (size_t)which * kScanStatusBytes); }
static int build_grid(lili_ctx* ctx, MapIndex& m, const SrcCloud& src, const double mn[3], const double mx[3], double cell, int reach, DevBuf& sorted, DevBuf& aux_sorted,
                      DevBuf& cell_start, DevBuf& cell_start9, GridView& out, int64_t& n_cells, double& cell_used, unsigned long long* d_rank_sum, bool box_check, float touch_cells, bool narrow /* 8-bit count table (k_cell_count_narrow) */,
                      bool status_armed /* the scans' status words (ctx->misc) are still zero from the build's one memset */, bool want_srows /* super-row copy (if the options and the sizes allow it) */, const void* zeroed_p = nullptr, size_t zeroed_bytes = 0 /* the caller already cleared this much of cell_start (while the bounding box travelled) */) {
    const int n = (int)m.n;
    int64_t nx, ny, nz;
    for (;;) {
        nx = (int64_t)std::floor((mx[0] - mn[0]) / cell) + 1;
        ny = (int64_t)std::floor((mx[1] - mn[1]) / cell) + 1;
        nz = (int64_t)std::floor((mx[2] - mn[2]) / cell) + 1;
        double total = (double)nx * (double)ny * (double)nz;
        if (total <= (double)ctx->max_cells) break;
        cell *= std::cbrt(total / (double)ctx->max_cells) * 1.02;   // coarser cells stay exact, only slower
    }
    cell_used = cell;
    n_cells = nx * ny * nz;
    GridView g{};
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2]; g.inv_cell = 1.0 / cell; g.cell = 1.0 / g.inv_cell;
    g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz; g.n_points = n; g.reach = reach;
    const int64_t nc = n_cells;
    // ONE cell array: counts -> (in-place exclusive scan) -> cell_start; the atomic of the count pass also hands every point its rank
    HIPCHK(cell_start.ensure((size_t)(nc + 1) * sizeof(int)));
    HIPCHK(m.pt_cell.ensure((size_t)n * sizeof(int)));
    // super-rows: the sorted array continues with the 3x3-row copy of the box (<= 9n entries); positions stay 32-bit byte offsets, so 10n < 2^28.
    // The box: the whole grid, or the cells within the focus radius (+ one gate radius and a cell) of the focus point.
    int b0[3] = {0, 0, 0}, b1[3] = {(int)nx - 1, (int)ny - 1, (int)nz - 1};
    if (ctx->focus_radius > 0) {
        const int64_t dims[3] = {nx, ny, nz};
        for (int k = 0; k < 3; k++) {
            const double lo = (ctx->focus[k] - ctx->focus_radius - mn[k]) / cell - 2.0, hi = (ctx->focus[k] + ctx->focus_radius - mn[k]) / cell + 2.0;
            b0[k] = (int)std::min(std::max(std::floor(lo), 0.0), (double)(dims[k] - 1));
            b1[k] = (int)std::min(std::max(std::floor(hi), (double)b0[k]), (double)(dims[k] - 1));
        }
    }
    g.bx0 = b0[0]; g.by0 = b0[1]; g.bz0 = b0[2];
    g.bnx = b1[0] - b0[0] + 1; g.bny = b1[1] - b0[1] + 1; g.bnz = b1[2] - b0[2] + 1;
    const int64_t nc9 = (int64_t)g.bnx * g.bny * g.bnz, rows9 = (int64_t)g.bny * g.bnz;
    const bool srows = want_srows && ctx->super_rows && (int64_t)n * 10 < (1ll << 28) && nc9 + 2 < (1ll << 31);
    const size_t n_all = srows ? (size_t)n * 10 : (size_t)n;
    HIPCHK(sorted.ensure((n_all + 8) * sizeof(float4)));      // + slack: the super-row walk loads whole chunks of four, one of them past the run's end
    if (m.has_aux) HIPCHK(aux_sorted.ensure(n_all * sizeof(float)));
    if (srows) { HIPCHK(cell_start9.ensure((size_t)(nc9 + 2) * sizeof(int))); HIPCHK(m.row9.ensure((size_t)(rows9 + 2) * sizeof(int))); }
    const int nb_scan = nblocks(nc, 2048);
    HIPCHK(m.block_sums.ensure((size_t)(nb_scan + 2) * sizeof(unsigned long long)));
    const size_t nc_pad = ((size_t)nc + 16383) / 16384 * 16384;      // whole scan tiles
    if (narrow) {
        HIPCHK(m.cell_tmp.ensure(nc_pad));
        if (!(m.cell_tmp.p == zeroed_p && nc_pad <= zeroed_bytes)) HIPCHK(hipMemsetAsync(m.cell_tmp.p, 0, nc_pad, ctx->stream));
        hipLaunchKernelGGL(k_cell_count_narrow, dim3(nblocks(n, kBlock)), dim3(kBlock), 0, ctx->stream, src, n, g, m.cell_tmp.as<unsigned>(), m.pt_cell.as<unsigned char>(), d_rank_sum,
                           box_check ? 1 : 0, touch_cells);
    } else {
        if (!(cell_start.p == zeroed_p && (size_t)nc * sizeof(int) <= zeroed_bytes)) HIPCHK(hipMemsetAsync(cell_start.p, 0, (size_t)nc * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_cell_count, dim3(nblocks(n, kBlock)), dim3(kBlock), 0, ctx->stream, src, n, g, cell_start.as<int>(), m.pt_cell.as<int>(), d_rank_sum, box_check ? 1 : 0, touch_cells);
    }
    if (ctx->scan_lookback) {      // one pass over the cell array (status words: one per 16384-cell tile)
        const int nb_lb = nblocks(nc, 16384);
        unsigned long long* st = (size_t)nb_lb <= kScanStatusTiles ? scan_status(ctx, 0) : m.block_sums.as<unsigned long long>();
        if (!(status_armed && st == scan_status(ctx, 0))) HIPCHK(hipMemsetAsync(st, 0, (size_t)(nb_lb + 2) * sizeof(unsigned long long), ctx->stream));
        if (narrow) hipLaunchKernelGGL(k_scan_lookback_t<true>, dim3(nb_lb), dim3(kBlock), 0, ctx->stream, cell_start.as<int>(), m.cell_tmp.as<unsigned char>(), nc, st, scan_err_word(ctx));
        else hipLaunchKernelGGL(k_scan_lookback_t<false>, dim3(nb_lb), dim3(kBlock), 0, ctx->stream, cell_start.as<int>(), (const unsigned char*)nullptr, nc, st, scan_err_word(ctx));
    } else {
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nb_scan), dim3(kBlock), 0, ctx->stream, cell_start.as<int>(), nc, m.block_sums.as<int>());
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, ctx->stream, m.block_sums.as<int>(), nb_scan);
        hipLaunchKernelGGL(k_scan_apply, dim3(nb_scan), dim3(kBlock), 0, ctx->stream, cell_start.as<int>(), nc, m.block_sums.as<int>(), cell_start.as<int>());
    }
    if (narrow) hipLaunchKernelGGL(k_scatter_t<unsigned char>, dim3(8 * nblocks(nblocks(n, kBlock), 8)), dim3(kBlock), 0, ctx->stream, src, n, g, m.pt_cell.as<unsigned char>(), cell_start.as<int>(),
                                   sorted.as<float4>(), m.has_aux ? aux_sorted.as<float>() : nullptr);
    else hipLaunchKernelGGL(k_scatter_t<int>, dim3(8 * nblocks(nblocks(n, kBlock), 8)), dim3(kBlock), 0, ctx->stream, src, n, g, m.pt_cell.as<int>(), cell_start.as<int>(),
                            sorted.as<float4>(), m.has_aux ? aux_sorted.as<float>() : nullptr);
    if (srows) {      // first positions of the super-rows (populations -> scan), positions of the super cells, then the copy
        hipLaunchKernelGGL(k_rowtot9, dim3(nblocks(rows9, kBlock)), dim3(kBlock), 0, ctx->stream, cell_start.as<int>(), g, m.row9.as<int>());
        const int nb_lb = nblocks(rows9, 16384);
        unsigned long long* st = (size_t)nb_lb <= kScanStatusTiles ? scan_status(ctx, 1) : m.block_sums.as<unsigned long long>();
        if (!(status_armed && st == scan_status(ctx, 1))) HIPCHK(hipMemsetAsync(st, 0, (size_t)(nb_lb + 2) * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_scan_lookback_t<false>, dim3(nb_lb), dim3(kBlock), 0, ctx->stream, m.row9.as<int>(), (const unsigned char*)nullptr, rows9, st, scan_err_word(ctx));
        hipLaunchKernelGGL(k_start9, dim3(nblocks(g.bnx, 64), nblocks(g.bny, 4 * 8), (unsigned)g.bnz), dim3(256), 0, ctx->stream, cell_start.as<int>(), g, m.row9.as<int>(),
                           cell_start9.as<int>());
        {   // stretches per wave: more than one only where the index has so many (mostly empty) stretches that a wave per stretch is launch-bound — the fine index of
            // a dense map: 1.39 M stretches, 18 % of them populated (the walls at the ends of x put a few points into every stretch there) -> 8 per wave
            const int64_t n_str = (int64_t)nblocks(g.bnx, 64) * g.bny * g.bnz;
            int spw = 1;
            while (spw < 64 && n_str / (spw * 2) >= 131072) spw *= 2;
            const int64_t n_tiles = (int64_t)nblocks(g.bnx, 64) * nblocks(g.bny, 4) * nblocks(g.bnz, 4);
            hipLaunchKernelGGL(k_scatter9, dim3((unsigned)((n_tiles + spw - 1) / spw)), dim3(1024), 0, ctx->stream, cell_start.as<int>(), g, cell_start9.as<int>(),
                               sorted.as<float4>(), m.has_aux ? aux_sorted.as<float>() : nullptr, spw);
        }
    }
    HIPCHK(hipGetLastError());
    g.pts = sorted.as<float4>();
    g.aux = m.has_aux ? aux_sorted.as<float>() : nullptr;
    g.cell_start = cell_start.as<int>();
    g.cell_start9 = srows ? cell_start9.as<int>() : nullptr;
    out = g;
    return LILI_OK;
}

int lili_map_set(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius) { return lili_map_set_hinted(ctx, kind, cloud, max_sq_radius, nullptr, false); }

}  // extern "C"

// where the box of a build comes from
enum BoxSource { kBoxMeasure = 0,   // a bounding-box pass over the cloud and a read-back before the grid exists
                 kBoxGiven,         // the caller's (lili_localmap_commit: the centroids' box travels with their count), or the true box of a build whose guess failed
                 kBoxGuess };       // the previous build's true box + a margin, checked against this cloud's true box at the build's final read-back
static int map_set_impl(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius, const unsigned* box6, bool allow_guess);
int lili_map_set_hinted(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius, const unsigned* box6, bool /*in_place: every device cloud is read in place since round 4*/) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == LILI_KIND_SURF || kind == LILI_KIND_EDGE, "map_set: bad kind");
    return map_set_impl(ctx, kind, cloud, max_sq_radius, box6, true);
}
static float ord2f(unsigned u) { unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; }
static int map_set_impl(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius, const unsigned* box6, bool allow_guess) {
    ARGCHK(cloud, "map_set: null cloud");
    ARGCHK(max_sq_radius > 0 && std::isfinite(max_sq_radius), "map_set: max_sq_radius must be positive");
    ARGCHK(cloud->n < (1ll << 28), "map_set: at most 2^28 - 1 map points (32-bit byte offsets into the sorted array)");
    ARGCHK(cloud->n == 0 || cloud->data, "cloud: null data");
    ARGCHK(cloud->stride >= 12 && cloud->stride % 4 == 0, "cloud: stride must be a multiple of 4 and >= 12");
    ARGCHK(cloud->aux_offset < 0 || (size_t)cloud->aux_offset + 4 <= cloud->stride, "cloud: aux_offset outside the point");
    ARGCHK(cloud->mem == LILI_MEM_HOST || cloud->mem == LILI_MEM_DEVICE, "cloud: bad mem");
    HIPCHK(hipSetDevice(ctx->device));
    MapIndex& m = ctx->map[kind];
    m.valid = false;
    for (auto& s : ctx->slots) { s.k[kind].launches = 0; }
    m.n = (int64_t)cloud->n;
    m.has_aux = cloud->aux_offset >= 0;
    m.view = GridView{};
    m.n_cells = 0; m.cell = 0;
    if (m.n == 0) { m.valid = true; return LILI_OK; }
    const int n = (int)m.n;
    // The points are read where they lie — the caller's device array, or the staging copy of a host cloud — by the box pass, the count pass and the scatter
    // pass (round 4; rounds 1-3 first copied the map into a float4 array of the library's: 60-80 MB read + 80 MB written per 5 M points, 18 us).
    SrcCloud src{};
    src.p = reinterpret_cast<const unsigned char*>(cloud->data); src.stride = (int)cloud->stride; src.aux_off = cloud->aux_offset;
    if (cloud->mem == LILI_MEM_HOST) {
        HIPCHK(ctx->staging.ensure(cloud->n * cloud->stride));
        HIPCHK(hipMemcpyAsync(ctx->staging.p, cloud->data, cloud->n * cloud->stride, hipMemcpyHostToDevice, ctx->stream));
        src.p = ctx->staging.as<unsigned char>();
    }
    src.f4 = cloud->stride == sizeof(float4) && (reinterpret_cast<uintptr_t>(src.p) & 15) == 0 && (cloud->aux_offset == 12 || cloud->aux_offset < 0);
    // cell edge: >= 1.01 * gate radius so that the 27-cell neighbourhood covers the gate ball
    const int reach = ctx->grid_reach == 2 ? 2 : 1;
    // reach * cell >= 1.01 * gate radius; with reach 2 the cell edge is cell_pct % of the gate radius (50..100)
    double cell = std::sqrt(max_sq_radius) * 1.01 * (reach == 2 ? (double)ctx->cell_pct / 100.0 : 1.0);
    if (!(cell > 1e-6)) cell = 1e-6;
    // The box.  A build with no box from its caller GUESSES it when the previous build of this kind was of a cloud of about this size and the same gate (a pipeline
    // rebuilds the index of a slowly changing map per keyframe, L/src/BackendFusion.cpp:839-840): that build's true box, grown by 3/4 cell per side.  The count pass
    // checks the guess point by point (k_cell_count*: outside the grid / within a quarter cell of a face / near each face at all) and ORs the verdict into the density
    // banks; it comes back with the density at the END of the build — the one synchronisation a build has anyway.  A point outside (its cell would be a clamped one,
    // which the search's cell-distance bounds do not allow) or a face nothing comes near (some other cloud's box: correct, but a needlessly large grid) -> the box is
    // measured and the index built again before the call returns.  A build that guesses has no box pass and no host round trip before its kernels.
    unsigned* d_mm = ctx->misc.as<unsigned>();
    SpecBox& sb = ctx->spec_box[kind];
    // 8-bit cell counters (k_cell_count_narrow) unless a build of this kind has met a cell of more than 255 points (or the single-pass scan is off: A/B, fallback)
    const bool narrow = ctx->map_narrow_counts && !sb.wide_counts && ctx->scan_lookback;
    // a map of this kind, size and gate was dense at its last build: its gate-sized index goes without the super-row copy (nothing searches it while the fine index exists)
    const bool dense_hint = ctx->fine_grid && sb.dense && sb.max_sq_radius == max_sq_radius && (double)n <= 1.25 * (double)sb.n && (double)n >= 0.8 * (double)sb.n;
    BoxSource source = box6 ? kBoxGiven : kBoxMeasure;
    if (!box6 && allow_guess && ctx->map_guess_box && sb.valid && sb.max_sq_radius == max_sq_radius &&
        (double)n <= 1.25 * (double)sb.n && (double)n >= 0.8 * (double)sb.n) {
        if (sb.skip_guesses > 0) sb.skip_guesses--;      // this kind missed three guesses in a row: measure for the next few builds
        else source = kBoxGuess;
    }
    // every scratch word of the build — box banks, density banks, the scans' error word and status words, the guess's flags — starts from zero: one fill
    { const int rl = lili_lazy_sources_clear_of(ctx, ctx->misc.p, kMiscTotal); if (rl != LILI_OK) return rl; }
    HIPCHK(hipMemsetAsync(ctx->misc.p, 0, kMiscTotal, ctx->stream));
    unsigned banks[64 * 32], mm[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    const void* zeroed_p = nullptr;
    size_t zeroed_bytes = 0;
    double mn[3], mx[3];
    auto decode_box = [&](const unsigned w[6]) {
        bool any = true;
        for (int k = 0; k < 3; k++) { mn[k] = ord2f(w[k]); mx[k] = ord2f(w[3 + k]); if (!(mn[k] <= mx[k])) any = false; }
        if (!any) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }   // no finite point: one empty-ish cell
        return any;
    };
    if (source == kBoxGiven) { for (int k = 0; k < 6; k++) mm[k] = box6[k]; decode_box(mm); }
    else if (source == kBoxGuess) {
        decode_box(sb.mm);
        // 3/4 of a cell per side to begin with (x 2 after every failed guess): the table grows by 1.5 cells per axis, the cloud may drift by half a cell before the
        // count pass reports it within a quarter cell of a face
        const double margin = 0.75 * (double)sb.margin_cells * cell;
        for (int k = 0; k < 3; k++) { mn[k] -= margin; mx[k] += margin; }
    } else {
        hipLaunchKernelGGL(k_bbox_src, dim3(std::min(nblocks((int64_t)n, kBlock), 4096)), dim3(kBlock), 0, ctx->stream, src, n, d_mm);
        // the cell table of the previous build is cleared NOW, while the bounding box travels to the host and the GPU has nothing else to do (its size is only
        // known afterwards; a table that has to grow is cleared again in build_grid)
        DevBuf& table = narrow ? m.cell_tmp : m.cell_start;
        zeroed_p = table.p;
        zeroed_bytes = table.p ? table.cap : 0;
        if (zeroed_bytes) HIPCHK(hipMemsetAsync(table.p, 0, zeroed_bytes, ctx->stream));
        int rb = lili_readback_add(ctx, banks, d_mm, sizeof(banks)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb;
        unsigned inv_min[3] = {0u, 0u, 0u};          // the banks hold ~ordered(min) and ordered(max), both maximised from zero (bbox_to_banks)
        for (int b = 0; b < 64; b++) for (int k = 0; k < 3; k++) { inv_min[k] = std::max(inv_min[k], banks[b * 32 + k]); mm[3 + k] = std::max(mm[3 + k], banks[b * 32 + 3 + k]); }
        for (int k = 0; k < 3; k++) mm[k] = ~inv_min[k];
        decode_box(mm);
    }
    m.has_fine = false; m.fview = GridView{}; m.fbound = 0.f; m.fine_cell = 0; m.mean_occupancy = 0;
    constexpr size_t kRankBanks = 64;                       // k_cell_count: one bank per 128 bytes
    unsigned long long* d_rank = reinterpret_cast<unsigned long long*>(ctx->misc.as<char>() + 8192);
    unsigned scan_err = 0;
    bool err_read = false;
    int rc = build_grid(ctx, m, src, mn, mx, cell, reach, m.sorted, m.aux_sorted, m.cell_start, m.cell_start9, m.view, m.n_cells, m.cell, d_rank, source == kBoxGuess,
                        (float)(1.5 * (double)sb.margin_cells + 0.5) /* twice the margin: a cloud that moved towards one face moved away from the other */, narrow, true,
                        !dense_hint /* a dense map is searched through its fine index alone (lili_s2m_dense.hip): its gate-sized index only measures the density */, zeroed_p, zeroed_bytes);
    if (rc != LILI_OK) return rc;
    // The build's read-back: [density banks, with the check word of a guessed box in every bank | sticky error word of the look-back scans] lie side by side in ctx->misc.
    // Density adaptation (SURVEY §7 step 4, §8d Config 2 variant B): the point-weighted mean cell occupancy falls out of the count pass.
    // A map with many points per gate-sized cell gets a second, fine index whose cells hold ~3 points; k_associate_fine searches it first.
    {
        struct Back { unsigned long long rank[kRankBanks * 16]; unsigned err, pad; } back;
        static_assert(sizeof(Back) == 64 * 128 + 8, "layout of ctx->misc");
        {
            int rb = lili_readback_add(ctx, &back, ctx->misc.as<char>() + 8192, sizeof(back));
            // the caller's launches and read-backs that want to share this synchronisation (one shot; they must leave ctx->misc + 8192 onwards and the cloud alone)
            int hk = LILI_OK;
            ctx->hook_box_words_zero = source != kBoxMeasure;      // the box banks at the head of ctx->misc: written only by a build that measures its box
            if (rb == LILI_OK && ctx->pre_sync_hook) { auto hook = std::move(ctx->pre_sync_hook); ctx->pre_sync_hook = nullptr; hk = hook(); }
            if (rb == LILI_OK) rb = lili_readback_finish(ctx);      // (also behind a failed hook: `back` must not stay on the pending list)
            if (hk != LILI_OK) return hk;
            if (rb != LILI_OK) return rb;
        }
        scan_err = back.err;
        err_read = true;
        unsigned chk = 0;
        for (size_t b = 0; b < kRankBanks; b++) chk |= (unsigned)back.rank[b * 16 + 1];
        if (narrow && (chk & 256u)) {          // a cell of more than 255 points: this kind of map gets 32-bit counters from now on
            sb.wide_counts = true;
            ctx->narrow_overflows++;
            return map_set_impl(ctx, kind, cloud, max_sq_radius, source == kBoxGuess ? nullptr : mm, source == kBoxGuess);
        }
        if (source == kBoxGuess) {
            ctx->box_guesses++;
            if ((chk & 1u) || (chk & 0xFCu) != 0xFCu) {          // a point outside the guessed box, or a face of it that no point comes near (another cloud's box): measure, build again
                ctx->box_guess_misses++;
                // the margin grows after a miss but is capped where measuring is cheaper than a grid that much larger (8 cells per side; the
                // too-loose test scales with it), shrinks again after a run of hits (below), and a kind that keeps missing stops guessing for a while
                if (chk & 1u) sb.margin_cells = std::min(sb.margin_cells * 2, 8);
                sb.hits = 0;
                if (++sb.misses_in_a_row >= 3) { sb.skip_guesses = 8; sb.misses_in_a_row = 0; }
                return map_set_impl(ctx, kind, cloud, max_sq_radius, nullptr, false);
            }
            sb.misses_in_a_row = 0;
            if (++sb.hits >= 4 && sb.margin_cells > 1) { sb.margin_cells /= 2; sb.hits = 0; }      // four guesses in a row held: half the margin (towards 1)
            if (chk & 2u) sb.valid = false;      // the cloud comes within a quarter cell of a face of the grid: the next build measures its box again
        }
        if (ctx->fine_grid) {
            unsigned long long rank_sum = 0;
            for (size_t b = 0; b < kRankBanks; b++) rank_sum += back.rank[b * 16];
            m.mean_occupancy = 1.0 + 2.0 * (double)rank_sum / (double)n;
        }
        if (ctx->fine_grid && m.mean_occupancy > (double)ctx->fine_occupancy && m.cell == cell) {          // (a grid coarsened by max_cells is not refined)
            // surfaces: occupancy ~ cell^2; aim at ~3 points per fine cell, at least 4x and at most 64x finer cells per axis ... clamped
            double fc = cell * std::sqrt(3.0 / m.mean_occupancy);
            fc = std::min(std::max(fc, cell / 16.0), cell / 1.5);
            int64_t fcells = 0; double fcell_used = 0;
            // four EMPTY cells of margin on every side (round 6): the blocks of 3 x 3 rows that level 1 of the dense-map association walks lane by lane (k_associate_fine) then lie
            // inside the grid for every query inside the map's bounding box — surfaces ARE the faces of that box, and a block that straddles a face has no super-row of its own
            double fmn[3], fmx[3];
            for (int k = 0; k < 3; k++) { fmn[k] = mn[k] - 4.0 * fc; fmx[k] = mx[k] + 4.0 * fc; }
            rc = build_grid(ctx, m, src, fmn, fmx, fc, reach, m.sorted_f, m.aux_sorted_f, m.cell_start_f, m.cell_start9_f, m.fview, fcells, fcell_used, nullptr, false, 0.f, false /* the fine index keeps 32-bit counters: its check would need a read-back of its own */, false, true);
            if (rc != LILI_OK) return rc;
            const double rb = (double)reach * fcell_used / 1.01;
            float fb = (float)(rb * rb * (1.0 - 1e-6));
            if ((double)fb > rb * rb * (1.0 - 1e-6)) fb = std::nextafter(fb, 0.0f);                    // rounded DOWN: the bound only ever shrinks
            m.fbound = fb; m.fine_cell = fcell_used; m.has_fine = true;
            err_read = false;           // the fine index ran its own scans after the read-back
        }
    }
    // The single-pass scans publish tile prefixes between workgroups and rely on lower tiles making progress (HIP promises no dispatch
    // order); a look-back that gave up has left a wrong cell_start behind.  The sticky word travels with the density
    // read-back above — the synchronisation lili_map_set has anyway; only a build without it (fine_grid = 0) or with a second, fine
    // index pays a 4-byte read-back of its own.  If the word is set the whole index is rebuilt with the three-kernel scan, which has
    // no inter-workgroup dependency.
    if (ctx->scan_lookback) {
        if (!err_read) {
            { int rb = lili_readback_add(ctx, &scan_err, scan_err_word(ctx), sizeof(scan_err)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
        }
        if (scan_err) {
            ctx->scan_lookback = false;
            ctx->scan_fallbacks++;
            rc = map_set_impl(ctx, kind, cloud, max_sq_radius, source == kBoxGuess ? nullptr : mm, false);
            ctx->scan_lookback = true;
            return rc;
        }
    }
    // with scan_lookback = 0 the fine index's kernels (which read the CALLER's device cloud in place) were enqueued after the build's last read-back: the
    // header promises the cloud is free when the call returns, so the call waits for them
    else if (m.has_fine && cloud->mem == LILI_MEM_DEVICE) HIPCHK(hipStreamSynchronize(ctx->stream));
    // the hint was wrong (the map is not dense any more): its gate-sized index is the one that is searched and needs the super-row copy after all
    if (dense_hint && !m.has_fine) { sb.dense = false; return map_set_impl(ctx, kind, cloud, max_sq_radius, source == kBoxGuess ? nullptr : mm, source == kBoxGuess); }
    sb.dense = m.has_fine;
    // what the next build of this kind may start from: this cloud's TRUE box (a given box is the caller's, taken as true: lili_localmap_commit hands over the centroids'
    // own).  A build from a guess keeps the box it guessed from — margins do not pile up.
    if (source != kBoxGuess) {
        for (int k = 0; k < 6; k++) sb.mm[k] = mm[k];
        sb.valid = mm[0] <= mm[3] && mm[1] <= mm[4] && mm[2] <= mm[5];
        sb.max_sq_radius = max_sq_radius;
        sb.n = n;
        if (source == kBoxMeasure) { sb.hits = 0; }
    }
    m.valid = true;
    return LILI_OK;
}

extern "C" {

// Double-buffered map index: lili_map_set_begin builds the NEXT index of `kind` on a side stream (own staging and scratch words), so the
// kernels already enqueued on the context's stream — a keyframe's iterations — keep using the current index and run concurrently with
// the build (bandwidth-bound kernels under latency-bound ones); lili_map_set_end makes the new index current for everything enqueued
// after it.  The call itself still waits for the two small read-backs of the build (bounding box, density), not for the context's stream.
int lili_map_set_begin(lili_ctx* ctx, int kind, const lili_cloud* cloud, double max_sq_radius) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == LILI_KIND_SURF || kind == LILI_KIND_EDGE, "map_set_begin: bad kind");
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->build_stream) {          // lowest priority: the latency-bound iteration kernels on the context's stream go first
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = 0; }
        HIPCHK(hipStreamCreateWithPriority(&ctx->build_stream, hipStreamNonBlocking, lo));
    }
    if (!ctx->build_done[kind]) HIPCHK(hipEventCreateWithFlags(&ctx->build_done[kind], hipEventDisableTiming));
    if (!ctx->main_mark[kind]) HIPCHK(hipEventCreateWithFlags(&ctx->main_mark[kind], hipEventDisableTiming));
    if (ctx->misc_build.ensure(kMiscTotal) != hipSuccess) return ctx->fail(LILI_E_NOMEM, "map_set_begin: scratch allocation failed");
    // the buffers being rebuilt are the ones the index before the current one lived in: everything enqueued up to the swap that retired
    // them (main_mark, recorded by lili_map_set_end) has to be through before they are overwritten — NOT what was enqueued since, which
    // uses the current index and is what the build overlaps with
    if (ctx->main_marked[kind]) HIPCHK(hipStreamWaitEvent(ctx->build_stream, ctx->main_mark[kind], 0));
    // A cloud that already lives in device memory is normally produced by work enqueued on the context's stream (voxel filter, local-map
    // commit, an extractor): the build reads it from ANOTHER stream, so it has to wait for everything enqueued there so far — lili_map_set
    // got this ordering from the stream itself.  (Work on a third stream of the caller's is the caller's to order: synchronise it before _begin.)
    if (cloud && cloud->mem == LILI_MEM_DEVICE) {
        if (!ctx->cloud_ready) HIPCHK(hipEventCreateWithFlags(&ctx->cloud_ready, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ctx->cloud_ready, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->build_stream, ctx->cloud_ready, 0));
    }
    hipStream_t main_stream = ctx->stream;
    // run the ordinary build with the context's stream, scratch and index slot pointed at the build side (one host thread per context)
    ctx->stream = ctx->build_stream;
    ctx->staging.swap(ctx->staging_build); ctx->misc.swap(ctx->misc_build); ctx->map[kind].swap(ctx->map_next[kind]);
    const int rc = lili_map_set(ctx, kind, cloud, max_sq_radius);
    hipError_t e = rc == LILI_OK ? hipEventRecord(ctx->build_done[kind], ctx->build_stream) : hipSuccess;
    ctx->map[kind].swap(ctx->map_next[kind]); ctx->misc.swap(ctx->misc_build); ctx->staging.swap(ctx->staging_build);
    ctx->stream = main_stream;
    if (rc != LILI_OK) return rc;
    if (e != hipSuccess) return ctx->fail(LILI_E_HIP, std::string("map_set_begin: ") + hipGetErrorString(e));
    ctx->build_pending[kind] = true;
    return LILI_OK;
}

int lili_map_set_end(lili_ctx* ctx, int kind) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == LILI_KIND_SURF || kind == LILI_KIND_EDGE, "map_set_end: bad kind");
    if (!ctx->build_pending[kind]) return ctx->fail(LILI_E_STATE, "map_set_end: no lili_map_set_begin pending for this kind");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->build_done[kind], 0));      // later work on the context's stream sees the finished index
    ctx->map[kind].swap(ctx->map_next[kind]);
    HIPCHK(hipEventRecord(ctx->main_mark[kind], ctx->stream));             // everything that may still read the retired index is before this mark
    ctx->main_marked[kind] = true;
    for (auto& s : ctx->slots) { s.k[kind].launches = 0; }
    ctx->build_pending[kind] = false;
    return LILI_OK;
}

// Performance hint for maps much larger than a scan's footprint: the super-row copy (9x the points) is built only for the cells within
// `radius` of `center` at the following lili_map_set calls; queries elsewhere take the nine-row walk.  Results never depend on it.
int lili_map_focus(lili_ctx* ctx, const double center[3], double radius) {
    if (!ctx) return LILI_E_ARG;
    if (!(radius > 0) || !center) { ctx->focus_radius = 0; return LILI_OK; }
    ARGCHK(std::isfinite(center[0]) && std::isfinite(center[1]) && std::isfinite(center[2]) && std::isfinite(radius), "map_focus: non-finite argument");
    for (int k = 0; k < 3; k++) ctx->focus[k] = center[k];
    ctx->focus_radius = radius;
    return LILI_OK;
}

int lili_map_build_stats(lili_ctx* ctx, int32_t* box_guesses, int32_t* box_guess_misses, int32_t* scan_fallbacks) {
    if (!ctx) return LILI_E_ARG;
    if (box_guesses) *box_guesses = ctx->box_guesses;
    if (box_guess_misses) *box_guess_misses = ctx->box_guess_misses;
    if (scan_fallbacks) *scan_fallbacks = ctx->scan_fallbacks;
    return LILI_OK;
}

int lili_map_info(lili_ctx* ctx, int kind, int64_t* n_points, int64_t* n_cells, double* cell_edge) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == 0 || kind == 1, "map_info: bad kind");
    if (!ctx->map[kind].valid) return ctx->fail(LILI_E_STATE, "map_info: no map set");
    if (n_points) *n_points = ctx->map[kind].n;
    if (n_cells) *n_cells = ctx->map[kind].n_cells;
    if (cell_edge) *cell_edge = ctx->map[kind].cell;
    return LILI_OK;
}

int lili_map_density(lili_ctx* ctx, int kind, double* mean_occupancy, double* fine_cell_edge, double* fine_sq_radius) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == 0 || kind == 1, "map_density: bad kind");
    if (!ctx->map[kind].valid) return ctx->fail(LILI_E_STATE, "map_density: no map set");
    const MapIndex& m = ctx->map[kind];
    if (mean_occupancy) *mean_occupancy = m.mean_occupancy;
    if (fine_cell_edge) *fine_cell_edge = m.has_fine ? m.fine_cell : 0.0;
    if (fine_sq_radius) *fine_sq_radius = m.has_fine ? (double)m.fbound : 0.0;
    return LILI_OK;
}

}  // extern "C"
