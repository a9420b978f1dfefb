// Local-map assembly and pcl::VoxelGrid down-sampling on gfx950:
//   buildLocalMapWithLandMark + downSampleCloud + kd_tree->setInputCloud   — L/src/BackendFusion.cpp:1387-1528, 839-840
//   (front-end: buildLocalMap + downSampleCloud                            — L/src/LidarOdometry.cpp:280-323)
// Pieces:
//   * stable LSD radix sort of (u32 key, i32 value) pairs, 4-bit digits: per-tile digit histograms, one exclusive
//     scan over the digit-major table, stable scatter with ballot-based in-wave ranks (no atomics on the data path)
//   * VoxelGrid with PCL >= 1.8 semantics (App. B2): idx from f32 floor(x * inv_leaf) - min_b, output ordered by
//     voxel idx, centroid = f32 sums in INPUT ORDER / count (the stable sort keeps input order inside a voxel;
//     PCL's std::sort leaves that order unspecified — oracle mode stable_sort = 1)
//   * keyframe ring buffer: transformCloud (f64 rotate + translate, stored f32) of each pushed feature cloud,
//     oldest dropped beyond `width`, commit = concatenate + voxel filter + uniform-grid index (lili_map_set path)
#include "lili_ctx.h"
#include "lili_device_math.h"

namespace lili_detail { struct ConcatSeg { const float4* src; long long first; }; }      // one keyframe of the ring: its points and where they start in the concatenation

namespace lili {
__global__ void k_scan_block_sums(const int*, int64_t, int*);
__global__ void k_scan_sums(int*, int);
__global__ void k_scan_apply(const int*, int64_t, const int*, int*);
__global__ void k_bbox(const float4*, int, unsigned*);
__global__ void k_bbox_dev(const float4*, const int*, int, unsigned*);

constexpr int kSortBlock = 256, kSortItems = 8, kSortTile = kSortBlock * kSortItems;

// One atomic per DISTINCT table entry and wave: the lanes of a wave that count the same entry (`idx` < 2^16) are found with sixteen ballots and the lowest of them adds
// their number.  (Plain per-lane atomics on the digit tables serialised on a few hot counters — neighbouring points share voxels: +85 us per frame of the front-end pipeline.)
__device__ __forceinline__ void wave_hist_add(int* __restrict__ hist, unsigned idx, bool live) {
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const unsigned long long m = __ballot((idx >> b) & 1u);
        peers &= ((idx >> b) & 1u) ? m : ~m;
    }
    const int lane = threadIdx.x & 63;
    if (live && (peers & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&hist[idx], __popcll(peers));
}

__global__ __launch_bounds__(kSortBlock) void k_sort_hist(const unsigned* __restrict__ keys, int n, int shift, int nb, int* __restrict__ hist /*[16][nb]*/) {
    __shared__ int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    int base = blockIdx.x * kSortTile;
    int cnt[16];
#pragma unroll
    for (int d = 0; d < 16; d++) cnt[d] = 0;
    for (int r = 0; r < kSortItems; r++) {
        int i = base + r * kSortBlock + threadIdx.x;
        if (i < n) {
            int d = (keys[i] >> shift) & 15;
#pragma unroll
            for (int q = 0; q < 16; q++) cnt[q] += (q == d) ? 1 : 0;
        }
    }
#pragma unroll
    for (int d = 0; d < 16; d++) {
        int v = cnt[d];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&h[d], v);   // LDS integer atomics: 4 waves x 16 words
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kSortBlock) void k_sort_scatter(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift, int nb,
                                                             const int* __restrict__ offs /*[16][nb] exclusive*/, unsigned* __restrict__ keys_out, int* __restrict__ vals_out) {
    __shared__ int base[16];                  // running offset of each digit inside this tile
    __shared__ int wcnt[kSortBlock / 64][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 16) base[threadIdx.x] = offs[threadIdx.x * nb + blockIdx.x];
    __syncthreads();
    const int tile = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; r++) {
        int i = tile + r * kSortBlock + threadIdx.x;
        bool live = i < n;
        unsigned key = live ? keys_in[i] : 0u;
        int val = live ? vals_in[i] : 0;
        int d = live ? (int)((key >> shift) & 15) : 16;
        // lanes of this wave with the same digit (4 ballots) -> stable in-wave rank
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            unsigned long long m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        int rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (threadIdx.x < (kSortBlock / 64) * 16) (&wcnt[0][0])[threadIdx.x] = 0;
        __syncthreads();
        if (live && rank == 0) wcnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (live) {
            int off = base[d];
            for (int w = 0; w < wave; w++) off += wcnt[w][d];
            keys_out[off + rank] = key;
            vals_out[off + rank] = val;
        }
        __syncthreads();
        if (threadIdx.x < 16) { int s = 0; for (int w = 0; w < kSortBlock / 64; w++) s += wcnt[w][threadIdx.x]; base[threadIdx.x] += s; }
        __syncthreads();
    }
}

// The same sort with 8-bit digits (round 3): a quarter of the passes for the price of 256-entry digit tables — the passes are launch- and
// latency-bound at the sizes of a keyframe ring (1 M keys: 8 x (hist + scan + scatter) 4-bit passes were ~25 launches for a 26-bit key).
// Stability as above: keys of a tile keep their order inside a digit (ballot ranks inside a wave, wave counts in LDS, rounds in order).
constexpr int kSortItems8 = 16;      // 4096 keys per tile: the 256 x tiles digit table of 1 M keys (62 k words) still takes the one-launch scan
__global__ __launch_bounds__(kSortBlock) void k_sort_hist8(const unsigned* __restrict__ keys, int n, int shift, int nb, int items, int* __restrict__ hist /*[256][nb]*/) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortBlock * items;
    for (int r = 0; r < items; r++) {
        const int i = base + r * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);      // LDS integer atomics
    }
    __syncthreads();
    hist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}
// `raw_hist`: `offs` is the COUNT table of k_sort_hist8 and every workgroup derives its offsets itself (digits before its digit: all tiles; its own digit:
// the tiles before it) — up to 256 tiles (256 KB of L2 reads per workgroup) that is cheaper than the scan launch between the two kernels (a pass = two
// launches instead of three; the launches, not the work, are what a 20 k-key sort costs; 200 k keys: 212 -> 190 us per filter; beyond ~1 M keys the
// table reads cost more than the scan, option sort_fused_max_tiles).
__global__ __launch_bounds__(kSortBlock) void k_sort_scatter8(const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift, int nb, int items,
                                                              const int* __restrict__ offs /*[256][nb] exclusive, or counts*/, int raw_hist, unsigned* __restrict__ keys_out,
                                                              int* __restrict__ vals_out, int* __restrict__ hist_next /*[256][nb] counts of the NEXT pass's digit per OUTPUT tile (zeroed by the
                                                              caller), or nullptr*/, int next_shift) {
    __shared__ int base[256];                 // running offset of each digit inside this tile
    __shared__ int wcnt[kSortBlock / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (raw_hist) {
        __shared__ int wtot[kSortBlock / 64];
        int row = 0, pre = 0;
        for (int b = 0; b < nb; b++) { const int c = offs[threadIdx.x * nb + b]; row += c; if (b < (int)blockIdx.x) pre += c; }
        int inc = row;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wtot[w];
        base[threadIdx.x] = before + inc - row + pre;
    } else
    base[threadIdx.x] = offs[threadIdx.x * nb + blockIdx.x];
    const int tile = blockIdx.x * kSortBlock * items;
    for (int r = 0; r < items; r++) {
        const int i = tile + r * kSortBlock + threadIdx.x;
        const bool live = i < n;
        const unsigned key = live ? keys_in[i] : 0u;
        const int val = live ? vals_in[i] : 0;
        const int d = live ? (int)((key >> shift) & 255u) : 256;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
#pragma unroll
        for (int w = 0; w < kSortBlock / 64; w++) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
        if (live && rank == 0) wcnt[wave][d] = __popcll(peers);
        __syncthreads();
        int opos = 0;
        if (live) {
            int off = base[d];
            for (int w = 0; w < wave; w++) off += wcnt[w][d];
            keys_out[off + rank] = key;
            vals_out[off + rank] = val;
            opos = off + rank;
        }
        if (hist_next) wave_hist_add(hist_next, ((key >> next_shift) & 255u) * (unsigned)nb + (unsigned)(opos / (kSortBlock * items)), live);      // (round 6: saves the next pass's k_sort_hist8 launch)
        __syncthreads();
        { int s = 0; for (int w = 0; w < kSortBlock / 64; w++) s += wcnt[w][threadIdx.x]; base[threadIdx.x] += s; }
        __syncthreads();
    }
}

// arms the six ordered-uint words of a bounding box (min xyz = all ones, max xyz = 0) — a kernel, not a host-to-device copy of 24 bytes from pageable memory, which the
// runtime stages and serialises (~6 us of API time each)
__global__ void k_box_init(unsigned* __restrict__ mm) { if (threadIdx.x < 6) mm[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u; }

struct VoxDev { float inv_leaf; int min_b[3]; int mul[3]; unsigned sentinel; };   // sentinel = number of voxels of the bounding box: key of non-finite points

__global__ void k_vox_key(const float4* __restrict__ pts, int n, VoxDev V, unsigned* __restrict__ keys, int* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    vals[i] = i;
    // pcl::VoxelGrid skips non-finite points of a non-dense cloud (voxel_grid.hpp: `if (!isFinite(point)) continue`): they get a
    // key above every voxel index, sort to the end and are left out of the output
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = V.sentinel; return; }
    // pcl::VoxelGrid::applyFilter: static_cast<int>(floor(x * inverse_leaf_size) - static_cast<float>(min_b))
    int i0 = (int)(floorf(p.x * V.inv_leaf) - (float)V.min_b[0]);
    int i1 = (int)(floorf(p.y * V.inv_leaf) - (float)V.min_b[1]);
    int i2 = (int)(floorf(p.z * V.inv_leaf) - (float)V.min_b[2]);
    keys[i] = (unsigned)(i0 * V.mul[0] + i1 * V.mul[1] + i2 * V.mul[2]);
}
// k_bbox into words that start from ZERO (words 0-2: ~ordered(min), 3-5: ordered(max), all maximised): a caller whose scratch has just been cleared by a fill it needs
// anyway saves the launch that arms the box (k_box_init)
// `zero` / `n_zero` (round 6): words this launch clears on the way — the digit tables of the radix sort behind it (the histograms ride on the key and scatter kernels)
__global__ __launch_bounds__(256) void k_bbox_z(const float4* __restrict__ pts, int n, unsigned* __restrict__ mmz, int* __restrict__ zero, int n_zero) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) zero[i] = 0;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    __shared__ float smn[4][3], smx[4][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { smn[threadIdx.x >> 6][k] = mn[k]; smx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = smn[0][k], b = smx[0][k];
        for (int w = 1; w < 4; w++) { a = fminf(a, smn[w][k]); b = fmaxf(b, smx[w][k]); }
        auto ord = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        if (a <= b) { atomicMax(&mmz[k], ~ord(a)); atomicMax(&mmz[3 + k], ord(b)); }
    }
}
// k_vox_key with the bounding box read where k_bbox left it (ordered-uint words `mm`) — no host round trip before the sort (round 5).  The host does not know the
// box, so it cannot know how many key bits the sort has to cover: it GUESSES them (`bits_guess`: what the previous filter of this leaf size needed, rounded up to whole
// radix passes) and every thread checks the guess against the box it finds; res[0] = 0 (the keys fit), 1 (they do not: the caller repeats the filter the measured way),
// 2 (no finite point), 3 (PCL's int32 voxel-index overflow) and res[1] = the bits the keys need come back with the filter's voxel count.  Same arithmetic as the host
// code of voxel_sort + k_vox_key; non-finite points get the key 2^bits_guess - 1, above every voxel index.
__global__ void k_vox_key_dev(const float4* __restrict__ pts, int n, float inv_leaf, const unsigned* __restrict__ mm, int zform /*1: the minima are stored inverted (k_bbox_z)*/,
                              int bits_guess, unsigned* __restrict__ keys, int* __restrict__ vals, int* __restrict__ res,
                              int* __restrict__ hist0 /*[256][nb] counts of the lowest digit per sort tile, or nullptr*/, int nb, int tile) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto dec = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
    int min_b[3], div_b[3];
    bool any = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = dec(zform ? ~mm[k] : mm[k]), mx = dec(mm[3 + k]);
        if (!(mn <= mx)) any = false;
        min_b[k] = (int)floorf(mn * inv_leaf);
        div_b[k] = (int)floorf(mx * inv_leaf) - min_b[k] + 1;
    }
    const double total = any ? (double)div_b[0] * (double)div_b[1] * (double)div_b[2] : 0.0;
    const unsigned sentinel = bits_guess >= 32 ? 0xFFFFFFFFu : (1u << bits_guess) - 1u;
    int status = 0;
    if (!any) status = 2;
    else if (total > 2147483647.0) status = 3;
    else if (total > (double)sentinel) status = 1;      // voxel indices 0 .. total - 1 and the sentinel above them
    if (i == 0) {
        int bits = 1;
        if (status == 0 || status == 1) while (bits < 32 && (double)(1ull << bits) < total + 1.0) bits++;
        res[0] = status; res[1] = bits;
    }
    const bool live = i < n;
    unsigned key = sentinel;
    if (live) {
        vals[i] = i;
        const float4 p = pts[i];
        if (status == 0 && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const int i0 = (int)(floorf(p.x * inv_leaf) - (float)min_b[0]);
            const int i1 = (int)(floorf(p.y * inv_leaf) - (float)min_b[1]);
            const int i2 = (int)(floorf(p.z * inv_leaf) - (float)min_b[2]);
            key = (unsigned)(i0 * 1 + i1 * div_b[0] + i2 * (div_b[0] * div_b[1]));
        }
        keys[i] = key;
    }
    // the first pass's digit counts (k_sort_hist8 would count the same); table entry = digit * nb + tile, nb <= 256 tiles
    if (hist0) wave_hist_add(hist0, (key & 255u) * (unsigned)nb + (unsigned)(live ? i / tile : 0), live);
}
// Key of one keyframe's points for the sorted ring WITHOUT a host round trip: voxel coordinates relative to the keyframe's own bounding box
// (ordered-uint words `mm` left in device memory by k_bbox), packed (i2 : 10 bits, i1 : 11, i0 : 11) — the same lexicographic order as the
// box-relative voxel index.  A keyframe wider than 2047 x 2047 x 1023 voxels raises *bad (the commit then takes the full rebuild).
__global__ void k_vox_key_packed(const float4* __restrict__ pts, int n, float inv_leaf, const unsigned* __restrict__ mm, unsigned* __restrict__ keys, int* __restrict__ vals,
                                 unsigned* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto dec = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); };
    const float4 p = pts[i];
    vals[i] = i;
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { keys[i] = 0xFFFFFFFFu; return; }
    const int m0 = (int)floorf(dec(mm[0]) * inv_leaf), m1 = (int)floorf(dec(mm[1]) * inv_leaf), m2 = (int)floorf(dec(mm[2]) * inv_leaf);
    const int i0 = (int)floorf(p.x * inv_leaf) - m0, i1 = (int)floorf(p.y * inv_leaf) - m1, i2 = (int)floorf(p.z * inv_leaf) - m2;
    if (i0 < 0 || i0 > 2047 || i1 < 0 || i1 > 2047 || i2 < 0 || i2 > 1022) { *bad = 1u; keys[i] = 0xFFFFFFFEu; return; }
    keys[i] = ((unsigned)i2 << 22) | ((unsigned)i1 << 11) | (unsigned)i0;
}
// head flags of the sorted keys; the points are copied into sorted order on the way (round 5: k_vox_centroid then reads a voxel's members as ONE contiguous run
// instead of chasing an index per member)
__global__ void k_vox_heads(const unsigned* __restrict__ keys, const int* __restrict__ vals, const float4* __restrict__ pts, int n, unsigned sentinel, int* __restrict__ flags,
                            float4* __restrict__ spts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    spts[i] = pts[vals[i]];
    flags[i] = (keys[i] != sentinel && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}
__device__ __forceinline__ void wave_lds_order();
// CentroidPoint over the sorted cloud: float accumulators, members added in sorted (= input) order — the sum itself has to stay sequential, so the launch lasts as long as
// its fullest voxel (next to the sensor: up to ~600 of a Livox frame's 21 k surf features; nine voxels of ten hold 4 or fewer).  Round 5: the launch is a chain of THREE
// memory round trips whatever the voxel's size —
//   1. every position asks for its two slot words (a head is where the scan of the head flags steps) and, speculatively, the kFirst points from itself on;
//   2. a head asks for the next head's position (k_scan_flags left them compacted in head_pos): the voxel's length is known, no key is compared any more;
//   3. a voxel of up to kFirst + kMem members is finished by its head's thread with all its loads in flight at once; a longer one joins the workgroup's list, which the four
//      waves share out (the fullest voxels of a frame are neighbours in the sorted order): the wave requests ALL the voxel's remaining members at once (16 per lane),
//      parks them in LDS and every lane adds them up in the same order from a wave-uniform address, the next eight requested before the sums —
// where rounds 1-4 walked a voxel in trips of 8 / 32 members with one or two dependent round trips each (31 -> 20 -> 17 us per frame for the trips alone).
__global__ __launch_bounds__(256) void k_vox_centroid(const unsigned* __restrict__ keys, const int* __restrict__ slot /*exclusive scan of the head flags, [n+1]*/,
                               const int* __restrict__ head_pos /*position of voxel o's head; [number of voxels] = where the voxels end if has_end*/, int has_end,
                               const float4* __restrict__ spts /*points in sorted order*/, int n, unsigned sentinel, float4* __restrict__ out, int* __restrict__ out_cnt) {
    constexpr int kFirst = 4, kMem = 32, kStage = 1024;
    __shared__ float4 stage[4][kStage];
    __shared__ int q_n, q_o[8], q_m[8], q_end[8];
    __shared__ float4 q_s[8];
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ic = min(i, n - 1);
    const int s0 = slot[ic], s1 = slot[ic + 1], n_out = slot[n];
    float4 f[kFirst];
#pragma unroll
    for (int u = 0; u < kFirst; u++) f[u] = spts[min(i + u, n - 1)];
    const bool head = i < n && s1 != s0;
    int end = i + 1;
    if (head) {
        if (has_end || s0 + 1 < n_out) end = head_pos[s0 + 1];
        else { const unsigned k = keys[i]; while (end < n && keys[end] == k) end++; }      // (heads compacted by k_vox_head_pos: the last voxel ends where the keys stop being voxels; a chain of dependent loads)
    }
    const int len = end - i;
    float sx = 0.f, sy = 0.f, sz = 0.f, sa = 0.f;
#pragma unroll
    for (int u = 0; u < kFirst; u++) if (u < len) { sx += f[u].x; sy += f[u].y; sz += f[u].z; sa += f[u].w; }
    if (head && len > kFirst && len <= kFirst + kMem) {
        float4 pp[kMem];
#pragma unroll
        for (int u = 0; u < kMem; u++) pp[u] = spts[min(i + kFirst + u, end - 1)];
#pragma unroll
        for (int u = 0; u < kMem; u++) if (kFirst + u < len) { sx += pp[u].x; sy += pp[u].y; sz += pp[u].z; sa += pp[u].w; }
    }
    if (head && len <= kFirst + kMem) {
        const float fn = (float)len;
        out[s0] = make_float4(sx / fn, sy / fn, sz / fn, sa / fn);
        if (out_cnt) out_cnt[s0] = len;
    }
    if (head && len > kFirst + kMem) {
        const int j = atomicAdd(&q_n, 1);      // (at most 256 / (kFirst + kMem + 1) = 6 entries)
        q_o[j] = s0; q_m[j] = i + kFirst; q_end[j] = end; q_s[j] = make_float4(sx, sy, sz, sa);
    }
    __syncthreads();
    const int nq = q_n;
    for (int j = wave; j < nq; j += 4) {      // (wave-uniform)
        int mL = q_m[j];
        const int eL = q_end[j], oL = q_o[j];
        // lane c (of every four) carries component c of the sum: ONE dependent addition per member instead of four (the fullest voxel's ~600 members are the launch's tail)
        const int comp = lane & 3;
        float acc = comp == 0 ? q_s[j].x : comp == 1 ? q_s[j].y : comp == 2 ? q_s[j].z : q_s[j].w;
        const float* st = reinterpret_cast<const float*>(stage[wave]) + comp;
        while (mL < eL) {
            const int cnt = min(eL - mL, kStage);
            float4 reg[kStage / 64];
#pragma unroll
            for (int r = 0; r < kStage / 64; r++) reg[r] = spts[min(mL + 64 * r + lane, eL - 1)];      // (no branch: every request leaves before the first answer is awaited; rows behind the end re-read its last point)
#pragma unroll
            for (int r = 0; r < kStage / 64; r++) stage[wave][64 * r + lane] = reg[r];
            wave_lds_order();
            float p[16];
#pragma unroll
            for (int t = 0; t < 16; t++) p[t] = st[4 * t];
            for (int u = 0; u < cnt; u += 16) {
                float nx[16];
#pragma unroll
                for (int t = 0; t < 16; t++) nx[t] = st[4 * min(u + 16 + t, kStage - 1)];
#pragma unroll
                for (int t = 0; t < 16; t++) if (u + t < cnt) acc += p[t];
#pragma unroll
                for (int t = 0; t < 16; t++) p[t] = nx[t];
            }
            mL += cnt;
            wave_lds_order();      // the next members overwrite the stage
        }
        const float ax = __shfl(acc, 0), ay = __shfl(acc, 1), az = __shfl(acc, 2), aw = __shfl(acc, 3);
        if (lane == 0) {
            const int len_l = eL - (q_m[j] - kFirst);
            const float fn = (float)len_l;
            out[oL] = make_float4(ax / fn, ay / fn, az / fn, aw / fn);
            if (out_cnt) out_cnt[oL] = len_l;
        }
    }
}

// pcl::VoxelGrid of a SMALL cloud (<= kVoxSmallMax points: a Livox frame's surf features, the queries of the front end — L/src/LidarOdometry.cpp:320-322) in ONE
// launch of one workgroup (round 5): bounding box, PCL's voxel index, stable sort, voxel heads, centroids.  The general path is a chain of ~14 launches with two host
// round trips (box, count) — 115 us for 3 k points, of which the GPU works ~30.  Same arithmetic as k_vox_key / k_vox_centroid: the index
// (i0 - min0) + (i1 - min1) * dx + (i2 - min2) * dx * dy with i = floor(p * inverse_leaf), points sorted by (index, input position) — a bitonic network on 64-bit keys
// in LDS —, members of a voxel summed in input order with float accumulators, non-finite points skipped.  res[0] = number of voxels, res[1] = 1 if the voxel index
// would not fit 31 bits (the caller then takes the general path, which reports PCL's overflow error), 2 if no point is finite.
constexpr int kVoxSmallMax = 8192, kVoxSmallThreads = 1024;
// between two steps of a sort in which a wave reads what only ITS OWN lanes wrote: the LDS serves a wave's accesses in order; the compiler must keep them in order too
__device__ __forceinline__ void wave_lds_order() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__global__ __launch_bounds__(kVoxSmallThreads) void k_voxel_small(const float4* __restrict__ pts, int n, float inv_leaf, float4* __restrict__ out, int* __restrict__ out_cnt,
                                                                   int* __restrict__ res) {
    __shared__ unsigned long long key[kVoxSmallMax];
    __shared__ int red[6][kVoxSmallThreads / 64];
    __shared__ int box[6];
    __shared__ int wsum[kVoxSmallThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int np2 = 64; while (np2 < n) np2 <<= 1;
    // ---- bounding box of the finite points in voxel coordinates
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    for (int i = tid; i < n; i += kVoxSmallThreads) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const int c[3] = {(int)floorf(p.x * inv_leaf), (int)floorf(p.y * inv_leaf), (int)floorf(p.z * inv_leaf)};
#pragma unroll
            for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], c[k]); mx[k] = max(mx[k], c[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int o = 32; o > 0; o >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], o)); mx[k] = max(mx[k], __shfl_xor(mx[k], o)); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (tid < 6) {
        int v = red[tid][0];
        for (int w = 1; w < kVoxSmallThreads / 64; w++) v = tid < 3 ? min(v, red[tid][w]) : max(v, red[tid][w]);
        box[tid] = v;
    }
    __syncthreads();
    const bool any = box[0] <= box[3];
    const long long dx = any ? (long long)box[3] - box[0] + 1 : 1, dy = any ? (long long)box[4] - box[1] + 1 : 1, dz = any ? (long long)box[5] - box[2] + 1 : 1;
    const bool overflow = (double)dx * (double)dy * (double)dz > 2147483647.0;
    if (overflow || !any) {
        if (tid == 0) { res[0] = 0; res[1] = overflow ? 1 : 2; }      // 2: no finite point at all (an error, as in the general path)
        return;
    }
    // ---- keys: (voxel index << 13) | input position; non-finite points and the padding sort last
    for (int i = tid; i < np2; i += kVoxSmallThreads) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const float4 p = pts[i];
            if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                const long long i0 = (long long)(int)floorf(p.x * inv_leaf) - box[0], i1 = (long long)(int)floorf(p.y * inv_leaf) - box[1], i2 = (long long)(int)floorf(p.z * inv_leaf) - box[2];
                k = ((unsigned long long)(i0 + i1 * dx + i2 * dx * dy) << 13) | (unsigned long long)i;
            }
        }
        key[i] = k;
    }
    __syncthreads();
    // ---- bitonic sort (ascending).  Partner distances <= 64: the 64 pairs a wave handles per trip lie in ONE aligned run of 128 keys that no other wave touches
    //      until the next distance > 64 — those steps (51 of the 66 of 2 k keys) need no block barrier, only the wave's own LDS order
    auto step = [&](int t, int j, int k) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // the lower partner of the t-th pair at distance j
        const int l = i | j;
        const unsigned long long a = key[i], b = key[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { key[i] = b; key[l] = a; }
    };
    for (int k = 2; k <= np2; k <<= 1) {
        int j = k >> 1;
        for (; j > 64; j >>= 1) {
            for (int t = tid; t < np2 / 2; t += kVoxSmallThreads) step(t, j, k);
            __syncthreads();
        }
        for (int t = tid; t < np2 / 2; t += kVoxSmallThreads)
            for (int jj = j; jj > 0; jj >>= 1) { step(t, jj, k); wave_lds_order(); }
        __syncthreads();
    }
    // ---- voxel heads and their output slots (exclusive scan of the head flags over the sorted positions)
    const int per = (np2 + kVoxSmallThreads - 1) / kVoxSmallThreads;      // consecutive positions per thread (<= 8)
    const int base = tid * per;
    int heads = 0;
    for (int u = 0; u < per; u++) {
        const int i = base + u;
        if (i < np2 && key[i] != ~0ull && (i == 0 || (key[i] >> 13) != (key[i - 1] >> 13))) heads++;
    }
    int inc = heads;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int wofs = 0, total = 0;
    for (int w = 0; w < kVoxSmallThreads / 64; w++) { const int t = wsum[w]; if (w < wave) wofs += t; total += t; }
    int slot = wofs + inc - heads;
    // ---- centroids: the thread that owns a head walks its voxel's members in sorted (= input) order
    for (int u = 0; u < per; u++) {
        const int i = base + u;
        if (!(i < np2 && key[i] != ~0ull && (i == 0 || (key[i] >> 13) != (key[i - 1] >> 13)))) continue;
        const unsigned long long v = key[i] >> 13;
        float sx = 0.f, sy = 0.f, sz = 0.f, sa = 0.f; int c = 0;
        for (int m = i; m < np2 && (key[m] >> 13) == v && key[m] != ~0ull; m++) {
            const float4 p = pts[(int)(key[m] & 8191ull)];
            sx += p.x; sy += p.y; sz += p.z; sa += p.w; c++;
        }
        const float fn = (float)c;
        out[slot] = make_float4(sx / fn, sy / fn, sz / fn, sa / fn);
        if (out_cnt) out_cnt[slot] = c;
        slot++;
    }
    if (tid == 0) { res[0] = total; res[1] = 0; }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Incremental local map (round 3; the reference pops one keyframe and pushes one per step, L/src/BackendFusion.cpp:1407-1477, where
// lili_localmap_commit used to re-concatenate, re-sort and re-reduce the whole ring).  The ring's points are kept SORTED by an absolute
// voxel key — (k, j, i) = floor(p * inverse_leaf) packed lexicographically, which is the order pcl::VoxelGrid's box-relative index
// induces whatever the bounding box is — with older keyframes first inside a voxel, i.e. exactly the order the stable sort of the
// concatenated cloud produces.  A step then is: sort the NEW keyframe alone (20 k points), drop the popped keyframe's entries and merge the
// new ones in one streaming pass (ranks by prefix sum and binary search, no comparison network), and run the centroid pass — the same f32
// sums over the same members in the same order, so the map is bit-identical to the full rebuild (tests/test_voxel_gpu.py).
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long abs_voxel_key(float4 p, float inv_leaf) {
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return ~0ull;               // sorts last, never a voxel (voxel_grid.hpp: !isFinite -> skipped)
    const long long i = (long long)floorf(p.x * inv_leaf) + (1ll << 20), j = (long long)floorf(p.y * inv_leaf) + (1ll << 20), k = (long long)floorf(p.z * inv_leaf) + (1ll << 20);
    if ((i | j | k) < 0 || i >= (1ll << 21) || j >= (1ll << 21) || k >= (1ll << 21)) return ~0ull - 1ull;      // beyond +-2^20 voxels: the host falls back to the full rebuild (flagged)
    return ((unsigned long long)k << 42) | ((unsigned long long)j << 21) | (unsigned long long)i;
}
// sorted copy of one cloud: out_pt[r] = pts[order[r]], its absolute key and a constant sequence number; *bad is raised if a point lies outside the key range
__global__ void k_sorted_gather(const float4* __restrict__ pts, const int* __restrict__ order, int n, float inv_leaf, unsigned seq_const, const lili_detail::ConcatSeg* __restrict__ segs,
                                const unsigned* __restrict__ seg_seq, int n_seg, float4* __restrict__ out_pt, unsigned long long* __restrict__ out_key, unsigned* __restrict__ out_seq,
                                unsigned* __restrict__ bad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int src = order[r];
    const float4 p = pts[src];
    const unsigned long long key = abs_voxel_key(p, inv_leaf);
    if (key == ~0ull - 1ull) *bad = 1u;
    unsigned sq = seq_const;
    if (segs) {          // the cloud is the concatenation of the ring: the keyframe of a point by bisection over the segments' first positions
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].first <= (long long)src) lo = mid; else hi = mid - 1; }
        sq = seg_seq[lo];
    }
    out_pt[r] = p; out_key[r] = key; out_seq[r] = sq;
}
// The new keyframe of an incremental commit (<= kVoxSmallMax points: a frame's down-sampled features) sorted by absolute voxel key (round 5): what k_bbox +
// k_vox_key_packed + four radix passes + k_sorted_gather produce in twelve launches — the points in (key, input position) order with their keys and the keyframe's
// sequence number; *bad is raised for a point beyond the key range (the commit then rebuilds).  Sorting by COUNTING, spread over the chip: a wave holds 64 points (one
// per lane) and 64 keys of the keyframe (one per lane, broadcast lane by lane through SGPRs) and counts, per point, the (key, position) pairs below it among those 64;
// workgroup (bx, by) covers points 64 bx .. and the eight 64-key slices 8 by ..; the counts of a point add up in `rank` (zeroed by the caller), which k_rank_scatter turns
// into places.  n^2 / 4096 wave tasks of ~1.5 us each on n^2 / 32768 workgroups: ~4 us for the 2 k points of a Livox frame, where a single-workgroup bitonic network
// took 22-27 us and counting inside n / 64 workgroups 19.
constexpr int kRankWaves = 8;
constexpr size_t kRankOff = 2 * 64 * 128 + 256;      // ctx->misc behind the box / density banks (lili_map.hip: kMiscBytes): the scan-status words of a map build
static_assert(kRankOff + (size_t)kVoxSmallMax * 4 <= kMiscAlloc, "rank words of k_rank_count");
__global__ __launch_bounds__(64 * kRankWaves) void k_rank_count(const float4* __restrict__ pts, int n, float inv_leaf, int* __restrict__ rank, unsigned* __restrict__ bad) {
    __shared__ int part[kRankWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    unsigned long long ke = ~0ull;
    if (e < n) { ke = abs_voxel_key(pts[e], inv_leaf); if (ke == ~0ull - 1ull && wave == 0 && blockIdx.y == 0) *bad = 1u; }
    const int s0 = (blockIdx.y * kRankWaves + wave) * 64;      // this wave's slice of the keyframe
    int cnt = 0;
    if (s0 < n) {
        const unsigned long long ks = s0 + lane < n ? abs_voxel_key(pts[s0 + lane], inv_leaf) : ~0ull;
        const int len = min(64, n - s0);
        const unsigned ks_lo = (unsigned)ks, ks_hi = (unsigned)(ks >> 32);
#pragma unroll
        for (int u = 0; u < 64; u++) {
            const unsigned long long ku = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ks_hi, u) << 32) | (unsigned)__builtin_amdgcn_readlane((int)ks_lo, u);
            const int idx = s0 + u;
            if (u < len) cnt += (ku < ke || (ku == ke && idx < e)) ? 1 : 0;      // (key, input position): stable
        }
    }
    part[wave][lane] = cnt;
    __syncthreads();
    if (wave != 0 || e >= n) return;
    int r = 0;
#pragma unroll
    for (int w = 0; w < kRankWaves; w++) r += part[w][lane];
    if (r) atomicAdd(&rank[e], r);
}
__global__ void k_rank_scatter(const float4* __restrict__ pts, int n, float inv_leaf, unsigned seq, const int* __restrict__ rank, float4* __restrict__ out_pt,
                               unsigned long long* __restrict__ out_key, unsigned* __restrict__ out_seq) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float4 p = pts[e];
    const int r = rank[e];
    out_pt[r] = p; out_key[r] = abs_voxel_key(p, inv_leaf); out_seq[r] = seq;
}

struct DropSeqs { unsigned s[4]; int n; };
__global__ void k_keep_flags(const unsigned* __restrict__ seq, long long n, DropSeqs d, int* __restrict__ flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned q = seq[i];
    bool drop = false;
#pragma unroll
    for (int k = 0; k < 4; k++) drop = drop || (k < d.n && q == d.s[k]);
    flags[i] = drop ? 0 : 1;
}
// kept entry i of the old list goes to (kept entries before it) + (new entries with a SMALLER key: new ones follow old ones of the same voxel)
__device__ __forceinline__ void merge_old_item(const unsigned long long* __restrict__ key, const float4* __restrict__ pt, const unsigned* __restrict__ seq, bool keep,
                            const int* __restrict__ rank /*exclusive scan of the keep flags, [n+1]*/, long long i, const unsigned long long* __restrict__ nkey, int n_new,
                            unsigned long long* __restrict__ okey, float4* __restrict__ opt, unsigned* __restrict__ oseq) {
    if (!keep) return;
    const unsigned long long k = key[i];
    int lo = 0, hi = n_new;                       // lower_bound(nkey, k)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (nkey[mid] < k) lo = mid + 1; else hi = mid; }
    const long long pos = (long long)rank[i] + lo;
    okey[pos] = k; opt[pos] = pt[i]; oseq[pos] = seq[i];
}
// new entry j goes to j + (kept old entries with key <= its key)
__device__ __forceinline__ void merge_new_item(const unsigned long long* __restrict__ nkey, const float4* __restrict__ npt, int j, unsigned nseq, const unsigned long long* __restrict__ key,
                            const int* __restrict__ rank, long long n, unsigned long long* __restrict__ okey, float4* __restrict__ opt, unsigned* __restrict__ oseq) {
    const unsigned long long k = nkey[j];
    long long lo = 0, hi = n;                     // upper_bound(key, k)
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (key[mid] <= k) lo = mid + 1; else hi = mid; }
    const long long pos = (long long)j + (long long)rank[lo];
    okey[pos] = k; opt[pos] = npt[j]; oseq[pos] = nseq;
}
// both halves of the merge in ONE launch: threads [0, n) place the old list's entries, threads [n, n + n_new) the new keyframe's.  `flags` (the keep flags in memory) or,
// when null, the keyframes dropped (`drop`: the scan derived the flags itself, k_scan_flags<kScanKeep>)
__global__ void k_merge(const unsigned long long* __restrict__ key, const float4* __restrict__ pt, const unsigned* __restrict__ seq, const int* __restrict__ flags, DropSeqs drop,
                        const int* __restrict__ rank, long long n, const unsigned long long* __restrict__ nkey, const float4* __restrict__ npt, int n_new, unsigned nseq,
                        unsigned long long* __restrict__ okey, float4* __restrict__ opt, unsigned* __restrict__ oseq) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        bool keep;
        if (flags) keep = flags[i] != 0;
        else {
            const unsigned q = seq[i];
            bool d = false;
#pragma unroll
            for (int k = 0; k < 4; k++) d = d || (k < drop.n && q == drop.s[k]);
            keep = !d;
        }
        merge_old_item(key, pt, seq, keep, rank, i, nkey, n_new, okey, opt, oseq);
    } else if (i < n + n_new) merge_new_item(nkey, npt, (int)(i - n), nseq, key, rank, n, okey, opt, oseq);
}
__global__ void k_vox_heads64(const unsigned long long* __restrict__ keys, long long n, int* __restrict__ flags) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (keys[i] < ~0ull - 1ull && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}
// k_vox_centroid on the sorted ring, ONE THREAD PER VOXEL (members are consecutive, no indirection; the same sequential f32 sums in the same
// order).  With one thread per POINT only the run heads worked — ~15 % of the lanes on a 50-keyframe ring (6.6 points per voxel), 55 us per
// commit; the heads' positions are compacted first (k_vox_head_pos) and the launch is dense.
__global__ void k_vox_head_pos(const int* __restrict__ flags, const int* __restrict__ slot, long long n, int* __restrict__ head_pos) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) head_pos[slot[i]] = (int)i;
}
constexpr int kBoxBanks = 32;      // x 128 bytes
__device__ __forceinline__ unsigned f2ord_v(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // order-preserving float -> uint (as lili_s2m.hip)
// `boxz` (may be null): bounding box of the centroids in the ZERO-INITIALISED form — words 0-2 hold ~ordered(min), words 3-5 ordered(max), all six maximised from zero —
// so that the fill that arms the commit's other scratch words arms it too (round 5: k_box_init + k_bbox_dev were two launches of the frame pipeline's commit)
__global__ void k_vox_centroid64(const unsigned long long* __restrict__ keys, const float4* __restrict__ pts, const int* __restrict__ head_pos /*[n_out] = where the voxels end if has_end*/,
                                 int has_end, const int* __restrict__ n_out_p, long long n, float4* __restrict__ out, int* __restrict__ out_cnt, unsigned* __restrict__ boxz) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = *n_out_p;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (o < n_out) {
        const long long i = head_pos[o];
        // the voxel's members are [i, end): up to the next voxel's head; the last voxel ends where the keys stop being voxels (non-finite points sort last)
        long long end;
        if (has_end || o + 1 < n_out) end = head_pos[o + 1];
        else { const unsigned long long k = keys[i]; end = i + 1; while (end < n && keys[end] == k) end++; }
        // sequential f32 sums in list order (CentroidPoint); the member count is known, so the loads of a trip do not wait for a key comparison
        float sx = 0.f, sy = 0.f, sz = 0.f, sa = 0.f;
        for (long long m = i; m < end; m += 8) {
            float4 pp[8];
#pragma unroll
            for (int u = 0; u < 8; u++) pp[u] = pts[m + u < end ? m + u : end - 1];
#pragma unroll
            for (int u = 0; u < 8; u++) if (m + u < end) { sx += pp[u].x; sy += pp[u].y; sz += pp[u].z; sa += pp[u].w; }
        }
        const float fn = (float)(int)(end - i);
        const float4 c = make_float4(sx / fn, sy / fn, sz / fn, sa / fn);
        out[o] = c;
        if (out_cnt) out_cnt[o] = (int)(end - i);
        if (isfinite(c.x) && isfinite(c.y) && isfinite(c.z)) { mn[0] = mx[0] = c.x; mn[1] = mx[1] = c.y; mn[2] = mx[2] = c.z; }
    }
    if (!boxz) return;
    // one set of six atomics per WORKGROUP, spread over kBoxBanks banks of 128 bytes (same-address atomics serialise at 12-20 ns each: one set per wave into one
    // bank made this launch 4.7 -> 13.7 us on a frame's ring map and 5 -> 52 us on the back end's 152 k-voxel map); the host folds the banks
    __shared__ float smn[4][3], smx[4][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int s = 32; s > 0; s >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], s)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], s)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = smn[0][k], b = smx[0][k];
        for (int w = 1; w < 4; w++) { a = fminf(a, smn[w][k]); b = fmaxf(b, smx[w][k]); }
        unsigned* bank = boxz + (blockIdx.x % kBoxBanks) * 32;
        if (a <= b) { atomicMax(&bank[k], ~f2ord_v(a)); atomicMax(&bank[3 + k], f2ord_v(b)); }
    }
}

// transformCloud — L/src/BackendFusion.cpp:713-790: p' = q * p + t in f64, stored f32; aux carried along
__global__ void k_transform_cloud(const float4* __restrict__ in, int n, dq q, d3 t, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = in[i];
    d3 r = qrot(q, d3{(double)p.x, (double)p.y, (double)p.z}) + t;
    out[i] = make_float4((float)r.x, (float)r.y, (float)r.z, p.w);
}

// the same with the pose read from a slot's device state (lili_frontend_frame: the keyframe is pushed at the pose the matcher has just reached, before the
// host has seen it) — same expression, same operands
__global__ void k_transform_cloud_state(const float4* __restrict__ in, int n, const SlotState* __restrict__ st, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dq q{st->pose[3], st->pose[4], st->pose[5], st->pose[6]};
    const d3 t{st->pose[0], st->pose[1], st->pose[2]};
    float4 p = in[i];
    d3 r = qrot(q, d3{(double)p.x, (double)p.y, (double)p.z}) + t;
    out[i] = make_float4((float)r.x, (float)r.y, (float)r.z, p.w);
}

}  // namespace lili

namespace lili_detail {
struct Keyframe { DevBuf pts; int n = 0; unsigned seq = 0; };
// the ring of one kind, sorted by absolute voxel key (older keyframes first inside a voxel): see k_merge_old / k_merge_new
struct SortedRing {
    DevBuf key[2], pt[2], seq[2];
    int cur = 0;
    long long n = 0;
    float leaf = 0.f;
    bool valid = false;
    std::vector<std::pair<unsigned, int>> members;      // (sequence number, points) of the keyframes it holds, oldest first
    void release() { for (int b = 0; b < 2; b++) { key[b].release(); pt[b].release(); seq[b].release(); } valid = false; n = 0; members.clear(); }
};
struct VoxelBuffers {
    DevBuf keys_a, keys_b, vals_a, vals_b, hist, hist_scan, sums, flags, slots, out, out_cnt, in, concat, concat_tab;
    std::vector<Keyframe*> ring[2];    // per kind, oldest first
    SortedRing sorted[2];
    std::vector<Keyframe*> pool;       // popped keyframes: their device buffers are reused by the next push (no hipMalloc / hipFree per keyframe: each is a device-wide synchronisation of ~100 us)
    unsigned next_seq = 1;
    DevBuf head_pos;
    DevBuf kf_key, kf_pt, seg_seq;     // the new keyframe sorted by key; sequence numbers of the concatenation's segments
    std::vector<unsigned> seg_seq_host;
    int incremental_commits = 0, full_commits = 0;
    std::vector<ConcatSeg> seg_host;   // the table k_concat reads, kept alive until the upload has certainly happened (an async copy from a local vector)
    int n_out = 0;
    // round 5: a filter whose launches are enqueued but whose count has not come back yet (voxel_filter_enqueue / _complete); the second output (`alt`: the queries of
    // the frame pipeline, filtered while the local map in `out` is still being indexed); the key bits the last measured filter of `guess_leaf` needed
    struct Pending { int mode = 0; const float4* d_pts = nullptr; int n = 0; float leaf = 0; bool need_order = false, alt = false; int res[2] = {0, 0}; int n_out = 0; } pend;
    DevBuf spts, qout, qout_cnt;
    int qn_out = 0;
    // per use site: [0] the first output (local-map rebuild, lili_voxel_filter), [1] the second (`alt`: the frame pipeline's query filter) — a large-extent map
    // and a small-extent scan of the same leaf size no longer undo each other's guess.  A guess only GROWS at once; it shrinks after eight filters in a row needed fewer bits.
    int bits_guess[2] = {0, 0}; float guess_leaf[2] = {0, 0}; int guess_low[2] = {0, 0};
    int key_guesses = 0, key_guess_misses = 0;
    unsigned out_box[6] = {0, 0, 0, 0, 0, 0};   // bounding box of `out` (ordered-uint words, k_bbox_dev) after an incremental commit
    void release() {
        for (DevBuf* b : {&keys_a, &keys_b, &vals_a, &vals_b, &hist, &hist_scan, &sums, &flags, &slots, &out, &out_cnt, &in, &concat, &concat_tab}) b->release();
        for (auto& r : ring) { for (auto* k : r) { k->pts.release(); delete k; } r.clear(); }
        for (auto* k : pool) { k->pts.release(); delete k; }
        pool.clear();
        for (auto& sr : sorted) sr.release();
        kf_key.release(); kf_pt.release(); seg_seq.release(); head_pos.release(); spts.release(); qout.release(); qout_cnt.release();
    }
};
}  // namespace lili_detail

// concatenation of the ring's keyframes: thread i finds its keyframe by bisection over the (ascending) first positions
__global__ void k_concat(const lili_detail::ConcatSeg* __restrict__ segs, int n_seg, long long total, float4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].first <= i) lo = mid; else hi = mid - 1; }
    out[i] = segs[lo].src[i - segs[lo].first];
}

static lili_detail::VoxelBuffers* vox_of(lili_ctx* ctx) {
    if (!ctx->ext_voxel) { ctx->ext_voxel = new lili_detail::VoxelBuffers(); ctx->ext_voxel_free = [](void* p) { auto* r = static_cast<lili_detail::VoxelBuffers*>(p); r->release(); delete r; }; }
    return static_cast<lili_detail::VoxelBuffers*>(ctx->ext_voxel);
}

// Exclusive scan of a SHORT array (<= 64 k words: the head / keep flags of a frame's voxel filter and ring merge, the digit histograms of a radix pass) in ONE launch
// with NO communication between workgroups (round 5, second version): tile t (2048 words, 256 threads) sums the words in FRONT of it itself — at most 30 tiles of
// 8 KB out of the L2, read with 16-byte loads by every lane at once — and scans its own words behind that offset.  The one-workgroup scan it replaces (k_scan_single)
// took 10-16 us for the 20-40 k flags: one CU reads and writes 160 KB each at ~50 GB/s; the tiles take the read side of the LAST tile, ~3 us, with nothing to wait for.
constexpr int kScanTile = 2048;
__global__ __launch_bounds__(256) void k_scan_tiles(const int* __restrict__ in, int n, int* __restrict__ out /*[n+1]*/) {
    __shared__ int w_front[4], w_own[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * kScanTile;
    // (`in` is a DevBuf: 16-byte aligned, >= 256 bytes of slack behind its n words; words behind n count as zero)
    int front = 0;
    for (int i = tid * 4; i < base; i += 8 * 1024) {      // eight independent 16-byte loads in flight per lane (one after the other they were most of the launch)
        int4 q[8];
#pragma unroll
        for (int u = 0; u < 8; u++) q[u] = *reinterpret_cast<const int4*>(in + min(i + u * 1024, base - 4));      // (a group beyond the front re-reads its last one and is not counted)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) front += i + u * 1024 < base ? (q[u].x + q[u].y) + (q[u].z + q[u].w) : 0;
    }
    const int i0 = base + tid * 8;
    int v[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int i = i0 + 4 * h;
        int4 q = make_int4(0, 0, 0, 0);
        if (i < n) q = *reinterpret_cast<const int4*>(in + i);
        v[4 * h] = q.x; v[4 * h + 1] = i + 1 < n ? q.y : 0; v[4 * h + 2] = i + 2 < n ? q.z : 0; v[4 * h + 3] = i + 3 < n ? q.w : 0;
    }
    int s8 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { const int t = v[k]; v[k] = s8; s8 += t; }      // the items become their exclusive prefixes inside the thread
    int inc = s8;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) front += __shfl_xor(front, o);
    if (lane == 63) w_own[wave] = inc;
    if (lane == 0) w_front[wave] = front;
    __syncthreads();
    int ofs = (w_front[0] + w_front[1]) + (w_front[2] + w_front[3]);
    int tile_total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int t = w_own[w]; if (w < wave) ofs += t; tile_total += t; }
    const int e0 = ofs + inc - s8;
    if (i0 + 8 <= n) {
        *reinterpret_cast<int4*>(out + i0) = make_int4(e0 + v[0], e0 + v[1], e0 + v[2], e0 + v[3]);
        *reinterpret_cast<int4*>(out + i0 + 4) = make_int4(e0 + v[4], e0 + v[5], e0 + v[6], e0 + v[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < n) out[i0 + k] = e0 + v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) out[n] = (w_front[0] + w_front[1]) + (w_front[2] + w_front[3]) + tile_total;
}
// k_scan_tiles over flags that are never stored (round 5): the launches that only wrote a flag per item for the scan to read are folded into it.  MODE
//   kScanKeep   : item i of the sorted ring survives the merge (its keyframe is not one of `drop`)                       — was k_keep_flags + scan
//   kScanHead64 : item i starts a voxel of the sorted ring (64-bit absolute keys); head_pos[its slot] = i on the way      — was k_vox_heads64 + scan + k_vox_head_pos
//   kScanHead32 : item i starts a voxel of a sorted cloud (32-bit keys, `sentinel` = no voxel); spts[i] = pts[vals[i]]     — was k_vox_heads + scan
// Same tiles, same sums; the items in front of a tile are re-derived from the keys (the same bytes the flags would have been, twice for 64-bit keys).
enum { kScanKeep = 1, kScanHead64 = 2, kScanHead32 = 3 };
struct ScanFlagArgs { const void* a; DropSeqs drop; unsigned sentinel; const int* vals; const float4* pts; float4* spts; int* head_pos; };
// the raw words behind four consecutive flags (two steps, so that a caller can request several groups before it looks at any: a load behind a branch, or mixed with
// the arithmetic of the previous group, waits for its own round trip)
struct ScanRaw { uint4 a, b; unsigned long long prev; };
template <int MODE> __device__ __forceinline__ ScanRaw scan_flags_load(const ScanFlagArgs& A, int i /*multiple of 4*/) {
    ScanRaw r{};
    if (MODE == kScanKeep) r.a = *reinterpret_cast<const uint4*>(static_cast<const unsigned*>(A.a) + i);
    else if (MODE == kScanHead64) {
        const unsigned long long* key = static_cast<const unsigned long long*>(A.a);
        r.a = *reinterpret_cast<const uint4*>(key + i); r.b = *reinterpret_cast<const uint4*>(key + i + 2); r.prev = key[max(i - 1, 0)];
    } else {
        const unsigned* key = static_cast<const unsigned*>(A.a);
        r.a = *reinterpret_cast<const uint4*>(key + i); r.prev = key[max(i - 1, 0)];
    }
    return r;
}
// v[k] = flag of item i + k; endm (head modes) = bit k set where item i + k is the FIRST item that is no voxel (non-finite points sort last): where the last voxel ends
template <int MODE> __device__ __forceinline__ void scan_flags_eval(const ScanFlagArgs& A, const ScanRaw& r, int i, int n, int v[4], unsigned* endm = nullptr) {
    if (MODE == kScanKeep) {
        const unsigned s[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bool drop = false;
#pragma unroll
            for (int d = 0; d < 4; d++) drop = drop || (d < A.drop.n && s[k] == A.drop.s[d]);
            v[k] = (i + k < n && !drop) ? 1 : 0;
        }
    } else if (MODE == kScanHead64) {
        const unsigned long long kk[5] = {r.prev, ((unsigned long long)r.a.y << 32) | r.a.x, ((unsigned long long)r.a.w << 32) | r.a.z, ((unsigned long long)r.b.y << 32) | r.b.x,
                                          ((unsigned long long)r.b.w << 32) | r.b.z};
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i + k < n && kk[k + 1] < ~0ull - 1ull && (i + k == 0 || kk[k + 1] != kk[k])) ? 1 : 0;
        if (endm) { *endm = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) if (i + k < n && kk[k + 1] >= ~0ull - 1ull && (i + k == 0 || kk[k] < ~0ull - 1ull)) *endm |= 1u << k; }
    } else {
        const unsigned kk[5] = {(unsigned)r.prev, r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i + k < n && kk[k + 1] != A.sentinel && (i + k == 0 || kk[k + 1] != kk[k])) ? 1 : 0;
        if (endm) { *endm = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) if (i + k < n && kk[k + 1] == A.sentinel && (i + k == 0 || kk[k] != A.sentinel)) *endm |= 1u << k; }
    }
}
// 1024 threads: the items in FRONT of the tile are read by sixteen waves in one batch of loads per lane (two for the 64-bit keys of a 40 k ring) — the launch lasts as
// long as its last tile's chain of dependent round trips, ~2 us each on data another launch has just written; with 256 threads and one group in flight it was 39 of them.
// The tile's own 2048 items belong to the first four waves, eight per lane.
constexpr int kScanFlagThreads = 1024;
template <int MODE> __global__ __launch_bounds__(kScanFlagThreads) void k_scan_flags(ScanFlagArgs A, int n, int* __restrict__ out /*[n+1]*/) {
    __shared__ int w_front[kScanFlagThreads / 64], w_own[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * kScanTile;
    constexpr int kB = MODE == kScanHead64 ? 5 : MODE == kScanHead32 ? 8 : 16;      // groups of four items in flight per lane (1024 threads: 128 registers each)
    // the tile's own items (first four waves, eight per lane) are requested first, so that they travel with the front's
    const int i0 = base + tid * 8;
    const int last4 = (n - 1) & ~3;      // (n >= 1) a group behind the end re-reads the last one; its flags are zero by position
    const ScanRaw r0 = scan_flags_load<MODE>(A, min(i0, last4)), r1 = scan_flags_load<MODE>(A, min(i0 + 4, last4));      // (the other twelve waves load too and drop it: no branch in front of the batch)
    int src[8];
#pragma unroll
    for (int k = 0; k < 8; k++) src[k] = MODE == kScanHead32 ? A.vals[min(i0 + k, n - 1)] : 0;
    int front = 0;
    for (int i = tid * 4; i < base; i += kB * 4 * kScanFlagThreads) {
        ScanRaw raw[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) raw[u] = scan_flags_load<MODE>(A, min(i + u * 4 * kScanFlagThreads, base - 4));      // (a group beyond the front re-reads its last one and is not counted)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kB; u++) {
            int f[4];
            scan_flags_eval<MODE>(A, raw[u], min(i + u * 4 * kScanFlagThreads, base - 4), n, f);
            front += i + u * 4 * kScanFlagThreads < base ? (f[0] + f[1]) + (f[2] + f[3]) : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) front += __shfl_xor(front, o);
    if (lane == 0) w_front[wave] = front;
    int v[8], flag[8];
    unsigned endm = 0;
    int s8 = 0, inc = 0;
    if (wave < 4) {
        unsigned e0m = 0, e1m = 0;
        scan_flags_eval<MODE>(A, r0, i0, n, v, MODE == kScanKeep ? nullptr : &e0m);
        scan_flags_eval<MODE>(A, r1, i0 + 4, n, v + 4, MODE == kScanKeep ? nullptr : &e1m);
        endm = e0m | (e1m << 4);
        if (MODE == kScanHead32) {
            const float4* __restrict__ gp = A.pts;
            float4* __restrict__ sp = A.spts;
            const float4 g0 = gp[src[0]], g1 = gp[src[1]], g2 = gp[src[2]], g3 = gp[src[3]], g4 = gp[src[4]], g5 = gp[src[5]], g6 = gp[src[6]], g7 = gp[src[7]];
            if (i0 < n) {      // (up to seven points behind the end land in the buffer's slack: DevBuf keeps 256 bytes)
                sp[i0] = g0; sp[i0 + 1] = g1; sp[i0 + 2] = g2; sp[i0 + 3] = g3; sp[i0 + 4] = g4; sp[i0 + 5] = g5; sp[i0 + 6] = g6; sp[i0 + 7] = g7;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { flag[k] = v[k]; v[k] = s8; s8 += flag[k]; }
        inc = s8;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) w_own[wave] = inc;
    }
    __syncthreads();
    if (wave >= 4) return;
    int ofs = 0;
#pragma unroll
    for (int w = 0; w < kScanFlagThreads / 64; w++) ofs += w_front[w];
    const int front_total = ofs;
    int tile_total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int t = w_own[w]; if (w < wave) ofs += t; tile_total += t; }
    const int e0 = ofs + inc - s8;
    if (i0 + 8 <= n) {
        *reinterpret_cast<int4*>(out + i0) = make_int4(e0 + v[0], e0 + v[1], e0 + v[2], e0 + v[3]);
        *reinterpret_cast<int4*>(out + i0 + 4) = make_int4(e0 + v[4], e0 + v[5], e0 + v[6], e0 + v[7]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < n) out[i0 + k] = e0 + v[k];
    }
    if (MODE != kScanKeep && A.head_pos) {      // (wave-uniform) compacted head positions, closed by the position where the voxels end: head_pos[number of voxels]
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (flag[k]) A.head_pos[e0 + v[k]] = i0 + k;
            if (endm & (1u << k)) A.head_pos[e0 + v[k]] = i0 + k;      // (no head at or behind it: its prefix is the number of voxels)
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        out[n] = front_total + tile_total;
        if (MODE != kScanKeep && A.head_pos) {      // no item that is no voxel: the voxels end at n
            bool none;
            if (MODE == kScanHead64) none = static_cast<const unsigned long long*>(A.a)[n - 1] < ~0ull - 1ull;
            else none = static_cast<const unsigned*>(A.a)[n - 1] != A.sentinel;
            if (none) A.head_pos[front_total + tile_total] = n;
        }
    }
}
constexpr long long kScanFlagsMax = 65536;      // (beyond: flags in memory and the three-kernel scan)

static int exclusive_scan(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const int* in, int64_t n, int* out /*[n+1]*/) {
    if (n <= 65536) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(std::max(1, nblocks(n, kScanTile))), dim3(256), 0, ctx->stream, in, (int)n, out);
        HIPCHK(hipGetLastError());
        return LILI_OK;
    }
    const int nb = nblocks(n, 2048);
    HIPCHK(V->sums.ensure((size_t)nb * sizeof(int)));
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, ctx->stream, in, n, V->sums.as<int>());
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, ctx->stream, V->sums.as<int>(), nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, ctx->stream, in, n, V->sums.as<int>(), out);
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// sorts (keys_a, vals_a) by the low `bits` bits, stable; result ends in (keys_a, vals_a).  8-bit digits (option sort_digit_bits = 4: the
// round-2 passes, for A/B)
// `ride`: the digit counts of pass 0 are already in table 0 of V->hist (k_vox_key_dev counted them) and tables 1 .. are zero: every scatter pass counts the next pass's
// digits as it places its keys — a pass is ONE launch (round 6; the caller has sized and cleared V->hist through radix_ride_tables).
static bool radix_can_ride(const lili_ctx* ctx, int n) {
    if (ctx->sort_digit_bits == 4 || !ctx->sort_fused_scan || !ctx->sort_ride_hist) return false;
    const int items8 = n <= 262144 ? 4 : 16;
    return (n + 256 * items8 - 1) / (256 * items8) <= ctx->sort_fused_max_tiles;
}
static void radix_geometry(int n, int& items8, int& nb) { items8 = n <= 262144 ? 4 : 16; nb = (n + 256 * items8 - 1) / (256 * items8); }
static int radix_sort(lili_ctx* ctx, lili_detail::VoxelBuffers* V, int n, int bits, bool ride = false) {
    const int dbits = ctx->sort_digit_bits == 4 ? 4 : 8, ndig = 1 << dbits;
    // 8-bit digits: keys per tile by size — a tile is walked in rounds of 256 keys (one block barrier set per round), so a keyframe's 20 k keys
    // spread over 20 four-round tiles sort in a third of the time five 16-round tiles take; 1 M keys keep 4096-key tiles (digit table 62 k words)
    const int items8 = n <= 262144 ? 4 : kSortItems8;      // (one-round tiles made the digit table — 256 words per tile — the bottleneck: its scan is one workgroup)
    const int nb = nblocks(n, dbits == 8 ? kSortBlock * items8 : kSortTile);
    if (!ride) HIPCHK(V->hist.ensure((size_t)ndig * nb * sizeof(int)));
    HIPCHK(V->hist_scan.ensure(((size_t)ndig * nb + 1) * sizeof(int)));
    HIPCHK(V->keys_b.ensure((size_t)n * 4)); HIPCHK(V->vals_b.ensure((size_t)n * 4));
    const int passes = std::max(1, (bits + dbits - 1) / dbits);
    if (ride) {
        for (int p = 0; p < passes; p++) {
            int* tab = V->hist.as<int>() + (size_t)p * ndig * nb;
            hipLaunchKernelGGL(k_sort_scatter8, dim3(nb), dim3(kSortBlock), 0, ctx->stream, V->keys_a.as<unsigned>(), V->vals_a.as<int>(), n, dbits * p, nb, items8, (const int*)tab, 1,
                               V->keys_b.as<unsigned>(), V->vals_b.as<int>(), p + 1 < passes ? tab + (size_t)ndig * nb : (int*)nullptr, dbits * (p + 1));
            HIPCHK(hipGetLastError());
            V->keys_a.swap(V->keys_b); V->vals_a.swap(V->vals_b);
        }
        return LILI_OK;
    }
    for (int p = 0; p < passes; p++) {
        const int shift = dbits * p;
        unsigned *ka = V->keys_a.as<unsigned>(), *kb = V->keys_b.as<unsigned>();
        int *va = V->vals_a.as<int>(), *vb = V->vals_b.as<int>();
        if (dbits == 8) hipLaunchKernelGGL(k_sort_hist8, dim3(nb), dim3(kSortBlock), 0, ctx->stream, ka, n, shift, nb, items8, V->hist.as<int>());
        else hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(kSortBlock), 0, ctx->stream, ka, n, shift, nb, V->hist.as<int>());
        const bool fused_scan = ctx->sort_fused_scan && dbits == 8 && nb <= ctx->sort_fused_max_tiles;      // (see k_sort_scatter8)
        if (!fused_scan) { const int rc = exclusive_scan(ctx, V, V->hist.as<int>(), (int64_t)ndig * nb, V->hist_scan.as<int>()); if (rc != LILI_OK) return rc; }
        if (dbits == 8) hipLaunchKernelGGL(k_sort_scatter8, dim3(nb), dim3(kSortBlock), 0, ctx->stream, ka, va, n, shift, nb, items8,
                                           fused_scan ? V->hist.as<int>() : V->hist_scan.as<int>(), fused_scan ? 1 : 0, kb, vb, (int*)nullptr, 0);
        else hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(kSortBlock), 0, ctx->stream, ka, va, n, shift, nb, V->hist_scan.as<int>(), kb, vb);
        HIPCHK(hipGetLastError());
        V->keys_a.swap(V->keys_b); V->vals_a.swap(V->vals_b);      // the sorted pairs are the new `a` (buffers trade places, nothing is copied)
    }
    return LILI_OK;
}

// Stable order of a device cloud by pcl::VoxelGrid's voxel index (box-relative, App. B2): keys in V->keys_a, the order (source indices) in
// V->vals_a.  Blocking: the bounding box is read back.
static int voxel_sort(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const float4* d_pts, int n, float leaf, VoxDev& P, bool alt = false) {
    unsigned* d_mm = ctx->misc.as<unsigned>();
    hipLaunchKernelGGL(k_box_init, dim3(1), dim3(64), 0, ctx->stream, d_mm);
    hipLaunchKernelGGL(k_bbox, dim3(std::min(nblocks(n, kBlock), 512)), dim3(kBlock), 0, ctx->stream, d_pts, n, d_mm);
    unsigned mm[6];
    { int rb = lili_readback_add(ctx, mm, d_mm, sizeof(mm)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    auto dec = [](unsigned u) { unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; };
    P = VoxDev{};
    P.inv_leaf = 1.0f / leaf;
    int div_b[3];
    for (int k = 0; k < 3; k++) {
        float mn = dec(mm[k]), mx = dec(mm[3 + k]);
        if (!(mn <= mx)) return ctx->fail(LILI_E_ARG, "voxel_filter: cloud holds no finite point");
        P.min_b[k] = (int)std::floor(mn * P.inv_leaf);
        div_b[k] = (int)std::floor(mx * P.inv_leaf) - P.min_b[k] + 1;
    }
    const double total = (double)div_b[0] * (double)div_b[1] * (double)div_b[2];
    if (total > 2147483647.0) return ctx->fail(LILI_E_ARG, "voxel_filter: leaf size too small for the cloud extent (voxel index would overflow int32, as in PCL)");
    P.mul[0] = 1; P.mul[1] = div_b[0]; P.mul[2] = div_b[0] * div_b[1];
    P.sentinel = (unsigned)total;                 // <= 2^31 - 1
    int bits = 1; while (bits < 32 && (1ull << bits) < (unsigned long long)total + 1ull) bits++;   // keys 0 .. total (sentinel included)
    V->bits_guess[alt ? 1 : 0] = std::max(8, (bits + 7) / 8 * 8); V->guess_leaf[alt ? 1 : 0] = leaf; V->guess_low[alt ? 1 : 0] = 0;      // what the next filter of this site and leaf size may assume (voxel_filter_enqueue)
    HIPCHK(V->keys_a.ensure((size_t)n * 4)); HIPCHK(V->vals_a.ensure((size_t)n * 4));
    hipLaunchKernelGGL(k_vox_key, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, d_pts, n, P, V->keys_a.as<unsigned>(), V->vals_a.as<int>());
    return radix_sort(ctx, V, n, bits);
}

// the launches behind the sort: head flags (and the points in sorted order), their scan, the centroids
static int voxel_filter_tail(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const float4* d_pts, int n, unsigned sentinel, bool alt) {
    DevBuf& out = alt ? V->qout : V->out; DevBuf& out_cnt = alt ? V->qout_cnt : V->out_cnt;
    HIPCHK(V->flags.ensure((size_t)n * 4)); HIPCHK(V->slots.ensure(((size_t)n + 1) * 4)); HIPCHK(V->spts.ensure((size_t)n * 16)); HIPCHK(V->head_pos.ensure(((size_t)n + 1) * 4));
    HIPCHK(out.ensure((size_t)n * 16)); HIPCHK(out_cnt.ensure((size_t)n * 4));
    if (n <= kScanFlagsMax) {      // head flags, the points in sorted order and the scan in one launch
        ScanFlagArgs A{}; A.a = V->keys_a.p; A.sentinel = sentinel; A.vals = V->vals_a.as<int>(); A.pts = d_pts; A.spts = V->spts.as<float4>(); A.head_pos = V->head_pos.as<int>();
        hipLaunchKernelGGL(k_scan_flags<kScanHead32>, dim3(nblocks(n, kScanTile)), dim3(kScanFlagThreads), 0, ctx->stream, A, n, V->slots.as<int>());
    } else {
        hipLaunchKernelGGL(k_vox_heads, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, V->keys_a.as<unsigned>(), V->vals_a.as<int>(), d_pts, n, sentinel, V->flags.as<int>(), V->spts.as<float4>());
        const int rc = exclusive_scan(ctx, V, V->flags.as<int>(), n, V->slots.as<int>());
        if (rc != LILI_OK) return rc;
        hipLaunchKernelGGL(k_vox_head_pos, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, V->flags.as<int>(), V->slots.as<int>(), (long long)n, V->head_pos.as<int>());
    }
    hipLaunchKernelGGL(k_vox_centroid, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, V->keys_a.as<unsigned>(), V->slots.as<int>(), V->head_pos.as<int>(), n <= kScanFlagsMax ? 1 : 0, V->spts.as<float4>(), n, sentinel,
                       out.as<float4>(), out_cnt.as<int>());
    HIPCHK(hipGetLastError());
    return LILI_OK;
}

// VoxelGrid of a device float4 cloud the measured way: the bounding box comes to the host, the host sizes the keys.  Blocking (two small read-backs).
static int voxel_filter_measured(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const float4* d_pts, int n, float leaf, bool alt) {
    VoxDev P;
    int rc = voxel_sort(ctx, V, d_pts, n, leaf, P, alt);
    if (rc != LILI_OK) return rc;
    rc = voxel_filter_tail(ctx, V, d_pts, n, P.sentinel, alt);
    if (rc != LILI_OK) return rc;
    int& n_out = alt ? V->qn_out : V->n_out;
    { int rb = lili_readback_add(ctx, &n_out, V->slots.as<int>() + n, sizeof(int)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    return LILI_OK;
}

// VoxelGrid of a device float4 cloud in two halves (round 5): _enqueue launches what can be launched without knowing anything about the cloud and adds the read-backs
// of the outcome to the context's pending list; the caller synchronises (lili_readback_finish — or lets a synchronisation it has anyway deliver them: the frame pipeline
// enqueues its query filter behind the local map's index build) and calls _complete, which accepts the outcome or repeats the filter the measured way.
//   <= 8192 points (and no caller that needs the sort's order): k_voxel_small, one launch;
//   else, when a filter of this leaf size has been measured before: the box stays on the device (k_vox_key_dev) and the key bits are guessed from that filter;
//   else nothing is enqueued and _complete runs the measured filter.
// Result in V->out / V->out_cnt / V->n_out, or in V->qout / V->qout_cnt / V->qn_out (`alt`).
enum { kPendDone = 0, kPendMeasure = 1, kPendSmall = 2, kPendGuess = 3 };
static int voxel_filter_enqueue(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const float4* d_pts, int n, float leaf, bool need_order, bool alt, bool box_zeroed = false) {
    auto& Q = V->pend;
    Q = lili_detail::VoxelBuffers::Pending{};
    Q.d_pts = d_pts; Q.n = n; Q.leaf = leaf; Q.need_order = need_order; Q.alt = alt;
    (alt ? V->qn_out : V->n_out) = 0;
    if (n == 0) { Q.mode = kPendDone; return LILI_OK; }
    DevBuf& out = alt ? V->qout : V->out; DevBuf& out_cnt = alt ? V->qout_cnt : V->out_cnt;
    int* d_res = reinterpret_cast<int*>(ctx->misc.as<char>() + 1024);
    if (n <= kVoxSmallMax && ctx->voxel_small && !need_order) {
        HIPCHK(out.ensure((size_t)n * 16)); HIPCHK(out_cnt.ensure((size_t)n * 4));
        hipLaunchKernelGGL(k_voxel_small, dim3(1), dim3(kVoxSmallThreads), 0, ctx->stream, d_pts, n, 1.0f / leaf, out.as<float4>(), out_cnt.as<int>(), d_res);
        HIPCHK(hipGetLastError());
        Q.mode = kPendSmall;
        return lili_readback_add(ctx, Q.res, d_res, sizeof(Q.res));
    }
    if (ctx->voxel_guess_bits && V->bits_guess[alt ? 1 : 0] > 0 && V->guess_leaf[alt ? 1 : 0] == leaf) {
        unsigned* d_mm = ctx->misc.as<unsigned>();
        // (box_zeroed: the caller vouches that the first six words of ctx->misc are zero — the frame pipeline enqueues this behind an index build's scratch fill)
        const int bits = V->bits_guess[alt ? 1 : 0];
        // round 6: the sort's histograms ride on the key kernel and on the scatter passes (three launches fewer per filter); the box pass clears their tables
        const bool ride = box_zeroed && radix_can_ride(ctx, n);
        int items8 = 0, nb_sort = 0;
        radix_geometry(n, items8, nb_sort);
        const int passes = std::max(1, (bits + 7) / 8);
        if (ride) HIPCHK(V->hist.ensure((size_t)passes * 256 * nb_sort * sizeof(int)));
        if (box_zeroed) hipLaunchKernelGGL(k_bbox_z, dim3(std::min(nblocks(n, 256), 512)), dim3(256), 0, ctx->stream, d_pts, n, d_mm, ride ? V->hist.as<int>() : (int*)nullptr,
                                           ride ? passes * 256 * nb_sort : 0);
        else {
            hipLaunchKernelGGL(k_box_init, dim3(1), dim3(64), 0, ctx->stream, d_mm);
            hipLaunchKernelGGL(k_bbox, dim3(std::min(nblocks(n, kBlock), 512)), dim3(kBlock), 0, ctx->stream, d_pts, n, d_mm);
        }
        HIPCHK(V->keys_a.ensure((size_t)n * 4)); HIPCHK(V->vals_a.ensure((size_t)n * 4));
        hipLaunchKernelGGL(k_vox_key_dev, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, d_pts, n, 1.0f / leaf, (const unsigned*)d_mm, box_zeroed ? 1 : 0, bits, V->keys_a.as<unsigned>(),
                           V->vals_a.as<int>(), d_res, ride ? V->hist.as<int>() : (int*)nullptr, nb_sort, kSortBlock * items8);
        int rc = radix_sort(ctx, V, n, bits, ride);
        if (rc != LILI_OK) return rc;
        rc = voxel_filter_tail(ctx, V, d_pts, n, bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u, alt);
        if (rc != LILI_OK) return rc;
        Q.mode = kPendGuess;
        rc = lili_readback_add(ctx, &Q.n_out, V->slots.as<int>() + n, sizeof(int));
        if (rc == LILI_OK) rc = lili_readback_add(ctx, Q.res, d_res, sizeof(Q.res));
        return rc;
    }
    Q.mode = kPendMeasure;
    return LILI_OK;
}
static int voxel_filter_complete(lili_ctx* ctx, lili_detail::VoxelBuffers* V) {
    auto& Q = V->pend;
    int& n_out = Q.alt ? V->qn_out : V->n_out;
    const int mode = Q.mode;
    Q.mode = kPendDone;
    if (mode == kPendDone) return LILI_OK;
    if (mode == kPendSmall) {
        if (!Q.res[1]) { n_out = Q.res[0]; return LILI_OK; }
        if (Q.res[1] == 2) return ctx->fail(LILI_E_ARG, "voxel_filter: cloud holds no finite point");
        // (voxel index beyond 31 bits: the measured path reports it the way PCL does)
    }
    if (mode == kPendGuess) {
        V->key_guesses++;
        if (Q.res[0] == 0) {
            n_out = Q.n_out;
            const int site = Q.alt ? 1 : 0, need = std::max(8, (Q.res[1] + 7) / 8 * 8);
            if (need >= V->bits_guess[site]) { V->bits_guess[site] = need; V->guess_low[site] = 0; }
            else if (++V->guess_low[site] >= 8) { V->bits_guess[site] = need; V->guess_low[site] = 0; }
            return LILI_OK;
        }
        V->key_guess_misses++;      // more key bits than guessed, or one of the error cases: the measured path sorts it out
    }
    return voxel_filter_measured(ctx, V, Q.d_pts, Q.n, Q.leaf, Q.alt);
}
// Blocking: result in V->out / V->out_cnt, V->n_out.
// need_order: the caller goes on to use the sort's order (V->vals_a: lili_localmap_commit builds its sorted ring from it) — k_voxel_small leaves none
static int voxel_filter_device(lili_ctx* ctx, lili_detail::VoxelBuffers* V, const float4* d_pts, int n, float leaf, bool need_order = false, bool alt = false) {
    int rc = voxel_filter_enqueue(ctx, V, d_pts, n, leaf, need_order, alt);
    if (rc != LILI_OK) return rc;
    if (V->pend.mode == kPendSmall || V->pend.mode == kPendGuess) { rc = lili_readback_finish(ctx); if (rc != LILI_OK) return rc; }
    return voxel_filter_complete(ctx, V);
}

extern "C" {

int lili_voxel_filter(lili_ctx* ctx, const lili_cloud* cloud, float leaf, lili_feature_out* out, int32_t* counts) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(cloud && out && leaf > 0, "voxel_filter: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    auto* V = vox_of(ctx);
    int rc = lili_ingest_cloud(ctx, cloud, V->in);
    if (rc != LILI_OK) return rc;
    rc = voxel_filter_device(ctx, V, V->in.as<float4>(), (int)cloud->n, leaf);
    if (rc != LILI_OK) return rc;
    out->count = (size_t)V->n_out;
    size_t k = std::min(out->count, out->capacity);
    if (out->data && k) {
        size_t stride = out->stride ? out->stride : 16;
        ARGCHK(stride >= 16, "voxel_filter: stride must be >= 16");
        const hipMemcpyKind kind = out->mem == LILI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (stride == 16) HIPCHK(hipMemcpyAsync(out->data, V->out.p, k * 16, kind, ctx->stream));
        else HIPCHK(hipMemcpy2DAsync(out->data, stride, V->out.p, 16, 16, k, kind, ctx->stream));
        if (counts) HIPCHK(hipMemcpyAsync(counts, V->out_cnt.p, k * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return LILI_OK;
}

int lili_localmap_reset(lili_ctx* ctx, int kind) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(kind == 0 || kind == 1, "localmap_reset: bad kind");
    auto* V = vox_of(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto* k : V->ring[kind]) { k->pts.release(); delete k; }
    V->ring[kind].clear();
    V->sorted[kind].valid = false; V->sorted[kind].n = 0; V->sorted[kind].members.clear();
    return LILI_OK;
}

// the ring step of lili_localmap_push on a device float4 cloud; the pose either by value (t, q) or read by the kernel from `d_state`
static int localmap_push_f4(lili_ctx* ctx, int kind, const float4* d_in, int n, const double* t, const double* q, const SlotState* d_state, int width) {
    auto* V = vox_of(ctx);
    // a buffer from the pool of popped keyframes if one is large enough (the steady state: keyframes of similar size), else a new one
    lili_detail::Keyframe* kf = nullptr;
    for (size_t i = 0; i < V->pool.size(); i++) if (V->pool[i]->pts.cap >= (size_t)n * 16) { kf = V->pool[i]; V->pool.erase(V->pool.begin() + (long)i); break; }
    if (!kf) kf = new lili_detail::Keyframe();
    kf->n = n;
    kf->seq = V->next_seq++;
    if (kf->n > 0) {
        hipError_t e = kf->pts.ensure((size_t)kf->n * 16);
        if (e != hipSuccess) { delete kf; return ctx->fail(LILI_E_HIP, "localmap_push: allocation failed"); }
        if (d_state) hipLaunchKernelGGL(k_transform_cloud_state, dim3(nblocks(kf->n, 256)), dim3(256), 0, ctx->stream, d_in, kf->n, d_state, kf->pts.as<float4>());
        else hipLaunchKernelGGL(k_transform_cloud, dim3(nblocks(kf->n, 256)), dim3(256), 0, ctx->stream, d_in, kf->n,
                                dq{q[0], q[1], q[2], q[3]}, d3{t[0], t[1], t[2]}, kf->pts.as<float4>());
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { delete kf; return ctx->fail(LILI_E_HIP, std::string("localmap_push: ") + hipGetErrorString(le)); }
    }
    V->ring[kind].push_back(kf);
    while ((int)V->ring[kind].size() > width) {   // recent_*_keyframes.pop_front() (L:1449-1450)
        // the popped keyframe's buffer goes to the pool: everything that reads it was enqueued on this stream before whatever will overwrite it
        auto* old = V->ring[kind].front();
        V->ring[kind].erase(V->ring[kind].begin());
        if (V->pool.size() < 4) V->pool.push_back(old);
        else { HIPCHK(hipStreamSynchronize(ctx->stream)); old->pts.release(); delete old; }
    }
    return LILI_OK;
}

int lili_localmap_push(lili_ctx* ctx, int kind, const lili_cloud* features, const double t[3], const double q[4], int width) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK((kind == 0 || kind == 1) && features && t && q && width >= 1, "localmap_push: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    auto* V = vox_of(ctx);
    int rc = lili_ingest_cloud(ctx, features, V->in);
    if (rc != LILI_OK) return rc;
    return localmap_push_f4(ctx, kind, V->in.as<float4>(), (int)features->n, t, q, nullptr, width);
}

// One keyframe step on the sorted ring: drop the entries of the keyframes in `drop`, merge the keyframe `add` (may be null) in.
static int sorted_ring_step(lili_ctx* ctx, lili_detail::VoxelBuffers* V, lili_detail::SortedRing& S, const DropSeqs& drop, long long n_drop, lili_detail::Keyframe* add, float leaf,
                            unsigned* d_bad) {
    int n_new = add ? add->n : 0;
    if (n_new > 0 && n_new <= kVoxSmallMax && ctx->voxel_small) {      // a frame's worth of features: one launch (k_sort_keyframe_small)
        HIPCHK(V->kf_key.ensure((size_t)n_new * 8)); HIPCHK(V->kf_pt.ensure((size_t)n_new * 16)); HIPCHK(V->flags.ensure((size_t)std::max<long long>(n_new, S.n) * 4));
        // (`rank`: the scan-status words of a map build, idle during a commit; lili_localmap_commit zeroes them together with *bad)
        int* d_rank = reinterpret_cast<int*>(ctx->misc.as<char>() + kRankOff);
        const int slices = nblocks(n_new, 64);
        hipLaunchKernelGGL(k_rank_count, dim3(slices, nblocks(slices, kRankWaves)), dim3(64 * kRankWaves), 0, ctx->stream, add->pts.as<float4>(), n_new, 1.0f / leaf, d_rank, d_bad);
        hipLaunchKernelGGL(k_rank_scatter, dim3(nblocks(n_new, 256)), dim3(256), 0, ctx->stream, add->pts.as<float4>(), n_new, 1.0f / leaf, add->seq, (const int*)d_rank,
                           V->kf_pt.as<float4>(), V->kf_key.as<unsigned long long>(), V->flags.as<unsigned>());
        HIPCHK(hipGetLastError());
    } else
    if (n_new > 0) {       // the new keyframe alone, sorted by voxel (stable): 20 k points, no host round trip (the bounding box stays on the device)
        unsigned* d_mm = ctx->misc.as<unsigned>();
        hipLaunchKernelGGL(k_box_init, dim3(1), dim3(64), 0, ctx->stream, d_mm);
        hipLaunchKernelGGL(k_bbox, dim3(std::min(nblocks(n_new, kBlock), 512)), dim3(kBlock), 0, ctx->stream, add->pts.as<float4>(), n_new, d_mm);
        HIPCHK(V->keys_a.ensure((size_t)n_new * 4)); HIPCHK(V->vals_a.ensure((size_t)n_new * 4));
        hipLaunchKernelGGL(k_vox_key_packed, dim3(nblocks(n_new, 256)), dim3(256), 0, ctx->stream, add->pts.as<float4>(), n_new, 1.0f / leaf, (const unsigned*)d_mm,
                           V->keys_a.as<unsigned>(), V->vals_a.as<int>(), d_bad);
        int rc = radix_sort(ctx, V, n_new, 32);
        if (rc != LILI_OK) return rc;
        HIPCHK(V->kf_key.ensure((size_t)n_new * 8)); HIPCHK(V->kf_pt.ensure((size_t)n_new * 16)); HIPCHK(V->flags.ensure((size_t)std::max<long long>(n_new, S.n) * 4));
        hipLaunchKernelGGL(k_sorted_gather, dim3(nblocks(n_new, 256)), dim3(256), 0, ctx->stream, add->pts.as<float4>(), V->vals_a.as<int>(), n_new, 1.0f / leaf, add->seq,
                           (const lili_detail::ConcatSeg*)nullptr, (const unsigned*)nullptr, 0, V->kf_pt.as<float4>(), V->kf_key.as<unsigned long long>(), V->flags.as<unsigned>(), d_bad);
    }
    const long long n_old = S.n, n_out = n_old - n_drop + n_new;
    const int dst = 1 - S.cur;
    HIPCHK(S.key[dst].ensure((size_t)std::max<long long>(n_out, 1) * 8)); HIPCHK(S.pt[dst].ensure((size_t)std::max<long long>(n_out, 1) * 16)); HIPCHK(S.seq[dst].ensure((size_t)std::max<long long>(n_out, 1) * 4));
    HIPCHK(V->flags.ensure((size_t)std::max<long long>(n_old, 1) * 4)); HIPCHK(V->slots.ensure(((size_t)n_old + 1) * 4));
    const bool fused_flags = n_old <= kScanFlagsMax;      // the scan derives the keep flags itself; the merge does too
    if (n_old > 0) {
        if (fused_flags) {
            ScanFlagArgs A{}; A.a = S.seq[S.cur].p; A.drop = drop;
            hipLaunchKernelGGL(k_scan_flags<kScanKeep>, dim3(nblocks(n_old, kScanTile)), dim3(kScanFlagThreads), 0, ctx->stream, A, (int)n_old, V->slots.as<int>());
        } else {
            hipLaunchKernelGGL(k_keep_flags, dim3(nblocks(n_old, 256)), dim3(256), 0, ctx->stream, S.seq[S.cur].as<unsigned>(), n_old, drop, V->flags.as<int>());
            int rc = exclusive_scan(ctx, V, V->flags.as<int>(), n_old, V->slots.as<int>());
            if (rc != LILI_OK) return rc;
        }
    } else HIPCHK(hipMemsetAsync(V->slots.p, 0, 4, ctx->stream));
    if (n_old + n_new > 0)
        hipLaunchKernelGGL(k_merge, dim3(nblocks(n_old + n_new, 256)), dim3(256), 0, ctx->stream, S.key[S.cur].as<unsigned long long>(), S.pt[S.cur].as<float4>(), S.seq[S.cur].as<unsigned>(),
                           fused_flags ? (const int*)nullptr : (const int*)V->flags.as<int>(), drop, (const int*)V->slots.as<int>(), n_old, V->kf_key.as<unsigned long long>(), V->kf_pt.as<float4>(),
                           n_new, add ? add->seq : 0u, S.key[dst].as<unsigned long long>(), S.pt[dst].as<float4>(), S.seq[dst].as<unsigned>());
    HIPCHK(hipGetLastError());
    S.cur = dst; S.n = n_out;
    return LILI_OK;
}

int lili_localmap_commit(lili_ctx* ctx, int kind, float leaf, double max_sq_radius, int64_t* n_raw, int64_t* n_map) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK((kind == 0 || kind == 1) && leaf > 0, "localmap_commit: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    auto* V = vox_of(ctx);
    auto& ring = V->ring[kind];
    lili_detail::SortedRing& S = V->sorted[kind];
    size_t total = 0;
    for (auto* k : ring) total += (size_t)k->n;
    if (n_raw) *n_raw = (int64_t)total;
    unsigned* d_bad = reinterpret_cast<unsigned*>(ctx->misc.as<char>() + 256);      // (the bounding-box words of voxel_sort live in the first 24 bytes)
    // ---- what changed since the sorted ring was built: keyframes popped at the front, keyframes pushed at the back
    bool inc = ctx->localmap_incremental && S.valid && S.leaf == leaf && S.n > 0 && !ring.empty() && total < (1ull << 31);
    DropSeqs drop{}; long long n_drop = 0;
    std::vector<lili_detail::Keyframe*> add;
    if (inc) {
        size_t f = 0;
        while (f < S.members.size() && S.members[f].first != ring.front()->seq) f++;
        if (f == S.members.size() || f > 4) inc = false;
        else {
            for (size_t i = 0; i < f; i++) { drop.s[i] = S.members[i].first; n_drop += S.members[i].second; }
            drop.n = (int)f;
            size_t m = 0;
            while (f + m < S.members.size() && m < ring.size() && S.members[f + m].first == ring[m]->seq) m++;
            if (f + m != S.members.size()) inc = false;                    // the ring is not "old members, then new keyframes"
            for (size_t i = m; i < ring.size(); i++) add.push_back(ring[i]);
            if (add.size() > 2) inc = false;                               // several pending keyframes: the rebuild is cheaper than as many merge passes
        }
    }
    bool have_box = false;
    if (inc) {
        // ---- incremental step(s): the merge, then the centroid pass over the sorted ring
        { const int rl = lili_lazy_sources_clear_of(ctx, d_bad, kRankOff - 256 + (size_t)kVoxSmallMax * 4); if (rl != LILI_OK) return rl; }
        HIPCHK(hipMemsetAsync(d_bad, 0, kRankOff - 256 + (size_t)kVoxSmallMax * 4, ctx->stream));      // *bad ... the rank words of k_rank_count: one fill
        int rc = LILI_OK;
        if (add.empty()) rc = sorted_ring_step(ctx, V, S, drop, n_drop, nullptr, leaf, d_bad);
        for (size_t i = 0; i < add.size() && rc == LILI_OK; i++) {
            if (i > 0) HIPCHK(hipMemsetAsync(ctx->misc.as<char>() + kRankOff, 0, (size_t)kVoxSmallMax * 4, ctx->stream));      // (a second pending keyframe counts from zero again)
            rc = sorted_ring_step(ctx, V, S, i == 0 ? drop : DropSeqs{}, i == 0 ? n_drop : 0, add[i], leaf, d_bad);
        }
        if (rc != LILI_OK) { S.valid = false; return rc; }
        S.members.clear();
        for (auto* k : ring) S.members.push_back({k->seq, k->n});
        const long long n = S.n;
        V->n_out = 0;
        unsigned bad = 0;
        static_assert(2048 + kBoxBanks * 128 <= 8192, "box banks of the commit inside the first 8 KB of ctx->misc");
        unsigned box_banks[kBoxBanks * 32];
        if (n > 0) {
            HIPCHK(V->flags.ensure((size_t)n * 4)); HIPCHK(V->slots.ensure(((size_t)n + 1) * 4));
            HIPCHK(V->out.ensure((size_t)n * 16)); HIPCHK(V->out_cnt.ensure((size_t)n * 4));
            HIPCHK(V->head_pos.ensure(((size_t)n + 1) * 4));
            if (n <= kScanFlagsMax) {      // heads, their scan and the compaction of the heads' positions in one launch
                ScanFlagArgs A{}; A.a = S.key[S.cur].p; A.head_pos = V->head_pos.as<int>();
                hipLaunchKernelGGL(k_scan_flags<kScanHead64>, dim3(nblocks(n, kScanTile)), dim3(kScanFlagThreads), 0, ctx->stream, A, (int)n, V->slots.as<int>());
            } else {
                hipLaunchKernelGGL(k_vox_heads64, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, S.key[S.cur].as<unsigned long long>(), n, V->flags.as<int>());
                rc = exclusive_scan(ctx, V, V->flags.as<int>(), n, V->slots.as<int>());
                if (rc != LILI_OK) return rc;
                hipLaunchKernelGGL(k_vox_head_pos, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, V->flags.as<int>(), V->slots.as<int>(), n, V->head_pos.as<int>());
            }
            // one thread per voxel: the grid covers the upper bound (every point its own voxel), threads beyond the count on the device leave at once
            // the bounding box of the centroids falls out of the centroid pass and travels with their count: the index build below starts without a read-back of its own
            unsigned* d_box = reinterpret_cast<unsigned*>(ctx->misc.as<char>() + 2048);      // (kBoxBanks x 128 bytes, zeroed with *bad above)
            hipLaunchKernelGGL(k_vox_centroid64, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, S.key[S.cur].as<unsigned long long>(), S.pt[S.cur].as<float4>(), V->head_pos.as<int>(),
                               n <= kScanFlagsMax ? 1 : 0, (const int*)(V->slots.as<int>() + n), n, V->out.as<float4>(), V->out_cnt.as<int>(), d_box);
            HIPCHK(hipGetLastError());
            rc = lili_readback_add(ctx, &V->n_out, V->slots.as<int>() + n, sizeof(int));
            if (rc == LILI_OK) { rc = lili_readback_add(ctx, box_banks, d_box, sizeof(box_banks)); have_box = rc == LILI_OK; }
        }
        if (rc == LILI_OK) rc = lili_readback_add(ctx, &bad, d_bad, 4);
        { const int rb = lili_readback_finish(ctx); if (rc != LILI_OK) return rc; if (rb != LILI_OK) return rb; }
        if (have_box) {      // fold the banks; the minima travel inverted (k_vox_centroid64)
            unsigned w[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < kBoxBanks; b++) for (int k = 0; k < 6; k++) w[k] = std::max(w[k], box_banks[b * 32 + k]);
            for (int k = 0; k < 3; k++) { V->out_box[k] = ~w[k]; V->out_box[3 + k] = w[3 + k]; }
        }
        if (bad) { S.valid = false; inc = false; have_box = false; }      // a point beyond the absolute key range: the box-relative rebuild below handles it
        else {
            // The guards of the full rebuild (voxel_sort: "no finite point", PCL's int32 voxel-index overflow) apply to the same ring content whichever
            // way the map is produced: no centroid at all, or a box of centroids whose voxel count — with one voxel of slack per side and axis,
            // a centroid may round across its voxel's face — would not fit int32, hands the commit to the rebuild, which decides exactly and reports the
            // error a first commit would report.
            bool guard = total > 0 && V->n_out == 0;
            if (!guard && have_box) {
                auto dec = [](unsigned u) { unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &b, 4); return f; };
                const float inv_leaf = 1.0f / leaf;
                double cells = 1.0;
                for (int k = 0; k < 3; k++) cells *= (double)((long long)std::floor(dec(V->out_box[3 + k]) * inv_leaf) - (long long)std::floor(dec(V->out_box[k]) * inv_leaf) + 3);
                guard = !(cells <= 2147483647.0);
            }
            if (guard) { S.valid = false; inc = false; have_box = false; }
            else V->incremental_commits++;
        }
    }
    if (!inc) {
        // ---- full rebuild: concatenate, sort, reduce
        HIPCHK(V->concat.ensure(std::max<size_t>(total, 1) * 16));
        // *surf_local_map += *recent_surf_keyframes[i] (L:1479-1483): ONE gather launch over a table of (source, first output position) per keyframe —
        // fifty device-to-device copies of ~300 KB cost ~2 us each on the stream (0.12 ms per commit)
        std::vector<lili_detail::ConcatSeg>& segs = V->seg_host;      // members, not locals: hipMemcpyAsync may read them after this block
        std::vector<unsigned>& sseq = V->seg_seq_host;
        segs.clear(); sseq.clear();
        size_t off = 0;
        for (auto* k : ring) { if (k->n) { segs.push_back({k->pts.as<float4>(), (long long)off}); sseq.push_back(k->seq); } off += (size_t)k->n; }
        if (!segs.empty()) {
            HIPCHK(V->concat_tab.ensure(segs.size() * sizeof(lili_detail::ConcatSeg)));
            HIPCHK(hipMemcpyAsync(V->concat_tab.p, segs.data(), segs.size() * sizeof(lili_detail::ConcatSeg), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_concat, dim3(nblocks((int64_t)total, 256)), dim3(256), 0, ctx->stream, V->concat_tab.as<lili_detail::ConcatSeg>(), (int)segs.size(),
                               (long long)total, V->concat.as<float4>());
            HIPCHK(hipGetLastError());
        }
        int rc = voxel_filter_device(ctx, V, V->concat.as<float4>(), (int)total, leaf, true);   // ds_filter_*_map.filter (L:1488-1492); the sorted ring below needs the order
        if (rc != LILI_OK) return rc;
        V->full_commits++;
        // the sorted ring for the following steps: the order the sort has just produced, with absolute keys and the keyframe of every point
        S.valid = false;
        if (ctx->localmap_incremental && total > 0 && total < (1ull << 31) && !segs.empty()) {
            HIPCHK(S.key[0].ensure(total * 8)); HIPCHK(S.pt[0].ensure(total * 16)); HIPCHK(S.seq[0].ensure(total * 4));
            HIPCHK(V->seg_seq.ensure(sseq.size() * 4));
            HIPCHK(hipMemcpyAsync(V->seg_seq.p, sseq.data(), sseq.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_sorted_gather, dim3(nblocks((int64_t)total, 256)), dim3(256), 0, ctx->stream, V->concat.as<float4>(), V->vals_a.as<int>(), (int)total, 1.0f / leaf, 0u,
                               V->concat_tab.as<lili_detail::ConcatSeg>(), V->seg_seq.as<unsigned>(), (int)segs.size(), S.pt[0].as<float4>(), S.key[0].as<unsigned long long>(),
                               S.seq[0].as<unsigned>(), d_bad);
            HIPCHK(hipGetLastError());
            unsigned bad = 0;
            { int rb = lili_readback_add(ctx, &bad, d_bad, 4); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
            S.cur = 0; S.n = (long long)total; S.leaf = leaf; S.valid = bad == 0;
            S.members.clear();
            for (auto* k : ring) S.members.push_back({k->seq, k->n});
        }
    }
    if (n_map) *n_map = V->n_out;
    lili_cloud c{V->out.p, (size_t)V->n_out, 16, 12, LILI_MEM_DEVICE};
    // The super-row copy (9x the points, a layout that never changes a result) pays for maps that serve many large launches; a keyframe-ring map of
    // a few hundred thousand points is rebuilt per keyframe and serves ~30 launches of 1-3 k queries: 70 us of copy against ~1 us saved per launch
    // (measured: k_scatter9 45 + k_start9 12 + k_rowtot9 5 + scan 10 us on the 152 k-point ring map).  Option "localmap_super_rows" = 1 keeps it.
    const bool srows = ctx->super_rows;
    if (!ctx->localmap_super_rows && V->n_out < 400000 && !(ctx->focus_radius > 0)) ctx->super_rows = false;
    // setInputCloud (L:839-840).  After an incremental step the centroids are indexed where they lie (no ingestion copy) with the box that came with their count.
    const int rc_map = have_box && V->n_out > 0 ? lili_map_set_hinted(ctx, kind, &c, max_sq_radius, V->out_box, true) : lili_map_set(ctx, kind, &c, max_sq_radius);
    ctx->super_rows = srows;
    return rc_map;
}

// The cloud the last lili_localmap_commit (or lili_voxel_filter) produced: the down-sampled local map in map order (what the reference
// publishes as its local map and hands to setInputCloud).  Blocking.
int lili_localmap_get(lili_ctx* ctx, lili_feature_out* out) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(out, "localmap_get: null out");
    HIPCHK(hipSetDevice(ctx->device));
    auto* V = vox_of(ctx);
    out->count = (size_t)V->n_out;
    const size_t k = std::min(out->count, out->capacity);
    if (out->data && k) {
        const size_t stride = out->stride ? out->stride : 16;
        ARGCHK(stride >= 16, "localmap_get: stride must be >= 16");
        const hipMemcpyKind kind = out->mem == LILI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (stride == 16) HIPCHK(hipMemcpyAsync(out->data, V->out.p, k * 16, kind, ctx->stream));
        else HIPCHK(hipMemcpy2DAsync(out->data, stride, V->out.p, 16, 16, k, kind, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return LILI_OK;
}

// how the commits of this context were served so far (tests, tools): incremental steps on the sorted ring / full rebuilds
int lili_localmap_stats(lili_ctx* ctx, int32_t* incremental_commits, int32_t* full_commits) {
    if (!ctx) return LILI_E_ARG;
    auto* V = vox_of(ctx);
    if (incremental_commits) *incremental_commits = V->incremental_commits;
    if (full_commits) *full_commits = V->full_commits;
    return LILI_OK;
}

// how the VoxelGrid filters of more than 8192 points were served so far: with guessed key bits (no host round trip for the box) / guesses that did not hold
int lili_voxel_filter_stats(lili_ctx* ctx, int32_t* key_guesses, int32_t* key_guess_misses) {
    if (!ctx) return LILI_E_ARG;
    auto* V = vox_of(ctx);
    if (key_guesses) *key_guesses = V->key_guesses;
    if (key_guess_misses) *key_guess_misses = V->key_guess_misses;
    return LILI_OK;
}

}  // extern "C"

// ---- internal hooks of lili_pipeline.hip (declared in lili_ctx.h; not part of the ABI) ----
// pcl::VoxelGrid of a device float4 cloud; the centroids stay in the filter's own output buffer (valid until the next filter / commit on this context)
int lili_voxel_filter_dev(lili_ctx* ctx, const float4* d_pts, int n, float leaf, const float4** d_out, int* n_out) {
    auto* V = vox_of(ctx);
    const int rc = voxel_filter_device(ctx, V, d_pts, n, leaf);
    if (rc != LILI_OK) return rc;
    if (d_out) *d_out = V->out.as<float4>();
    if (n_out) *n_out = V->n_out;
    return LILI_OK;
}
// the same in two halves, into the filter's SECOND output buffer (so that it may be enqueued while the first still holds a local map that is being indexed):
// _enqueue never blocks; *pending = a read-back has joined the context's list and wants a lili_readback_finish before _complete
int lili_voxel_filter_dev_enqueue(lili_ctx* ctx, const float4* d_pts, int n, float leaf, bool box_zeroed, bool* pending) {
    auto* V = vox_of(ctx);
    const int rc = voxel_filter_enqueue(ctx, V, d_pts, n, leaf, false, true, box_zeroed);
    if (pending) *pending = V->pend.mode == kPendSmall || V->pend.mode == kPendGuess;
    return rc;
}
int lili_voxel_filter_dev_complete(lili_ctx* ctx, const float4** d_out, int* n_out) {
    auto* V = vox_of(ctx);
    const int rc = voxel_filter_complete(ctx, V);
    if (rc != LILI_OK) return rc;
    if (d_out) *d_out = V->qout.as<float4>();
    if (n_out) *n_out = V->qn_out;
    return LILI_OK;
}
// lili_localmap_push of a device float4 cloud at the pose held in a slot's DEVICE state (async: no host round trip for the pose)
int lili_localmap_push_dev(lili_ctx* ctx, int kind, const float4* d_pts, int n, const SlotState* d_state, int width) {
    return localmap_push_f4(ctx, kind, d_pts, n, nullptr, nullptr, d_state, width);
}
int lili_localmap_ring_size(lili_ctx* ctx, int kind) { return (int)vox_of(ctx)->ring[kind].size(); }

