// Scan-to-map matcher kernels for gfx950 (MI355X): map index build (K7), surf / edge association
// (K5 / K6: transform + exact 5-NN + plane / line fit + gates), linearisation (residual + analytic
// Jacobian + loss corrector + deterministic Gram reduction) and the on-device Gauss-Newton update.
//
// Reference behaviour replaced (L/ = LiLi-OM/, R/ = LiLi-OM-ROT/):
//   findCorrespondingSurfFeatures    L/src/BackendFusion.cpp:1601-1681, R/src/BackendFusion.cpp:1464-1520,
//                                    L/src/LidarOdometry.cpp:352-413
//   findCorrespondingCornerFeatures  L/src/BackendFusion.cpp:1531-1599, R/src/BackendFusion.cpp:1394-1462
//   LidarEdgeFactor / LidarPlaneNormFactor / LidarPlaneNormIncreFactor   L/include/factors/LidarKeyframeFactor.h:12-139
//   loss corrector + J^T J accumulation   L/src/MarginalizationFactor.cpp:3-29,44-70
// HBM-bound gather work: no MFMA (the J^T J contraction is N x 8 -> 8 x 8), f64 on the vector ALUs.
#include <type_traits>
#include "lili_kernels.h"
#include "lili_device_math.h"
#include "lili_s2m_dev.h"

namespace lili {

// ================================================================================================
// cloud ingestion: AoS points (stride 32 / 48 B ...) -> float4 (x, y, z, aux)
// ================================================================================================
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// `mm` (optional, 64 banks x 32 words, words 0..5 of a bank = ordered-uint min xyz / max xyz): the bounding box of the finite points is reduced in the same pass (the map
// index needs it before anything else; a separate k_bbox pass re-read the whole cloud) — one atomic set per block.
__global__ __launch_bounds__(256) void k_cloud_to_f4(const unsigned char* __restrict__ raw, int n, int stride, int aux_off, float4* __restrict__ out, unsigned* __restrict__ mm) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(raw + (size_t)i * stride);
        float4 v;
        v.x = p[0]; v.y = p[1]; v.z = p[2];
        v.w = aux_off >= 0 ? *reinterpret_cast<const float*>(raw + (size_t)i * stride + aux_off) : 0.f;
        out[i] = v;
        if (mm && isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) {
            mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
            mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
        }
    }
    if (!mm) return;
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
        // 64 banks of 128 bytes (the host folds them): 4096 blocks on six words of ONE line were ~50 us of same-address atomics
        unsigned* bank = mm + (size_t)(blockIdx.x & 63) * 32;
        if (a <= b) { atomicMin(&bank[k], f2ord(a)); atomicMax(&bank[3 + k], f2ord(b)); }
    }
}

// ================================================================================================
// K7 — map index build: bounding box, cell histogram, exclusive scan, scatter
// ================================================================================================
__global__ void k_bbox(const float4* __restrict__ pts, int n, unsigned* __restrict__ mm /*[6]: min xyz, max xyz (ordered-uint)*/) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    }
    // one atomic set per BLOCK (same-address atomics serialise at ~12 ns each on MI355X)
    __shared__ float smn[kBlock / 64][3], smx[kBlock / 64][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { smn[threadIdx.x >> 6][k] = mn[k]; smx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        int k = threadIdx.x;
        float a = smn[0][k], b = smx[0][k];
        for (int w = 1; w < kBlock / 64; w++) { a = fminf(a, smn[w][k]); b = fmaxf(b, smx[w][k]); }
        atomicMin(&mm[k], f2ord(a)); atomicMax(&mm[3 + k], f2ord(b));
    }
}

// the same with the number of points read on the device (the local map's centroid count is not on the host yet)
__global__ void k_bbox_dev(const float4* __restrict__ pts, const int* __restrict__ n_dev, int n_max, unsigned* __restrict__ mm /*[6]: min xyz, max xyz (ordered-uint)*/) {
    const int n = min(*n_dev, n_max);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    }
    // one atomic set per BLOCK (same-address atomics serialise at ~12 ns each on MI355X)
    __shared__ float smn[kBlock / 64][3], smx[kBlock / 64][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { smn[threadIdx.x >> 6][k] = mn[k]; smx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        int k = threadIdx.x;
        float a = smn[0][k], b = smx[0][k];
        for (int w = 1; w < kBlock / 64; w++) { a = fminf(a, smn[w][k]); b = fmaxf(b, smx[w][k]); }
        atomicMin(&mm[k], f2ord(a)); atomicMax(&mm[3 + k], f2ord(b));
    }
}


__device__ __forceinline__ float4 load_src(const SrcCloud& s, int i) {
    if (s.f4) { float4 v = reinterpret_cast<const float4*>(s.p)[i]; if (s.aux_off < 0) v.w = 0.f; return v; }
    const unsigned char* q = s.p + (size_t)i * s.stride;
    const float* f = reinterpret_cast<const float*>(q);
    float4 v;
    v.x = f[0]; v.y = f[1]; v.z = f[2];
    v.w = s.aux_off >= 0 ? *reinterpret_cast<const float*>(q + s.aux_off) : 0.f;
    return v;
}
// 64 banks x 32 words, all reduced with atomicMax from ZERO (one memset arms every scratch word of a build): words 0..2 = ~ordered-uint of min xyz, words 3..5 =
// ordered-uint of max xyz; one atomic set per workgroup; the host folds the banks.
__device__ __forceinline__ void bbox_to_banks(float mn[3], float mx[3], unsigned* __restrict__ mm, float (*smn)[3], float (*smx)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { smn[threadIdx.x >> 6][k] = mn[k]; smx[threadIdx.x >> 6][k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = smn[0][k], b = smx[0][k];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) { a = fminf(a, smn[w][k]); b = fmaxf(b, smx[w][k]); }
        unsigned* bank = mm + (size_t)(blockIdx.x & 63) * 32;
        if (a <= b) { atomicMax(&bank[k], ~f2ord(a)); atomicMax(&bank[3 + k], f2ord(b)); }
    }
}
// Bounding box of the finite points of a caller's cloud, read where it lies (the first build of a map; later builds of a similar cloud start from
// the previous build's box and check it afterwards: lili_map_set).
__global__ __launch_bounds__(kBlock) void k_bbox_src(SrcCloud src, int n, unsigned* __restrict__ mm) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = load_src(src, i);
        if (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) {
            mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
            mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
        }
    }
    __shared__ float smn[kBlock / 64][3], smx[kBlock / 64][3];
    bbox_to_banks(mn, mx, mm, smn, smx);
}

// One global atomic per point in the WHOLE build: the value it returns is the point's rank inside its cell, kept for the scatter pass,
// which therefore needs no second counter array, no second 108 MB memset and no atomics (round 1: an atomic here, whose result
// was dropped, and another one in k_scatter).  Round 4: the points are read from the caller's cloud where it lies (no float4 copy of the map: -160 MB
// per 5 M points) and only the rank is stored (the scatter pass recomputes the cell: -40 MB).
// Neighbouring lanes that fall into the same cell (maps come out of the voxel filter in voxel order: consecutive points are
// neighbours in x) share ONE atomic: run heads add the run length, the members take base + offset.  No loop, ~10 instructions;
// an unordered cloud degenerates to one atomic per point.  All 64 lanes of a wave must be active.
// `check` (a build whose box is a GUESS — the previous build's true box plus a margin, lili_map_set): every workgroup ORs a word into its bank (next to its rank sum:
// the same line, one more fire-and-forget atomic): bit 0 if a finite point lies outside the grid (its cell is a clamped one: the build is repeated with the measured box),
// bit 1 if a point lies within a quarter cell of the grid's faces (the cloud has grown into the margin: the next build measures again), bits 2..7 if a point lies within
// `touch` cells of the faces x-, x+, y-, y+, z-, z+ (a face nothing comes near: the guess was the box of some other cloud — too loose, the build is repeated as well).
__global__ __launch_bounds__(kBlock) void k_cell_count(SrcCloud src, int n, GridView g, int* __restrict__ cell_count, int* __restrict__ pt_rank,
                                                        unsigned long long* __restrict__ banks /*64 x 16 words: [0] rank sum, [1] box check*/, int check, float touch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1;
    unsigned want = 0;
    if (i < n) {
        const float4 p = load_src(src, i);
        c = 0;                                                              // non-finite: never selected (its distance is NaN), as cell_of
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            // cell_of's arithmetic (lili_s2m_dev.h), with the position in cells kept for the checks
            const double u[3] = {((double)p.x - g.ox) * g.inv_cell, ((double)p.y - g.oy) * g.inv_cell, ((double)p.z - g.oz) * g.inv_cell};
            const int dims[3] = {g.nx, g.ny, g.nz};
            int cc[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int raw = (int)floor(u[k]);
                cc[k] = min(max(raw, 0), dims[k] - 1);
                if (check) {
                    if (raw != cc[k]) want |= 3u;
                    if (u[k] < 0.25 || u[k] > (double)dims[k] - 0.25) want |= 2u;
                    if (u[k] < (double)touch) want |= 4u << (2 * k);
                    if (u[k] > (double)dims[k] - 1.0 - (double)touch) want |= 8u << (2 * k);
                }
            }
            c = (cc[2] * g.ny + cc[1]) * g.nx + cc[0];
        }
    }
    const int prev = __shfl_up(c, 1);
    const bool head = lane == 0 || c != prev;
    const unsigned long long hm = __ballot(head);
    const unsigned long long upto = (2ull << lane) - 1ull;                 // lanes <= mine (lane 63: all)
    const int hl = 63 - __clzll((long long)(hm & upto));                   // my run's head lane
    const unsigned long long above = hm & ~upto;
    const int len = (above ? __ffsll((long long)above) - 1 : 64) - lane;   // run length (meaningful on head lanes)
    int base = 0;
    if (head && c >= 0) base = atomicAdd(&cell_count[c], len);
    base = __shfl(base, hl);
    const int rank = base + (lane - hl);
    if (i < n) pt_rank[i] = rank;
    // density estimate for free: a point's rank is its position inside its cell, so sum(rank) = sum over cells of occ (occ - 1) / 2 and the
    // point-weighted mean cell occupancy is 1 + 2 sum(rank) / n (lili_map_set decides on the fine grid with it).  One atomic per block,
    // spread over 64 banks that the host adds up.
    if (banks) {
        unsigned long long r64 = i < n && c >= 0 ? (unsigned long long)rank : 0ull;
        for (int o = 32; o > 0; o >>= 1) r64 += __shfl_xor(r64, o);
        unsigned w = 0;
        if (check) {
#pragma unroll
            for (int b = 0; b < 8; b++) if (__ballot((want >> b) & 1u)) w |= 1u << b;
        }
        __shared__ unsigned long long wsum[kBlock / 64];
        __shared__ unsigned wchk[kBlock / 64];
        if (lane == 0) { wsum[threadIdx.x >> 6] = r64; wchk[threadIdx.x >> 6] = w; }
        __syncthreads();
        // 64 banks, 128 bytes apart (same-address atomics cost ~12 ns each: 19.5 k blocks on ONE word were 0.2 ms of a 5 M-point build)
        if (threadIdx.x == 0) {
            unsigned long long t = 0; unsigned f = 0;
            for (int k = 0; k < kBlock / 64; k++) { t += wsum[k]; f |= wchk[k]; }
            unsigned long long* bank = banks + (size_t)(blockIdx.x & 63) * 16;
            atomicAdd(bank, t);
            if (check) atomicOr(bank + 1, (unsigned long long)f);
        }
    }
}

// The same pass on the NARROW count table (round 4): four 8-bit counters per 32-bit word — a voxel-filtered map holds a handful of points per cell —, so the table the
// count pass clears, fills and the scan reads is a quarter of the 32-bit one (24 MB instead of 97 MB for the bench map's 24 M cells), and lanes whose cells share a WORD
// share one atomic (cells along x are neighbours in the table: the pipeline's voxel-ordered maps need one atomic per ~6 points instead of one per 1.6).  The atomic
// returns the word's four old counts; a lane's rank = the old count of its cell + the lanes of its word run before it that fall into the same cell (ballots, no loop).
// Ranks are bytes.  A cell that would exceed 255 points raises bit 8 of the check word: lili_map_set then repeats the build with the 32-bit table and keeps to it for
// this kind of map (dense maps: their coarse index is rebuilt once).
__global__ __launch_bounds__(kBlock) void k_cell_count_narrow(SrcCloud src, int n, GridView g, unsigned* __restrict__ count4, unsigned char* __restrict__ pt_rank,
                                                               unsigned long long* __restrict__ banks /*64 x 16 words: [0] rank sum, [1] check word*/, int check, float touch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int c = -1;
    unsigned want = 0;
    if (i < n) {
        const float4 p = load_src(src, i);
        c = 0;
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const double u[3] = {((double)p.x - g.ox) * g.inv_cell, ((double)p.y - g.oy) * g.inv_cell, ((double)p.z - g.oz) * g.inv_cell};
            const int dims[3] = {g.nx, g.ny, g.nz};
            int cc[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int raw = (int)floor(u[k]);
                cc[k] = min(max(raw, 0), dims[k] - 1);
                if (check) {
                    if (raw != cc[k]) want |= 3u;
                    if (u[k] < 0.25 || u[k] > (double)dims[k] - 0.25) want |= 2u;
                    if (u[k] < (double)touch) want |= 4u << (2 * k);
                    if (u[k] > (double)dims[k] - 1.0 - (double)touch) want |= 8u << (2 * k);
                }
            }
            c = (cc[2] * g.ny + cc[1]) * g.nx + cc[0];
        }
    }
    const int w = c >> 2;                                                  // (-1 for the lanes past the end)
    const int sub = c & 3;
    const int prevw = __shfl_up(w, 1);
    const bool head = lane == 0 || w != prevw;
    const unsigned long long hm = __ballot(head);
    const unsigned long long upto = (2ull << lane) - 1ull;                 // lanes <= mine (lane 63: all)
    const unsigned long long below = upto >> 1;                            // lanes < mine
    const int hl = 63 - __clzll((long long)(hm & upto));                   // my word run's head lane
    const unsigned long long above = hm & ~upto;
    const int nxt = above ? __ffsll((long long)above) - 1 : 64;            // first lane of the next run
    const unsigned long long run = (nxt == 64 ? ~0ull : ((1ull << nxt) - 1ull)) & ~((1ull << hl) - 1ull);
    const unsigned long long m0 = __ballot(c >= 0 && sub == 0), m1 = __ballot(c >= 0 && sub == 1), m2 = __ballot(c >= 0 && sub == 2), m3 = __ballot(c >= 0 && sub == 3);
    const unsigned add = (unsigned)__popcll(m0 & run) | ((unsigned)__popcll(m1 & run) << 8) | ((unsigned)__popcll(m2 & run) << 16) | ((unsigned)__popcll(m3 & run) << 24);
    unsigned old = 0;
    if (head && c >= 0) old = atomicAdd(&count4[w], add);
    old = __shfl(old, hl);
    const unsigned long long mine = sub == 0 ? m0 : sub == 1 ? m1 : sub == 2 ? m2 : m3;
    const unsigned before = (old >> (8 * sub)) & 0xffu;
    const int rank = (int)before + __popcll(mine & run & below);
    if (c >= 0 && before + ((add >> (8 * sub)) & 0xffu) > 255u) want |= 256u;          // the counter would wrap (and carry into its neighbour)
    if (i < n) pt_rank[i] = (unsigned char)rank;
    unsigned long long r64 = i < n && c >= 0 ? (unsigned long long)rank : 0ull;
    for (int o = 32; o > 0; o >>= 1) r64 += __shfl_xor(r64, o);
    unsigned f = 0;
#pragma unroll
    for (int b = 0; b < 9; b++) if (__ballot((want >> b) & 1u)) f |= 1u << b;
    __shared__ unsigned long long wsum[kBlock / 64];
    __shared__ unsigned wchk[kBlock / 64];
    if (lane == 0) { wsum[threadIdx.x >> 6] = r64; wchk[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0; unsigned ff = 0;
        for (int k = 0; k < kBlock / 64; k++) { t += wsum[k]; ff |= wchk[k]; }
        unsigned long long* bank = banks + (size_t)(blockIdx.x & 63) * 16;
        atomicAdd(bank, t);
        if (ff) atomicOr(bank + 1, (unsigned long long)ff);
    }
}

// exclusive scan of n ints, 3 kernels: per-block sums, scan of block sums (single block), apply.
constexpr int kScanItems = 8;                       // items per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 per block
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds /*[kBlock/64 + 1]*/, int& total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; w++) { int s = lds[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}
__global__ void k_scan_block_sums(const int* __restrict__ in, int64_t n, int* __restrict__ block_sums) {
    __shared__ int lds[kBlock / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) if (base + k < n) s += in[base + k];
    int tot; block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
__global__ void k_scan_sums(int* __restrict__ block_sums, int nb) {   // single block, in place, exclusive
    __shared__ int lds[kBlock / 64 + 1];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += kBlock) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int tot; int ex = block_exclusive_scan(v, lds, tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
}
__global__ void k_scan_apply(const int* in, int64_t n, const int* __restrict__ block_offs, int* out /*[n+1]; may be `in` itself: every thread reads its items before it writes them*/) {
    __shared__ int lds[kBlock / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int v[kScanItems]; int s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    int tot; int ex = block_exclusive_scan(s, lds, tot) + block_offs[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) out[n] = block_offs[blockIdx.x] + tot;
}

// In-place exclusive scan of n ints in ONE pass (decoupled look-back): the dense cell array (27 M cells = 108 MB for the 5 M-point
// map) is read once and written once, where the three-kernel scan above reads it twice and writes it once.  Tile b (16384 items) belongs
// to workgroup b — no ticket: 6.6 k same-address atomics cost ~12 ns each on MI355X, more than the whole scan (measured: 183 us with a
// ticket against 170 us for the three kernels).  Workgroups are dispatched in index order per XCD, so the lowest unfinished tile is
// always resident and a tile only ever waits for lower ones; should that ever not hold (HIP promises no dispatch order), the bounded spin
// gives up, raises the sticky word `err` and carries on with a wrong prefix instead of hanging the GPU: lili_map_set reads `err` back
// at the synchronisation it has anyway and then repeats the build with the three-kernel scan.  Every tile publishes {flag, value} as ONE 64-bit word (flag 1 = the tile's own sum, 2 = the inclusive
// prefix up to and including the tile), written and read with agent-scope atomics — the word is its own flag, no fence.  One wave
// per tile looks back 64 predecessors at a time.  `ws`: [0], [1] = error flag, [2 ...] one status word per tile — zeroed by the
// caller.  data[n] receives the total.
// Round 4: the items of a tile are taken ROW-WISE — a wave owns 4096 contiguous items as 16 rows of 256, lane l holding items 4 l .. 4 l + 3 of every row — so that
// every load and store instruction of a wave covers one contiguous kilobyte (256 bytes of the narrow table).  Before, a lane owned 64 consecutive items: 16-byte
// accesses 256 bytes apart, 64 partial lines per instruction (52 us for the 24 M-cell table; the row-wise pass is bound by the 107 MB it writes).
// NARROW: the input is the byte-per-cell count table of k_cell_count_narrow (`in8`), the output the 32-bit `data` — 5 bytes moved per cell instead of 8.
constexpr int kLbRows = 16;
constexpr int kLbTile = kBlock * kLbRows * 4;      // 16384 items per workgroup of 256
template <bool NARROW>
__global__ __launch_bounds__(kBlock) void k_scan_lookback_t(int* data, const unsigned char* __restrict__ in8, int64_t n, unsigned long long* __restrict__ ws, unsigned* __restrict__ err /*sticky: set if a look-back gave up*/) {
    __shared__ int wtot[kBlock / 64];
    __shared__ int s_prefix;
    const int tile = (int)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* status = ws + 2;
    const int64_t wbase = (int64_t)tile * kLbTile + (int64_t)wave * (kLbRows * 256) + lane * 4;      // this lane's items of row r: wbase + 256 r .. + 3
    const bool whole = (int64_t)(tile + 1) * kLbTile <= n;
    int v[kLbRows][4];
    if (NARROW) {            // the table is padded to whole tiles (and cleared up to there): no bounds on the loads
#pragma unroll
        for (int r = 0; r < kLbRows; r++) {
            const unsigned q = *reinterpret_cast<const unsigned*>(in8 + wbase + 256 * r);
            v[r][0] = (int)(q & 0xffu); v[r][1] = (int)((q >> 8) & 0xffu); v[r][2] = (int)((q >> 16) & 0xffu); v[r][3] = (int)(q >> 24);
        }
    } else if (whole) {
#pragma unroll
        for (int r = 0; r < kLbRows; r++) { const int4 q = *reinterpret_cast<const int4*>(data + wbase + 256 * r); v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w; }
    } else {
#pragma unroll
        for (int r = 0; r < kLbRows; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) { const int64_t i = wbase + 256 * r + k; v[r][k] = i < n ? data[i] : 0; }
    }
    // exclusive prefix of every item inside its wave's stretch: 4 items per lane serially, 64 lanes by a wave scan, the 16 rows by a running carry
    int ex[kLbRows];
    int carry = 0;
#pragma unroll
    for (int r = 0; r < kLbRows; r++) {
        const int s4 = (v[r][0] + v[r][1]) + (v[r][2] + v[r][3]);
        int inc = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        ex[r] = carry + inc - s4;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) wtot[wave] = carry;
    __syncthreads();
    int wofs = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { const int t = wtot[w]; if (w < wave) wofs += t; tot += t; }
    if (threadIdx.x == 0) {
        const unsigned long long w = ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned long long)(unsigned)tot;
        __hip_atomic_store(&status[tile], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile == 0) s_prefix = 0;
    }
    if (tile > 0 && threadIdx.x < 64) {          // one wave walks back until it meets an inclusive prefix
        int acc = 0;
        for (int hi = tile - 1; ; hi -= 64) {
            const int t = hi - lane;
            unsigned long long w = 2ull << 32;       // tiles before 0: "prefix 0"
            if (t >= 0) {
                int spins = 0;
                do {
                    w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > (1 << 21)) { ws[1] = 1ull; if (err) *err = 1u; w = 2ull << 32; }      // cannot happen while lower tiles run; never hang the GPU — lili_map_set reads `err` back and rebuilds with the three-kernel scan
                } while ((w >> 32) == 0ull);
            }
            const bool is_p = (w >> 32) == 2ull;
            const unsigned long long pm = __ballot(is_p);
            const int first_p = pm ? __ffsll((long long)pm) - 1 : 64;          // nearest predecessor with a prefix
            int val = lane <= first_p ? (int)(unsigned)(w & 0xffffffffull) : 0;
            for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o);
            acc += val;
            if (pm) break;
        }
        if (lane == 0) {
            s_prefix = acc;
            __hip_atomic_store(&status[tile], (2ull << 32) | (unsigned long long)(unsigned)(acc + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int base = s_prefix + wofs;
#pragma unroll
    for (int r = 0; r < kLbRows; r++) {
        int4 q;
        q.x = base + ex[r]; q.y = q.x + v[r][0]; q.z = q.y + v[r][1]; q.w = q.z + v[r][2];
        const int64_t i = wbase + 256 * r;
        if (whole) *reinterpret_cast<int4*>(data + i) = q;
        else {
            if (i < n) data[i] = q.x;
            if (i + 1 < n) data[i + 1] = q.y;
            if (i + 2 < n) data[i + 2] = q.z;
            if (i + 3 < n) data[i + 3] = q.w;
        }
    }
    if (tile == (int)gridDim.x - 1 && threadIdx.x == 0) data[n] = s_prefix + tot;      // the total (items past n count as zero)
}
template __global__ void k_scan_lookback_t<false>(int*, const unsigned char*, int64_t, unsigned long long*, unsigned*);
template __global__ void k_scan_lookback_t<true>(int*, const unsigned char*, int64_t, unsigned long long*, unsigned*);

// Population of every super-row of the box (its nine source rows, box columns only); the scan of these is the row's first position.
__global__ void k_rowtot9(const int* __restrict__ cell_start, GridView g, int* __restrict__ rowtot) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= g.bny * g.bnz) return;
    const int ys = g.by0 + r % g.bny, zs = g.bz0 + r / g.bny;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int y = ys + k % 3 - 1, z = zs + k / 3 - 1;
        if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
        const int* cs = cell_start + ((size_t)z * g.ny + y) * g.nx + g.bx0;
        s += cs[g.bnx] - cs[0];
    }
    rowtot[r] = s;
}
// start9[(x, y', z')] = n + first position of the super-row + points of its nine source rows in the box columns before x.
// One wave per (64 x cells, kS9Rows super-rows, plane z'): (kS9Rows + 2) x 3 row loads, shared by the rows of the wave.
constexpr int kS9Rows = 8;
__global__ __launch_bounds__(256) void k_start9(const int* __restrict__ cell_start, GridView g, const int* __restrict__ rowbase, int* __restrict__ start9) {
    const int lane = threadIdx.x & 63;
    const int x = g.bx0 + blockIdx.x * 64 + lane;
    const int zs = g.bz0 + (int)blockIdx.z;
    const int y0 = g.by0 + ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * kS9Rows;   // first y' of this wave
    if (y0 >= g.by0 + g.bny) return;
    const int xc = min(x, g.bx0 + g.bnx - 1);
    int c3[kS9Rows + 2];                                                            // per source row y: box-relative prefix at x, summed over the three planes
    // Round 6: most stretches of a sparse index (the fine index of a dense map: 98 %) have no point in any of their (kS9Rows + 2) x 3 source rows — then every lane
    // of the wave holds the SAME prefixes, those at the stretch's first cell.  Lane r * 3 + p tests source row r of plane p (two words) and holds that prefix; the
    // sums come from shuffles: two loads per lane for 30 lanes instead of 60 per lane (329 -> ~90 us for the 84 M super cells of configs[2] variant B).
    const int xa = g.bx0 + (int)blockIdx.x * 64, xb = min(xa + 64, g.bx0 + g.bnx);      // the stretch: cells [xa, xb)
    int pfx = 0; bool occ = false;
    if (lane < (kS9Rows + 2) * 3) {
        const int r = lane / 3, y = y0 - 1 + r, z = zs + lane % 3 - 1;
        if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
            const int* cs = cell_start + ((size_t)z * g.ny + y) * g.nx;
            const int a = cs[xa], b = cs[xb];
            pfx = a - cs[g.bx0]; occ = b != a;
        }
    }
    if (!__any(occ)) {
#pragma unroll
        for (int r = 0; r < kS9Rows + 2; r++) c3[r] = __shfl(pfx, 3 * r) + __shfl(pfx, 3 * r + 1) + __shfl(pfx, 3 * r + 2);
    } else {
#pragma unroll
    for (int r = 0; r < kS9Rows + 2; r++) {
        const int y = y0 - 1 + r;
        int s = 0;
#pragma unroll
        for (int dz = -1; dz <= 1; dz++) {
            const int z = zs + dz;
            const bool in = y >= 0 && y < g.ny && z >= 0 && z < g.nz;
            const int* cs = cell_start + ((size_t)(in ? z : 0) * g.ny + (in ? y : 0)) * g.nx;
            const int a = cs[g.bx0], b = cs[xc];
            s += in ? b - a : 0;
        }
        c3[r] = s;
    }
    }
    if (x >= g.bx0 + g.bnx) return;
#pragma unroll
    for (int k = 0; k < kS9Rows; k++) {
        const int ys = y0 + k;
        if (ys < g.by0 + g.bny) start9[srow_index(g, x, ys, zs)] = g.n_points + rowbase[(zs - g.bz0) * g.bny + (ys - g.by0)] + c3[k] + c3[k + 1] + c3[k + 2];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        start9[(size_t)g.bnx * g.bny * g.bnz] = g.n_points + rowbase[g.bny * g.bnz];          // end of the array
}

template <typename RankT>
__global__ __launch_bounds__(kBlock) void k_scatter_t(SrcCloud src, int n, GridView g, const RankT* __restrict__ pt_rank, const int* __restrict__ cell_start,
                                                      float4* __restrict__ sorted, float* __restrict__ aux_sorted) {
    // XCD-aware order (round 6): workgroup ids go round-robin over the eight XCDs, each with its own L2.  A map that arrives in voxel order puts the points of one cell row
    // into ~2.7 runs thousands of points apart: with consecutive workgroups on different XCDs the partial writes of one destination line came from different L2s
    // and reached HBM as three partial writes (269 MB for 80 MB of output).  Here XCD x takes the x-th EIGHTH of the input — a slab of the map — so that the runs that share
    // destination lines meet in one L2 before the line is evicted.  (The grid covers 8 x ceil(blocks / 8) workgroups; the surplus ones leave.)
    const int nb = (n + (int)blockDim.x - 1) / (int)blockDim.x, per = (nb + 7) / 8;
    const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    int i = b * blockDim.x + threadIdx.x;
    if (b >= nb || i >= n) return;
    const int rank = (int)pt_rank[i];
    float4 p = load_src(src, i);
    const int pos = cell_start[cell_of(p, g)] + rank;
    if (aux_sorted) aux_sorted[pos] = p.w;
    p.w = __int_as_float(i);
    sorted[pos] = p;
}
template __global__ void k_scatter_t<int>(SrcCloud, int, GridView, const int*, const int*, float4*, float*);
template __global__ void k_scatter_t<unsigned char>(SrcCloud, int, GridView, const unsigned char*, const int*, float4*, float*);

// Super-row copy, by DESTINATION: a stretch = 64 consecutive super cells (x0..x0+63, y', z') — a contiguous piece of the super-row array — is filled by ONE wave
// from its nine source rows, each a contiguous run of the base array.  The nine runs are walked as ONE sequence, 64 points per trip; a point finds its super cell
// through its own x cell and its place through a 9 x 64 table (LDS) of "destination minus source" per (source row, cell).
// The 16 waves of a workgroup take a 4 x 4 tile of (y', z') at one x block, whose 36 source rows they share through the L2 of the XCD the workgroup runs on.
// Round 6: a workgroup takes `spw` consecutive tiles (x block fastest); lane j < spw of wave w tests the stretch at tile position w of the workgroup's j-th tile for
// emptiness (two words of start9) and the wave copies the non-empty ones one after the other.  spw = 1 is the one-stretch-per-wave kernel of rounds 3-5.  On an index whose
// cells are mostly empty (the fine index of a dense map: 1.39 M stretches, 18 % of them populated) a wave per stretch spent its time launching waves that load two words
// and leave: 639 us for 5 M points.
__global__ __launch_bounds__(1024) void k_scatter9(const int* __restrict__ cell_start, GridView g, const int* __restrict__ start9,
                                                   float4* __restrict__ sorted, float* __restrict__ aux_sorted, int spw) {
    __shared__ int delta[16][9][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nxb = (g.bnx + 63) / 64, nty = (g.bny + 3) / 4, ntz = (g.bnz + 3) / 4;
    const long long n_tiles = (long long)nxb * nty * ntz;
    const long long tile0 = (long long)blockIdx.x * spw;
    auto stretch_of = [&](long long tile, int& x0, int& ys, int& zs) -> bool {
        const int xb = (int)(tile % nxb); const long long r = tile / nxb;
        ys = g.by0 + (int)(r % nty) * 4 + (w & 3); zs = g.bz0 + (int)(r / nty) * 4 + (w >> 2); x0 = g.bx0 + xb * 64;
        return tile < n_tiles && ys < g.by0 + g.bny && zs < g.bz0 + g.bnz;
    };
    bool ne = false;
    if (lane < spw) {
        int x0, ys, zs;
        if (stretch_of(tile0 + lane, x0, ys, zs)) {
            ne = true;
            if (spw > 1) {      // (one stretch per wave: its emptiness shows in the loads the copy starts with — no dependent round trip in front of them)
                const int xn = min(64, g.bx0 + g.bnx - x0);
                const int* s9 = start9 + srow_index(g, x0, ys, zs);
                ne = s9[xn] != s9[0];
            }
        }
    }
    unsigned long long todo = __ballot(ne);
    while (todo) {
        int x0, ys, zs;
        stretch_of(tile0 + __builtin_ctzll(todo), x0, ys, zs);
        todo &= todo - 1ull;
        const int xn = min(64, g.bx0 + g.bnx - x0);            // cells of this stretch
        const int* s9 = start9 + srow_index(g, x0, ys, zs);
        const int lc = min(lane, xn - 1);
        int dnext = s9[lc];                                    // where the next source's points of cell x0+lane go
        const int d_end = s9[xn];
        const int D0 = __shfl(dnext, 0);
        const int len = d_end - D0;
        if (len == 0) continue;                                // (spw = 1: nothing lives here)
        int off[9], pre[10];
        pre[0] = 0;
        // all 27 range words of the nine source rows are requested BEFORE the first is used (round 4: 70.9 -> 64.4 us per focused copy)
        int ra[9], rbn[9], rend[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int y = ys + k % 3 - 1, z = zs + k / 3 - 1;
            const bool in = y >= 0 && y < g.ny && z >= 0 && z < g.nz;
            const int* cs = cell_start + ((size_t)(in ? z : 0) * g.ny + (in ? y : 0)) * g.nx + x0;
            ra[k] = cs[lc]; rbn[k] = cs[lc + 1]; rend[k] = cs[xn];
        }
        __builtin_amdgcn_wave_barrier();                       // the table of the stretch before has been read
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int y = ys + k % 3 - 1, z = zs + k / 3 - 1;
            const bool in = y >= 0 && y < g.ny && z >= 0 && z < g.nz;
            const int a = ra[k], b = rbn[k], e = rend[k];
            delta[w][k][lane] = dnext - a;
            dnext += in ? b - a : 0;
            const int rb = in ? __shfl(a, 0) : 0, re = in ? e : 0;
            off[k] = rb - pre[k];                              // position in the base array = position in the merged sequence + off[k]
            pre[k + 1] = pre[k] + (re - rb);
        }
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < len; t0 += 64) {                 // pre[9] == len   (four trips in flight were measured: no change, 64.4 vs 63.4 us)
            const int t = t0 + lane;
            const bool live = t < len;
            int k = 0, o = off[0];
#pragma unroll
            for (int i = 1; i < 9; i++) { const bool ge = t >= pre[i]; k += ge ? 1 : 0; o = ge ? off[i] : o; }
            const int j = live ? t + o : 0;
            const float4 p = sorted[j];
            int xc = 0;
            if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) xc = min(max(cell_coord(p.x, g.ox, g.inv_cell), 0), g.nx - 1);   // as cell_of
            const int l = min(max(xc - x0, 0), 63);
            const int d = j + delta[w][k][l];
            if (live) {
                sorted[d] = p;
                if (aux_sorted) aux_sorted[d] = aux_sorted[j];
            }
        }
    }
}

// A kind without records counts 0.  `out` (optional): a caller-owned int[2] that receives the same totals (multi-GPU
// callers all-reduce it in place).
// `v.seq != 0`: the totals are all-reduced over the ranks inside this launch (lili_p2p_dev.h) — one launch instead of count kernel +
// collective: state / out then hold the GLOBAL counts, identical on every rank.
__global__ __launch_bounds__(kBlock) void k_sum_counts(const int* __restrict__ bc_surf, int nb_surf, const int* __restrict__ bc_edge, int nb_edge,
                                                      SlotState* __restrict__ state, int* __restrict__ out, P2PView v) {
    const unsigned long long was_dead = p2p_dead_word(v);      // requested first: the round trip hides behind the count loads
    int t0 = bc_surf ? sum_block_counts(bc_surf, nb_surf) : 0;
    __syncthreads();
    int t1 = bc_edge ? sum_block_counts(bc_edge, nb_edge) : 0;
    if (threadIdx.x >= 64) return;
    if (v.seq) {
        unsigned long long s0, s1;
        const int mine = threadIdx.x == 0 ? t0 : t1;
        if (!p2p_exchange_wave<false>(v, 2, (unsigned long long)(unsigned)mine, 0ull, s0, s1, was_dead)) { if (threadIdx.x == 0) state->gn_status = 2; return; }
        t0 = (int)(unsigned)__shfl((int)(unsigned)s0, 0); t1 = (int)(unsigned)__shfl((int)(unsigned)s0, 1);
    }
    if (threadIdx.x == 0) {
        state->n_res[0] = t0; state->n_res[1] = t1;
        if (out) { out[0] = t0; out[1] = t1; }
    }
}



template <int BS, class TAB>
__device__ __forceinline__ void assoc_surf_body(
        const float4* __restrict__ queries, int n_q, const GridView& g, const PoseArg& pa, const MatchParams& P,
        float4* __restrict__ rec_nd, double* __restrict__ rec_score, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int bid, TAB& tab, LaneRec* rec_out = nullptr) {
#ifdef LILI_PHASE_PROBE
    const long long t_begin = (P.debug & 4096) ? (long long)__builtin_amdgcn_s_memrealtime() : 0ll;   // profiling build only (tools/assoc_blocks.py)
#endif
    const int i0 = bid * BS;
    const bool live = i0 + (int)threadIdx.x < n_q;
    const int i = live ? i0 + (int)threadIdx.x : 0;
    float4 ql = queries[i];     // requested before the (dependent, scalar) pose loads: the two latencies overlap
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;   // transformPoint, L:695-711
    float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    Top5 nn; nn.aux = 0; nn.have = false;
#ifdef LILI_PHASE_PROBE
    PhaseProbe probe;
#pragma unroll
    for (int k = 0; k < 8; k++) probe.t[k] = 0;
    PhaseProbe* const pp = &probe;
#else
    PhaseProbe* const pp = nullptr;
#endif
    PHASE_STAMP(pp, 1, px);                                                 // query loaded and moved into the map frame
    if (live) knn5_grid(g, tab, px, py, pz, gate_bound(P.kd_max_radius), nn, P.debug, pp);
    bool ok = false;
    if (live) {
        store_debug_nn(g, nn, i, dbg_idx, dbg_d2);
        float4 rn; double score;
        ok = surf_fit(g, P, nn, ql, px, py, pz, rn, score);
        PHASE_STAMP(pp, 6, rn.w);                                           // plane fitted, gates evaluated
        rec_nd[i] = rn;
        rec_score[i] = score;
        valid[i] = ok ? 1 : 0;
        if (rec_out) { rec_out->ql = ql; rec_out->r0 = rn; rec_out->score = score; }
    }
    if (rec_out) rec_out->ok = ok;
    store_block_count<BS>(ok, block_counts, bid);
#ifdef LILI_PHASE_PROBE
    if ((P.debug & 4096) && dbg_d2 && threadIdx.x == 0) {   // phase stamps as ticks since the block began, over the d2 debug rows of the block's first two queries
        for (int k = 1; k <= 6; k++) dbg_d2[(size_t)i0 * 5 + k] = probe.t[k] ? (float)(probe.t[k] - t_begin) : -1.0f;
    }
    if ((P.debug & 4096) && dbg_idx && threadIdx.x == 0) {   // per-workgroup begin / end ticks (100 MHz) and hardware id into the debug rows of the block's first query
        long long* o = (long long*)(dbg_idx + (size_t)i0 * 5);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = t_begin; o[1] = (long long)__builtin_amdgcn_s_memrealtime();
        dbg_idx[(size_t)i0 * 5 + 4] = (int)((hw & 0xffffu) | (xcc << 16));
    }
    if ((P.debug & 4096) && dbg_idx && live && threadIdx.x != 0) dbg_idx[(size_t)i * 5] = nn.aux;
#endif
}
template <int BS>
__global__ __launch_bounds__(BS) void k_associate_surf(
        const float4* __restrict__ queries, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_nd, double* __restrict__ rec_score, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts) {
    __shared__ RowTabT<BS> tab;
    assoc_surf_body<BS>(queries, n_q, g, pa, P, rec_nd, rec_score, valid, dbg_idx, dbg_d2, block_counts, (int)blockIdx.x, tab);
}

template <int BS, class TAB>
__device__ __forceinline__ void assoc_edge_body(
        const float4* __restrict__ queries, int n_q, const GridView& g, const PoseArg& pa, const MatchParams& P,
        float4* __restrict__ rec_a, float4* __restrict__ rec_b, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int bid, TAB& tab, LaneRec* rec_out = nullptr) {
    const int i0 = bid * BS;
    const bool live = i0 + (int)threadIdx.x < n_q;
    const int i = live ? i0 + (int)threadIdx.x : 0;
    float4 ql = queries[i];     // requested before the (dependent, scalar) pose loads: the two latencies overlap
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
    float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    Top5 nn; nn.have = false;
    if (live) knn5_grid(g, tab, px, py, pz, gate_bound(P.edge_gate), nn);
    bool ok = false;
    if (live) {
        store_debug_nn(g, nn, i, dbg_idx, dbg_d2);
        float4 ra, rb;
        ok = edge_fit(g, P, nn, px, py, pz, ra, rb);
        rec_a[i] = ra; rec_b[i] = rb; valid[i] = ok ? 1 : 0;
        if (rec_out) { rec_out->ql = ql; rec_out->r0 = ra; rec_out->r1 = rb; }
    }
    if (rec_out) rec_out->ok = ok;
    store_block_count<BS>(ok, block_counts, bid);
}
template <int BS>
__global__ __launch_bounds__(BS) void k_associate_edge(
        const float4* __restrict__ queries, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_a, float4* __restrict__ rec_b, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts) {
    __shared__ RowTabT<BS> tab;
    assoc_edge_body<BS>(queries, n_q, g, pa, P, rec_a, rec_b, valid, dbg_idx, dbg_d2, block_counts, (int)blockIdx.x, tab);
}

// Both kinds of one keyframe in ONE launch (the reference back-end associates corners and planes of a keyframe back to back,
// L/src/BackendFusion.cpp:935-936): workgroups [0, E.nb) take the (few) edge queries, the rest the surf queries — one wave per
// workgroup, direct search path.  Saves a kernel boundary and the fill / drain of a second grid per outer iteration.
__global__ __launch_bounds__(kAssocBlock) void k_associate_both(AssocArgs S, AssocArgs E, PoseArg pa, MatchParams P) {
    __shared__ RowTabT<kAssocBlock> tab;
    const int b = (int)blockIdx.x;
    if (b < E.nb) assoc_edge_body<kAssocBlock>(E.queries, E.n_q, E.g, pa, P, E.rec0, reinterpret_cast<float4*>(E.rec1), E.valid, E.dbg_idx, E.dbg_d2, E.block_counts, b, tab);
    else assoc_surf_body<kAssocBlock>(S.queries, S.n_q, S.g, pa, P, S.rec0, reinterpret_cast<double*>(S.rec1), S.valid, S.dbg_idx, S.dbg_d2, S.block_counts, b - E.nb, tab);
}
template __global__ void k_associate_surf<kAssocBlock>(const float4*, int, GridView, PoseArg, MatchParams, float4*, double*, unsigned char*, int*, float*, int*);
template __global__ void k_associate_edge<kAssocBlock>(const float4*, int, GridView, PoseArg, MatchParams, float4*, float4*, unsigned char*, int*, float*, int*);


// ---- fused tail: the whole inner iteration (linearise + reduce + solve + pose update) is ONE launch.  The block with the highest
// index is the REDUCER: after its own tile it sweeps the granules of all block partials until every one carries this launch's key
// (GramAcc::finish), adds them in the fixed order of reduce_partials_block — so the record does not depend on timing — and applies
// the Gauss-Newton update.  The other blocks just store and leave: no ticket, no fence, no drained-store wait
// (round 1's __threadfence pair, and a first round-2 version with write-through partials + sharded tickets, both cost more than
// the kernel boundary they saved: 46.1 / 39.9 vs 43.0 / 38.6 us per iteration — the chain store -> ack -> atomic -> atomic -> load is
// four memory round trips; the granule sweep is one).  key = launch_key(state->epoch); the reducer advances the epoch.
template <bool XCHG>
__device__ __forceinline__ void reduce_partials_block(const double* part_surf, int nb_surf, const double* part_edge, int nb_edge,
                                                      double* __restrict__ out, SlotState* __restrict__ state, int do_gn, unsigned long long key, const P2PView& xv,
                                                      unsigned long long pub_key = 0ull, double* pub = nullptr, SlotState* __restrict__ mirror = nullptr);   // defined below
__device__ __forceinline__ void fused_tail(const FuseTail& fz, unsigned long long key) {
    if (fz.mode != 1 && fz.mode != 2) return;      // 0: plain partials for k_reduce_partials; 3: publish only (the per-kind launches of merge_kinds = 0: the second launch reduces)
    if (blockIdx.x != gridDim.x - 1) return;
    P2PView none{};       // the fused tail is the single-GPU structure: no exchange (seq = 0)
    reduce_partials_block<false>(fz.part_surf, fz.nb_surf, fz.part_edge, fz.nb_edge, fz.out, fz.state, (fz.mode == 2 ? 1 : 0) | (fz.debug & 256), key, none);
}


// One launch for the kinds present: blocks [0, S.nb) linearise the surf records, blocks [S.nb, S.nb + E.nb) the edge records
// (either count may be 0); with fz.mode != 0 the last block to finish reduces all partials and (mode 2) applies the GN update.
__global__ __launch_bounds__(kLinBlock) void k_linearize(LinArgs S, LinArgs E, PoseArg pa, MatchParams P, const SlotState* __restrict__ state,
                                                         const int* __restrict__ n_global, FuseTail fz) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int bid = (int)blockIdx.x;
    const unsigned long long key = fz.mode ? launch_key(state->epoch) : 0ull;
    if (fz.mode && (fz.debug & 256) && bid == (int)gridDim.x - 1 && threadIdx.x == 0) fz.state->tprof[12] = (long long)__builtin_amdgcn_s_memrealtime();
    if (bid < S.nb) lin_surf_body(S, bid, pa, P, state, n_global, lds, key);
    else lin_edge_body(E, bid - S.nb, pa, P, state, n_global, lds, key);
    fused_tail(fz, key);
}

// All slots of a sliding window in one launch: block b belongs to the last slot whose first_block <= b (uniform per workgroup: the slot's
// arguments are scalar loads from the kernel-argument segment) and is that slot's block b - first_block of k_linearize.
__global__ __launch_bounds__(kLinBlock) void k_linearize_window(WinLinArgs W, MatchParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kWindowMaxSlots; k++) if (k < W.n && bid >= W.s[k].first_block) i = k;
    const WinLinSlot& ws = W.s[i];
    const int b = bid - ws.first_block;
    if (b < ws.S.nb) lin_surf_body(ws.S, b, ws.pa, P, ws.state, ws.n_global, lds, 0ull);
    else lin_edge_body(ws.E, b - ws.S.nb, ws.pa, P, ws.state, ws.n_global, lds, 0ull);
}

// ================================================================================================
// Final reduction of block partials (fixed order) -> 72-double record, optionally followed by the
// Gauss-Newton update in the same launch (single-GPU path; multi-GPU callers all-reduce in between).
// out: [0..63] full symmetric 8x8 Gram (row-major), [64] cost, [65] n_surf, [66] n_edge.
//
// GN step (device mirror of ceres::QuaternionParameterization):
//   P = blockdiag(I3, plusJacobian(q) 4x3); H = P^T G77 P, g = P^T G7r; solve H d = -g (Cholesky);
//   t += d[0:3]; q = [cos|dq|, sin|dq|/|dq| dq] (x) q
// ================================================================================================
// Association that LINEARISES ON THE FLY (VERDICT r1 #2 i): for the flavours whose residual weight does not depend on the correspondence
// count of the whole scan (Livox back end, front end: scale_*_num == 0 — the ROT back end divides by N, which exists only after the last
// wave), a lane that has just fitted its plane / line holds everything a linearisation lane would load again: query, record, body pose.
// It computes the same row (surf_lin_row / edge_lin_row on the ROUNDED record values), the wave reduces its 64 rows on the f64 MFMA
// exactly like a linearisation wave, and the per-wave partial goes where k_reduce_partials expects block partials.  The records are
// still stored (API, marginalisation feed).  One launch less per iteration, no second pass over the records, no count sum.
// Blocks [0, E.nb) take edge queries, the rest surf (either count may be 0).  The Gram staging rows reuse the row table's LDS.
// BS = 64 (one partial per wave; scans up to ~50 k queries, where the reducer takes them in one round of loads) or 256 (one partial per
// four waves, for larger scans).
template <int BS>
__global__ __launch_bounds__(BS) void k_associate_lin(AssocArgs S, AssocArgs E, PoseArg pa, MatchParams P, double* __restrict__ part_surf, double* __restrict__ part_edge) {
    __shared__ __attribute__((aligned(16))) RowTabT<BS> tab;
    static_assert(sizeof(RowTabT<BS>) >= (size_t)BS * kRow * sizeof(double), "Gram staging rows must fit the row table");
    const int b = (int)blockIdx.x;
    const bool edge = b < E.nb;
    LaneRec rec{};
    if (edge) assoc_edge_body<BS>(E.queries, E.n_q, E.g, pa, P, E.rec0, reinterpret_cast<float4*>(E.rec1), E.valid, E.dbg_idx, E.dbg_d2, E.block_counts, b, tab, &rec);
    else assoc_surf_body<BS>(S.queries, S.n_q, S.g, pa, P, S.rec0, reinterpret_cast<double*>(S.rec1), S.valid, S.dbg_idx, S.dbg_d2, S.block_counts, b - E.nb, tab, &rec);
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double cost = 0.0;
    if (rec.ok) {
        if (edge) cost = edge_lin_row(P, Q, T, rec.ql, rec.r0, rec.r1, (double)rec.r0.w, Jr);
        else cost = surf_lin_row(P, Q, T, dq{P.q_lb_inv_jet[0], P.q_lb_inv_jet[1], P.q_lb_inv_jet[2], P.q_lb_inv_jet[3]}, rec.ql, rec.r0, rec.score, Jr);
    }
    __syncthreads();                                  // (one wave) the search is over: its table becomes the staging area
    double* lds = reinterpret_cast<double*>(&tab);
    GramAcc ga; ga.init();
    ga.add_rows(Jr, cost, rec.ok, lds);
    ga.finish(lds, edge ? part_edge + (size_t)b * kPartialStride : part_surf + (size_t)(b - E.nb) * kPartialStride);
}
template __global__ void k_associate_lin<kAssocBlock>(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*);
template __global__ void k_associate_lin<kBlock>(AssocArgs, AssocArgs, PoseArg, MatchParams, double*, double*);

// xq: the quaternion of state->pose, loaded by the caller at kernel start (its latency hides behind the partial loads).
// Must be called by exactly ONE wave (lanes 0..63 of it).
// pub_key != 0: the resulting pose (the unchanged one if the step was rejected) is also PUBLISHED as seven keyed granules, kPubReplicas copies (lane l writes copy l), for
// an association launch that is already running behind this kernel (wait_published_pose, option "overlap_gn").
// `mirror` (optional): a page-locked copy of the slot's state as the device sees it — the lane that updates the pose writes pose and status there as well, so that a caller
// whose next step is "read the pose" synchronises without a copy launch in between (lili_pipeline.hip; gn_status there stays at the host's sentinel if this function is
// not reached).
__device__ void gn_update_block(const double* gram /*LDS or global, 64+*/, SlotState* __restrict__ state, const double xq[4], unsigned long long pub_key = 0ull, double* pub = nullptr,
                                SlotState* __restrict__ mirror = nullptr) {
    __shared__ double H[6][6];
    __shared__ double gvec[6];
    int tid = threadIdx.x & 63;
    const double x0 = xq[0], x1 = xq[1], x2 = xq[2], x3 = xq[3];
    // plus-Jacobian Jq (4x3) of ceres::QuaternionParameterization, rows [-x1 -x2 -x3; x0 x3 -x2; -x3 x0 x1; x2 -x1 x0]; every lane
    // builds the column(s) it needs in registers.  H = P^T G77 P and g = -P^T G7r with P = blockdiag(I3, Jq), evaluated as
    // M = G P (4-term sums, left to right) and H = P^T M exactly like round 1's three LDS-staged steps — one step now.
    auto jcol = [&](int c, double o[4]) {
        o[0] = c == 0 ? -x1 : c == 1 ? -x2 : -x3;
        o[1] = c == 0 ? x0 : c == 1 ? x3 : -x2;
        o[2] = c == 0 ? -x3 : c == 1 ? x0 : x1;
        o[3] = c == 0 ? x2 : c == 1 ? -x1 : x0;
    };
    if (tid < 42) {
        const int a = tid < 36 ? tid / 6 : tid - 36, b = tid < 36 ? tid % 6 : 7;   // b == 7: the J^T r column
        double jb[4] = {0, 0, 0, 0}, ja[4] = {0, 0, 0, 0};
        if (b >= 3 && b < 6) jcol(b - 3, jb);
        if (a >= 3) jcol(a - 3, ja);
        auto Mrow = [&](int i) -> double {      // (G P)[i][b];  for b == 7 the plain column G[i][7]
            if (b < 3 || b == 7) return gram[i * 8 + b];
            return ((gram[i * 8 + 3] * jb[0] + gram[i * 8 + 4] * jb[1]) + gram[i * 8 + 5] * jb[2]) + gram[i * 8 + 6] * jb[3];
        };
        double v;
        if (a < 3) v = Mrow(a);
        else v = ((ja[0] * Mrow(3) + ja[1] * Mrow(4)) + ja[2] * Mrow(5)) + ja[3] * Mrow(6);
        if (tid < 36) H[a][b] = v; else gvec[a] = -v;
    }
    LILI_WAVE_SYNC();
    double pz[7] = {0, 0, 0, x0, x1, x2, x3};
    if (tid == 0) {
        // 6x6 LDL^T solve entirely in registers (all indices are compile-time constants after unrolling): six dependent
        // divisions (1/d_j) instead of the 6 square roots + 27 divisions of a Cholesky with per-element divides — the
        // f64 divide / sqrt sequences dominated this kernel's critical path
        double L[6][6], W[6][6], dinv[6], d[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            d[i] = gvec[i];
#pragma unroll
            for (int j = 0; j <= i; j++) L[i][j] = H[i][j];
        }
        bool okc = true;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double dj = L[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) dj -= L[j][k] * W[j][k];     // W[j][k] = L[j][k] * d_k
            if (!(dj > 0)) okc = false;
            dinv[j] = 1.0 / dj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
                double sv = L[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) sv -= L[i][k] * W[j][k];
                W[i][j] = sv;
                L[i][j] = sv * dinv[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {          // L z = g
            double sv = d[i];
#pragma unroll
            for (int k = 0; k < i; k++) sv -= L[i][k] * d[k];
            d[i] = sv;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) d[i] = d[i] * dinv[i];   // D y = z
#pragma unroll
        for (int i = 5; i >= 0; i--) {         // L^T x = y
            double sv = d[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) sv -= L[k][i] * d[k];
            d[i] = sv;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) if (!(d[i] == d[i])) okc = false;
        if (pub_key) { pz[0] = state->pose[0]; pz[1] = state->pose[1]; pz[2] = state->pose[2]; }
        if (okc) {
            if (pub_key) { pz[0] += d[0]; pz[1] += d[1]; pz[2] += d[2]; state->pose[0] = pz[0]; state->pose[1] = pz[1]; state->pose[2] = pz[2]; }
            else if (mirror) {
                const double p0 = state->pose[0] + d[0], p1 = state->pose[1] + d[1], p2 = state->pose[2] + d[2];
                state->pose[0] = p0; state->pose[1] = p1; state->pose[2] = p2; mirror->pose[0] = p0; mirror->pose[1] = p1; mirror->pose[2] = p2;
            } else { state->pose[0] += d[0]; state->pose[1] += d[1]; state->pose[2] += d[2]; }
            const double nd2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
            if (nd2 > 0.0) {
                double sbd, cw;
                if (nd2 < 0.25) sinc_cos_small(nd2, sbd, cw);
                else { const double nd = sqrt(nd2); sbd = sin(nd) / nd; cw = cos(nd); }
                dq qd{cw, sbd * d[3], sbd * d[4], sbd * d[5]};
                dq r = qmul(qd, dq{x0, x1, x2, x3});
                state->pose[3] = r.w; state->pose[4] = r.x; state->pose[5] = r.y; state->pose[6] = r.z;
                pz[3] = r.w; pz[4] = r.x; pz[5] = r.y; pz[6] = r.z;
                if (mirror) { mirror->pose[3] = r.w; mirror->pose[4] = r.x; mirror->pose[5] = r.y; mirror->pose[6] = r.z; }
            } else if (mirror) { mirror->pose[3] = x0; mirror->pose[4] = x1; mirror->pose[5] = x2; mirror->pose[6] = x3; }
#pragma unroll
            for (int i = 0; i < 6; i++) state->last_delta[i] = d[i];
            state->gn_status = 0;
            if (mirror) mirror->gn_status = 0;
        } else {
            state->gn_status = 1;
            if (mirror) { for (int i = 0; i < 3; i++) mirror->pose[i] = state->pose[i]; mirror->pose[3] = x0; mirror->pose[4] = x1; mirror->pose[5] = x2; mirror->pose[6] = x3; mirror->gn_status = 1; }
        }
        state->iters += 1;
    }
    if (pub_key) {          // lane 0's result to every lane, then one copy per lane: 7 x 64 sixteen-byte write-through stores
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)__double_as_longlong(pz[k]) & 0xffffffffull));
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)__double_as_longlong(pz[k]) >> 32));
            store_granule(pub + (size_t)tid * kPubStride + 2 * k, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)), pub_key);
        }
    }
}

constexpr int kReduceThreads = 1024;
// One chunk of the fixed-order partial sum: lane (g, e) takes partials g + 25 c' (c' = 8 c .. 8 c + 7) of `part`.  key == 0: plain
// 40-double partials written by an earlier launch.  key != 0: granules published by the blocks of THIS launch — the loads are
// repeated until every granule of the chunk carries the key (block-wide vote), at most kMaxSweeps times.
constexpr int kMaxSweeps = 1 << 16;
// Many plain partials (the per-wave partials of k_associate_lin: one per 64 queries): 32 loads in flight per lane instead of 8.  The
// additions are the same sequence (partials g, g + 25, g + 50, ... one after the other), only the round trips are fewer.
__device__ __forceinline__ void sum_partial_wide(const double* part, int nb, int c, int g, int e, int groups, double& s) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) { const int b = g + (c * 32 + u) * groups; v[u] = (g < groups && b < nb) ? part[(size_t)b * kPartialStride + e] : 0.0; }
#pragma unroll
    for (int u = 0; u < 32; u++) s += v[u];
}
__device__ __forceinline__ bool sum_partial_chunk(const double* part, int nb, int c, int g, int e, int groups, unsigned long long key, double& s) {
    double v[8];
    if (!key) {
#pragma unroll
        for (int u = 0; u < 8; u++) { const int b = g + (c * 8 + u) * groups; v[u] = (g < groups && b < nb) ? part[(size_t)b * kPartialStride + e] : 0.0; }
    } else {
        for (int sweep = 0;; sweep++) {
            unsigned long long lo[8], hi[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int b = g + (c * 8 + u) * groups;
                if (g < groups && b < nb) load_granule(part + (size_t)b * kPartialStride + 2 * e, lo[u], hi[u]);
                else { lo[u] = 0ull; hi[u] = key; }
            }
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 8; u++) ok = ok && ((lo[u] ^ hi[u]) == key);
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = __longlong_as_double((long long)lo[u]);
            if (__syncthreads_and(ok ? 1 : 0)) break;
            if (sweep >= kMaxSweeps) return false;          // uniform: a block of this launch never published (cannot happen unless the launch was cut short)
            __builtin_amdgcn_s_sleep(8);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
    return true;
}
// all kReduceThreads threads of ONE block.  The order of the additions is fixed (25 groups of 40 lanes, group g adds partials
// g, g+25, ... in sequence, then the groups in sequence), so the record does not depend on which launch structure produced the
// partials or on timing.
// XCHG: compiled with the exchange across ranks (k_reduce_partials); the fused tail of the linearisation launch has none.  Inlined into
// both kernels: the view is a by-value kernel argument and must not travel through memory.
template <bool XCHG>
__device__ __forceinline__ void reduce_partials_block(const double* part_surf, int nb_surf, const double* part_edge, int nb_edge,
                                                      double* __restrict__ out, SlotState* __restrict__ state, int do_gn, unsigned long long key, const P2PView& xv,
                                                      unsigned long long pub_key, double* pub, SlotState* __restrict__ mirror) {
    tstamp(state, do_gn, (int)blockIdx.x, 8);
    const unsigned long long was_dead = XCHG ? p2p_dead_word(xv) : 0ull;      // requested first: the round trip hides behind the partial loads
    const double xq[4] = {state->pose[3], state->pose[4], state->pose[5], state->pose[6]};
    constexpr int kGroups = kReduceThreads / 40;   // 25 groups of 40 lanes, group g adds partials g, g+25, ...
    __shared__ double acc[kGroups][2][40];
    __shared__ double tri[40];
    __shared__ double full[72];
    const int e = threadIdx.x % 40, g = threadIdx.x / 40;
    bool ok = true;
    if (key) {
        // cheap wait first: ONE granule per block partial (its last one), one lane each, until all carry the key; the full
        // sweep below (which verifies every granule it adds) then normally passes at once
        const int nb_all = nb_surf + nb_edge;
        for (int sweep = 0;; sweep++) {
            bool seen = true;
            for (int b = threadIdx.x; b < nb_all; b += blockDim.x) {
                const double* gp = (b < nb_surf ? part_surf + (size_t)b * kPartialStride : part_edge + (size_t)(b - nb_surf) * kPartialStride) + 2 * 39;
                unsigned long long lo, hi;
                load_granule(gp, lo, hi);
                seen = seen && ((lo ^ hi) == key);
            }
            if (__syncthreads_and(seen ? 1 : 0)) break;
            if (sweep >= kMaxSweeps) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    tstamp(state, do_gn, (int)blockIdx.x, 13);
    {
        double s = 0.0, s2 = 0.0;
        if (!key && nb_surf > 8 * kGroups) { for (int c = 0; c * 32 * kGroups < nb_surf; c++) sum_partial_wide(part_surf, nb_surf, c, g, e, kGroups, s); }
        else for (int c = 0; c * 8 * kGroups < nb_surf && ok; c++) ok = sum_partial_chunk(part_surf, nb_surf, c, g, e, kGroups, key, s);
        if (!key && nb_edge > 8 * kGroups) { for (int c = 0; c * 32 * kGroups < nb_edge; c++) sum_partial_wide(part_edge, nb_edge, c, g, e, kGroups, s2); }
        else for (int c = 0; c * 8 * kGroups < nb_edge && ok; c++) ok = sum_partial_chunk(part_edge, nb_edge, c, g, e, kGroups, key, s2);
        if (!ok) {           // uniform
            if (threadIdx.x == 0) { state->gn_status = 2; state->epoch = state->epoch + 1ull; }
            return;
        }
        if (g < kGroups) { acc[g][0][e] = s; acc[g][1][e] = s2; }
    }
    tstamp(state, do_gn, (int)blockIdx.x, 9);
    __syncthreads();
    if (threadIdx.x >= 64) return;      // the rest is ONE wave's work: no further block barriers
    const int lane = threadIdx.x;
    if (lane < 40) {
        double ss = 0.0, se = 0.0;
        for (int gg = 0; gg < kGroups; gg++) ss += acc[gg][0][lane];
        if (nb_edge > 0) for (int gg = 0; gg < kGroups; gg++) se += acc[gg][1][lane];
        tri[lane] = ss + se;
        if (lane == 37) { full[65] = ss; full[66] = se; }
    }
    LILI_WAVE_SYNC();
    {
        int r = lane >> 3, c = lane & 7;
        int a = r < c ? r : c, b = r < c ? c : r;
        full[lane] = tri[a * 8 - a * (a - 1) / 2 + (b - a)];
        if (lane == 0) full[64] = tri[36];
        if (lane >= 3 && lane < 8) full[64 + lane] = 0.0;     // 67..71
    }
    LILI_WAVE_SYNC();
    if (XCHG && xv.seq) {      // multi-GPU: the record of every rank, added in rank order (identical bits everywhere), inside this launch
        unsigned long long s0, s1;
        const unsigned long long w0 = (unsigned long long)__double_as_longlong(full[lane]);
        const unsigned long long w1 = lane < 8 ? (unsigned long long)__double_as_longlong(full[64 + lane]) : 0ull;
        if (!p2p_exchange_wave<true>(xv, 72, w0, w1, s0, s1, was_dead)) { if (lane == 0) state->gn_status = 2; return; }
        full[lane] = __longlong_as_double((long long)s0);
        if (lane < 8) full[64 + lane] = __longlong_as_double((long long)s1);
        LILI_WAVE_SYNC();
    }
    out[lane] = full[lane];
    if (lane < 8) out[64 + lane] = full[64 + lane];
    if (key && lane == 0) state->epoch = state->epoch + 1ull;      // the next fused launch of this slot gets a new key (stream order)
    if (lane == 0) state->cnt_word = 0ull;                          // re-arms the count barrier of k_associate_coop (the next association of this slot comes after this launch)
    tstamp(state, do_gn, (int)blockIdx.x, 10);
    if (do_gn & 1) gn_update_block(full, state, xq, pub_key, pub, mirror);
    tstamp(state, do_gn, (int)blockIdx.x, 11);
}

__global__ __launch_bounds__(kReduceThreads) void k_reduce_partials(const double* __restrict__ part_surf, int nb_surf,
                                                            const double* __restrict__ part_edge, int nb_edge,
                                                            double* __restrict__ out, SlotState* __restrict__ state, int do_gn, P2PView v, unsigned long long pub_key, double* pub,
                                                            SlotState* __restrict__ mirror /*page-locked copy of the pose and status for the host, or nullptr*/) {
    reduce_partials_block<true>(part_surf, nb_surf, part_edge, nb_edge, out, state, do_gn, 0ull, v, pub_key, pub, mirror);
}

// ================================================================================================
// Sliding window across ranks (BASELINE configs[4]; the reference evaluates the lidar blocks of ALL keyframes of the window per solver
// evaluation, L/src/BackendFusion.cpp:919-992): ONE launch reduces the block partials of every slot, and — with a lili_p2p view — ends
// with ONE exchange of the n x 72 doubles (rank-order sums: identical bits on every rank), optionally followed by the Gauss-Newton update
// of every slot.  The per-slot record is built with the additions of reduce_partials_block in the same order, so it equals what
// lili_s2m_linearize returns for that slot bit for bit.
// ================================================================================================
__device__ __forceinline__ void reduce_partials_plain(const double* part_surf, int nb_surf, const double* part_edge, int nb_edge, double* __restrict__ rec /*LDS, 72*/) {
    constexpr int kGroups = kReduceThreads / 40;
    __shared__ double acc[kGroups][2][40];
    __shared__ double tri[40];
    const int e = threadIdx.x % 40, g = threadIdx.x / 40;
    double s = 0.0, s2 = 0.0;
    if (nb_surf > 8 * kGroups) { for (int c = 0; c * 32 * kGroups < nb_surf; c++) sum_partial_wide(part_surf, nb_surf, c, g, e, kGroups, s); }
    else for (int c = 0; c * 8 * kGroups < nb_surf; c++) sum_partial_chunk(part_surf, nb_surf, c, g, e, kGroups, 0ull, s);
    if (nb_edge > 8 * kGroups) { for (int c = 0; c * 32 * kGroups < nb_edge; c++) sum_partial_wide(part_edge, nb_edge, c, g, e, kGroups, s2); }
    else for (int c = 0; c * 8 * kGroups < nb_edge; c++) sum_partial_chunk(part_edge, nb_edge, c, g, e, kGroups, 0ull, s2);
    if (g < kGroups) { acc[g][0][e] = s; acc[g][1][e] = s2; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane < 40) {
            double ss = 0.0, se = 0.0;
            for (int gg = 0; gg < kGroups; gg++) ss += acc[gg][0][lane];
            if (nb_edge > 0) for (int gg = 0; gg < kGroups; gg++) se += acc[gg][1][lane];
            tri[lane] = ss + se;
            if (lane == 37) { rec[65] = ss; rec[66] = se; }
        }
        LILI_WAVE_SYNC();
        const int r = lane >> 3, c = lane & 7;
        const int a = r < c ? r : c, b = r < c ? c : r;
        rec[lane] = tri[a * 8 - a * (a - 1) / 2 + (b - a)];
        if (lane == 0) rec[64] = tri[36];
        if (lane >= 3 && lane < 8) rec[64 + lane] = 0.0;     // 67..71
    }
    __syncthreads();      // acc / tri are free for the next slot; rec is visible to the block
}
__global__ __launch_bounds__(kReduceThreads) void k_window_reduce(WindowArgs w, double* __restrict__ out /*n x 72*/, int do_gn, P2PView v) {
    __shared__ double rec[kWindowMaxSlots * 72];
    __shared__ double xq[kWindowMaxSlots][4];
    const unsigned long long was_dead = p2p_dead_word(v);      // requested first: the round trip hides behind the partial loads
    for (int i = 0; i < w.n; i++) if (threadIdx.x < 4) xq[i][threadIdx.x] = w.s[i].state->pose[3 + threadIdx.x];
    for (int i = 0; i < w.n; i++) reduce_partials_plain(w.s[i].part_surf, w.s[i].nb_surf, w.s[i].part_edge, w.s[i].nb_edge, rec + 72 * i);
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, count = 72 * w.n;
    if (v.seq) {
        unsigned long long ww[kWindowMaxSlots * 72 / 64], ss[kWindowMaxSlots * 72 / 64];
#pragma unroll
        for (int i = 0; i < kWindowMaxSlots * 72 / 64; i++) { ww[i] = lane + 64 * i < count ? (unsigned long long)__double_as_longlong(rec[lane + 64 * i]) : 0ull; ss[i] = 0ull; }
        if (!p2p_exchange_words<true, kWindowMaxSlots * 72 / 64>(v, count, ww, ss, was_dead)) { if (lane == 0) for (int i = 0; i < w.n; i++) w.s[i].state->gn_status = 2; return; }
#pragma unroll
        for (int i = 0; i < kWindowMaxSlots * 72 / 64; i++) if (lane + 64 * i < count) rec[lane + 64 * i] = __longlong_as_double((long long)ss[i]);
        LILI_WAVE_SYNC();
    }
    for (int i = lane; i < count; i += 64) out[i] = rec[i];
    if (do_gn) for (int i = 0; i < w.n; i++) { gn_update_block(rec + 72 * i, w.s[i].state, xq[i]); LILI_WAVE_SYNC(); }
}
// the Gauss-Newton update of every slot from its (all-reduced) record: the generic-collective form of k_window_reduce's tail
__global__ __launch_bounds__(64) void k_window_gn(WindowArgs w, const double* __restrict__ gram /*n x 72*/) {
    for (int i = 0; i < w.n; i++) {
        const double xq[4] = {w.s[i].state->pose[3], w.s[i].state->pose[4], w.s[i].state->pose[5], w.s[i].state->pose[6]};
        gn_update_block(gram + 72 * i, w.s[i].state, xq);
        LILI_WAVE_SYNC();
    }
}
// correspondence counts of every slot ([surf, edge] per slot) in one launch; with a lili_p2p view the 2 n totals are all-reduced inside:
// `out` and the slots' states then hold the GLOBAL counts (ROT: residual scale = num / global N, R/src/BackendFusion.cpp:843,861)
__global__ __launch_bounds__(kBlock) void k_window_counts(WindowArgs w, int* __restrict__ out /*2 n*/, P2PView v) {
    __shared__ int tot[2 * kWindowMaxSlots];
    const unsigned long long was_dead = p2p_dead_word(v);
    for (int i = 0; i < w.n; i++) {
        const int t0 = w.s[i].bc_surf ? sum_block_counts(w.s[i].bc_surf, w.s[i].nbc_surf) : 0;
        __syncthreads();
        const int t1 = w.s[i].bc_edge ? sum_block_counts(w.s[i].bc_edge, w.s[i].nbc_edge) : 0;
        __syncthreads();
        if (threadIdx.x == 0) { tot[2 * i] = t0; tot[2 * i + 1] = t1; }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, count = 2 * w.n;
    int val = lane < count ? tot[lane] : 0;
    if (v.seq) {
        const unsigned long long ww[1] = {(unsigned long long)(unsigned)val};
        unsigned long long ss[1] = {0ull};
        if (!p2p_exchange_words<false, 1>(v, count, ww, ss, was_dead)) { if (lane == 0) for (int i = 0; i < w.n; i++) w.s[i].state->gn_status = 2; return; }
        val = (int)(unsigned)ss[0];
    }
    if (lane < count) out[lane] = val;
    for (int i = 0; i < w.n; i++) if ((lane >> 1) == i && lane < count) w.s[i].state->n_res[lane & 1] = val;      // (uniform index into the argument struct)
}

// Restart of a registration: 56 bytes device to device.  hipMemcpyAsync(D2D) costs a 4.5 us copy kernel for this; one 8-lane
// workgroup of our own is done in well under half of that.
__global__ void k_pose_copy(SlotState* __restrict__ dst, const SlotState* __restrict__ src) {
    if (threadIdx.x < 7) dst->pose[threadIdx.x] = src->pose[threadIdx.x];
}

__global__ void k_gn_update(const double* __restrict__ gram, SlotState* __restrict__ state) {
    const double xq[4] = {state->pose[3], state->pose[4], state->pose[5], state->pose[6]};
    gn_update_block(gram, state, xq);
}

}  // namespace lili
