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

__device__ __forceinline__ int cell_coord(float v, double o, double inv_cell) {
    // f64 so that the covering argument of DESIGN.md §3 does not depend on f32 rounding of (v - o) / c
    return (int)floor(((double)v - o) * inv_cell);
}
__device__ __forceinline__ int cell_of(float4 p, const GridView& g) {
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return 0;   // never selected: its distance is NaN
    int cx = min(max(cell_coord(p.x, g.ox, g.inv_cell), 0), g.nx - 1);
    int cy = min(max(cell_coord(p.y, g.oy, g.inv_cell), 0), g.ny - 1);
    int cz = min(max(cell_coord(p.z, g.oz, g.inv_cell), 0), g.nz - 1);
    return (cz * g.ny + cy) * g.nx + cx;
}

// One global atomic per point in the WHOLE build: the value it returns is the point's rank inside its cell, kept next to the cell id,
// so the scatter pass needs no second counter array, no second 108 MB memset and no atomics (round 1: an atomic here, whose result
// was dropped, and another one in k_scatter).
// Neighbouring lanes that fall into the same cell (maps come out of the voxel filter in voxel order: consecutive points are
// neighbours in x) share ONE atomic: run heads add the run length, the members take base + offset.  No loop, ~10 instructions;
// an unordered cloud degenerates to one atomic per point.  All 64 lanes of a wave must be active.
__global__ void k_cell_count(const float4* __restrict__ pts, int n, GridView g, int* __restrict__ cell_count, int2* __restrict__ pt_cell,
                             unsigned long long* __restrict__ rank_sum /*optional*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = i < n ? cell_of(pts[i], g) : -1;
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
    if (i < n) pt_cell[i] = make_int2(c, rank);
    // density estimate for free: a point's rank is its position inside its cell, so sum(rank) = sum over cells of occ (occ - 1) / 2 and the
    // point-weighted mean cell occupancy is 1 + 2 sum(rank) / n (lili_map_set decides on the fine grid with it).  One atomic per block,
    // spread over 64 banks that the host adds up.
    if (rank_sum) {
        unsigned long long r64 = i < n && c >= 0 ? (unsigned long long)rank : 0ull;
        for (int o = 32; o > 0; o >>= 1) r64 += __shfl_xor(r64, o);
        __shared__ unsigned long long wsum[kBlock / 64];
        if (lane == 0) wsum[threadIdx.x >> 6] = r64;
        __syncthreads();
        // 64 banks, 128 bytes apart (same-address atomics cost ~12 ns each: 19.5 k blocks on ONE word were 0.2 ms of a 5 M-point build)
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < kBlock / 64; w++) t += wsum[w]; atomicAdd(rank_sum + (size_t)(blockIdx.x & 63) * 16, t); }
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
// map) is read once and written once, where the three-kernel scan above reads it twice and writes it once.  Tile b (4096 items) belongs
// to workgroup b — no ticket: 6.6 k same-address atomics cost ~12 ns each on MI355X, more than the whole scan (measured: 183 us with a
// ticket against 170 us for the three kernels).  Workgroups are dispatched in index order per XCD, so the lowest unfinished tile is
// always resident and a tile only ever waits for lower ones; should that ever not hold, the bounded spin gives up (ws[1] = 1)
// instead of hanging the GPU.  Every tile publishes {flag, value} as ONE 64-bit word (flag 1 = the tile's own sum, 2 = the inclusive
// prefix up to and including the tile), written and read with agent-scope atomics — the word is its own flag, no fence.  One wave
// per tile looks back 64 predecessors at a time.  `ws`: [0], [1] = error flag, [2 ...] one status word per tile — zeroed by the
// caller.  data[n] receives the total.
constexpr int kLbItems = 64;
constexpr int kLbTile = kBlock * kLbItems;
__global__ __launch_bounds__(kBlock) void k_scan_lookback(int* data, int64_t n, unsigned long long* __restrict__ ws) {
    __shared__ int lds[kBlock / 64 + 1];
    __shared__ int s_prefix;
    const int tile = (int)blockIdx.x;
    unsigned long long* status = ws + 2;
    const int64_t base = (int64_t)tile * kLbTile + (int64_t)threadIdx.x * kLbItems;
    int v[kLbItems]; int s = 0;
    if (base + kLbItems <= n) {
        const int4* p4 = reinterpret_cast<const int4*>(data + base);        // 16-byte loads: base is a multiple of 16 items
#pragma unroll
        for (int k = 0; k < kLbItems / 4; k++) { const int4 q = p4[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
    } else {
#pragma unroll
        for (int k = 0; k < kLbItems; k++) v[k] = base + k < n ? data[base + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < kLbItems; k++) s += v[k];
    int tot;
    const int ex = block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) {
        const unsigned long long w = ((tile == 0 ? 2ull : 1ull) << 32) | (unsigned long long)(unsigned)tot;
        __hip_atomic_store(&status[tile], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile == 0) s_prefix = 0;
    }
    if (tile > 0 && threadIdx.x < 64) {          // one wave walks back until it meets an inclusive prefix
        const int lane = threadIdx.x;
        int acc = 0;
        for (int hi = tile - 1; ; hi -= 64) {
            const int t = hi - lane;
            unsigned long long w = 2ull << 32;       // tiles before 0: "prefix 0"
            if (t >= 0) {
                int spins = 0;
                do {
                    w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > (1 << 21)) { ws[1] = 1ull; w = 2ull << 32; }      // cannot happen while lower tiles run; never hang the GPU
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
    int run = s_prefix + ex;
    if (base + kLbItems <= n) {
        int4* p4 = reinterpret_cast<int4*>(data + base);
#pragma unroll
        for (int k = 0; k < kLbItems / 4; k++) {
            int4 q;
            q.x = run; run += v[4 * k]; q.y = run; run += v[4 * k + 1]; q.z = run; run += v[4 * k + 2]; q.w = run; run += v[4 * k + 3];
            p4[k] = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kLbItems; k++) { if (base + k < n) data[base + k] = run; run += v[k]; }
    }
    if (base <= n - 1 && n - 1 < base + kLbItems) data[n] = s_prefix + ex + s;     // the thread that owns the last item: total
}

// Super-rows: a second copy of the map in which the points of the 3x3 (y,z) rows around a row are stored together, sorted by
// x cell — "super cell" (x, y', z') holds the points of the nine cells (x, y'+dy, z'+dz) in the fixed order k = (dz+1)*3 + (dy+1), each in
// its base order.  The 27-cell neighbourhood of a query in cell (cx, cy, cz) is then ONE contiguous run: super cells cx-1..cx+1 of super-row
// (cy, cz) — two range words instead of eighteen, no row table, no per-row bounds, full chunks.  Super-rows exist for the cells of a box
// (bx0.., by0.., bz0..; the whole grid unless lili_map_focus names a region): queries elsewhere take the nine-row walk.  The copy lives behind
// the base points in the same array ([base | super-rows], at most 10 n entries); start9 holds positions in that array.
__device__ __forceinline__ size_t srow_index(const GridView& g, int x, int y, int z) {
    return ((size_t)(z - g.bz0) * (size_t)g.bny + (size_t)(y - g.by0)) * (size_t)g.bnx + (size_t)(x - g.bx0);
}
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
    if (x >= g.bx0 + g.bnx) return;
#pragma unroll
    for (int k = 0; k < kS9Rows; k++) {
        const int ys = y0 + k;
        if (ys < g.by0 + g.bny) start9[srow_index(g, x, ys, zs)] = g.n_points + rowbase[(zs - g.bz0) * g.bny + (ys - g.by0)] + c3[k] + c3[k + 1] + c3[k + 2];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        start9[(size_t)g.bnx * g.bny * g.bnz] = g.n_points + rowbase[g.bny * g.bnz];          // end of the array
}

__global__ void k_scatter(const float4* __restrict__ pts, int n, const int2* __restrict__ pt_cell, const int* __restrict__ cell_start,
                          float4* __restrict__ sorted, float* __restrict__ aux_sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int2 cr = pt_cell[i];
    const int pos = cell_start[cr.x] + cr.y;
    float4 p = pts[i];
    if (aux_sorted) aux_sorted[pos] = p.w;
    p.w = __int_as_float(i);
    sorted[pos] = p;
}

// Super-row copy, by DESTINATION: one wave owns the stretch of 64 consecutive super cells (x0..x0+63, y', z') — a contiguous piece of the
// super-row array — and fills it from its nine source rows, each a contiguous run of the base array.  The nine runs are walked as ONE
// sequence, 64 points per trip; a point finds its super cell through its own x cell and its place through a 9 x 64 table (LDS) of
// "destination minus source" per (source row, cell).  The 16 waves of a workgroup take a 4 x 4 tile of (y', z'), whose 36 source rows
// they share through the L2 of the XCD the workgroup runs on.
__global__ __launch_bounds__(1024) void k_scatter9(const int* __restrict__ cell_start, GridView g, const int* __restrict__ start9,
                                                   float4* __restrict__ sorted, float* __restrict__ aux_sorted) {
    __shared__ int delta[16][9][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x0 = g.bx0 + blockIdx.x * 64;
    const int ys = g.by0 + (int)blockIdx.y * 4 + (w & 3);
    const int zs = g.bz0 + (int)blockIdx.z * 4 + (w >> 2);
    if (ys >= g.by0 + g.bny || zs >= g.bz0 + g.bnz) return;
    const int xn = min(64, g.bx0 + g.bnx - x0);            // cells of this stretch
    const int* s9 = start9 + srow_index(g, x0, ys, zs);
    const int lc = min(lane, xn - 1);
    int dnext = s9[lc];                                    // where the next source's points of cell x0+lane go
    const int d_end = s9[xn];
    const int D0 = __shfl(dnext, 0);
    const int len = d_end - D0;
    if (len == 0) return;                                  // nothing lives here
    int off[9], pre[10];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int y = ys + k % 3 - 1, z = zs + k / 3 - 1;
        const bool in = y >= 0 && y < g.ny && z >= 0 && z < g.nz;
        const int* cs = cell_start + ((size_t)(in ? z : 0) * g.ny + (in ? y : 0)) * g.nx + x0;
        const int a = cs[lc], b = cs[lc + 1], e = cs[xn];
        delta[w][k][lane] = dnext - a;
        dnext += in ? b - a : 0;
        const int rb = in ? __shfl(a, 0) : 0, re = in ? e : 0;
        off[k] = rb - pre[k];                              // position in the base array = position in the merged sequence + off[k]
        pre[k + 1] = pre[k] + (re - rb);
    }
    __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < len; t0 += 64) {                 // pre[9] == len
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

// ================================================================================================
// Query binning (once per scan): order the queries by the Morton code of the map super-cell
// (sb x sb cells in x,y) they fall into at the association pose, so that the 64 lanes of a wave walk
// the same cell runs (coalesced / broadcast loads, uniform loop trip counts).  The order only decides
// which thread handles which query — every per-query result is written at the query's own index, so
// results do not depend on it.
// ================================================================================================
__device__ __forceinline__ unsigned part1by1(unsigned x) {
    x &= 0x0000ffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
__device__ __forceinline__ void load_assoc_pose(const PoseArg& pa, const MatchParams& P, dq& Q2, d3& T2);

// atomicAdd(&arr[key], 1) for every active lane, issued as ONE atomic per distinct key per wave (queries
// of a wave mostly share a bin; 200 k same-address atomics cost milliseconds).  Returns the value before
// this wave's add for the lane's key; rank = the lane's position among the wave's lanes with that key.
// Must be called by all 64 lanes.
__device__ __forceinline__ int wave_aggregated_add(int* __restrict__ arr, int key, bool active, int& rank) {
    const int lane = threadIdx.x & 63;
    int result = 0;
    rank = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int k0 = __shfl(key, leader);
        unsigned long long m = __ballot(active && key == k0);
        int old = 0;
        if (lane == leader) old = atomicAdd(&arr[k0], __popcll(m));
        old = __shfl(old, leader);
        if (active && key == k0) { result = old; rank = __popcll(m & ((1ull << lane) - 1ull)); }
        todo &= ~m;
    }
    return result;
}

__global__ void k_bin_count(const float4* __restrict__ queries, int n_q, GridView g, PoseArg pa, MatchParams P, int sb_shift, int n_bins,
                            int* __restrict__ keys, int* __restrict__ hist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_q) i = n_q - 1;   // keep whole waves alive for the aggregated atomics; duplicates are masked below
    const bool live = blockIdx.x * blockDim.x + threadIdx.x < n_q;
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    float4 ql = queries[i];
    d3 pm = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
    float px = (float)pm.x, py = (float)pm.y;
    int key = n_bins - 1;   // non-finite queries go last
    if (isfinite(px) && isfinite(py)) {
        double fx = ((double)px - g.ox) * g.inv_cell, fy = ((double)py - g.oy) * g.inv_cell;
        int cx = (int)fmin(fmax(fx, 0.0), (double)(g.nx - 1));
        int cy = (int)fmin(fmax(fy, 0.0), (double)(g.ny - 1));
        unsigned k = part1by1((unsigned)(cx >> sb_shift)) | (part1by1((unsigned)(cy >> sb_shift)) << 1);
        key = (int)min(k, (unsigned)(n_bins - 1));
    }
    if (live) keys[i] = key;
    int rank;
    wave_aggregated_add(hist, key, live, rank);
}
__global__ void k_bin_scatter(const int* __restrict__ keys, int n_q, const int* __restrict__ starts, int* __restrict__ fill, int* __restrict__ perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_q;
    int k = live ? keys[i] : 0;
    int rank;
    int base = wave_aggregated_add(fill, k, live, rank);
    if (live) perm[starts[k] + base + rank] = i;
}

// Tile list: a tile is <= kBlock consecutive entries of `perm` that all belong to ONE bin (super-cell), so a
// block's neighbourhood is bounded by the super-cell size.  counts[b] = queries in bin b.
__global__ void k_tile_count(const int* __restrict__ counts, int n_bins, int* __restrict__ tile_cnt) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_bins) tile_cnt[b] = (counts[b] + kBlock - 1) / kBlock;
}
__global__ void k_tile_fill(const int* __restrict__ counts, const int* __restrict__ starts, const int* __restrict__ tile_off, int n_bins,
                            int2* __restrict__ tiles) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    int c = counts[b], s0 = starts[b], t0 = tile_off[b];
    for (int k = 0; k * kBlock < c; k++) tiles[t0 + k] = make_int2(s0 + k * kBlock, min(kBlock, c - k * kBlock));
}

// ================================================================================================
// exact 5-NN inside the 27-cell neighbourhood
// ================================================================================================
struct Top5 {
    float d[5];
    int j[5];     // position in the cell-sorted array
    int aux;      // profiling only: chunks of four candidates this query went through
    float4 p[5];  // the five points themselves when `have` (the key selector has just loaded them to recompute the exact distances:
    bool have;    //  the fit then skips its own gather)
};
// FLANN L2_Simple on 3 floats (f32, x then y then z, no FMA)
__device__ __forceinline__ float dist2(float4 p, float qx, float qy, float qz) {
    float r = 0.f;
    float dx = qx - p.x; r += dx * dx;
    float dy = qy - p.y; r += dy * dy;
    float dz = qz - p.z; r += dz * dz;
    return r;
}
// Running 5 best as a sorted list of packed keys (f32 distance bits << 32 | original map index): squared
// distances are >= 0 so their bit patterns order like the values, and the low word makes the order the
// oracle's lexicographic (d2, index) — FLANN's own tie order is unspecified (App. B1).  Insertion is a
// 5-stage compare-exchange chain without branches: on a 64-lane wave some lane inserts at almost every
// candidate, so a branchy insertion path is executed (and stalls) nearly every iteration anyway.
// NaN distances have bit patterns above +inf and therefore never displace the initial (+inf, INT_MAX) keys.
struct Sel5 {
    unsigned long long k[5];
    int j[5];
    // `bound`: candidates enter only with d <= bound.  +inf gives the plain 5-NN; the association passes the gate of the
    // reference (`pointSearchSqDis[4] < gate`, rounded UP to f32), so that rows and shell cells beyond the gate are pruned
    // from the first candidate on — queries without 5 neighbours inside the gate then end with d[4] = bound >= gate, j = -1.
    __device__ __forceinline__ void init(float bound = 3.0e38f) {
        bound = fminf(bound, 3.0e38f);   // finite, so that the +inf of masked slots never qualifies
#pragma unroll
        for (int s = 0; s < 5; s++) { k[s] = ((unsigned long long)__float_as_uint(bound) << 32) | 0x7fffffffull; j[s] = -1; }
    }
    // Sorted insertion by rank: the five comparisons are independent (no compare-exchange chain), slot s takes its left
    // neighbour if the key ranks before s-1, the key itself if it ranks exactly at s, else keeps its value.
    __device__ __forceinline__ void insert(float d, float4 p, int jpos) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        const bool c0 = key < k[0], c1 = key < k[1], c2 = key < k[2], c3 = key < k[3], c4 = key < k[4];
        k[4] = c3 ? k[3] : (c4 ? key : k[4]);  j[4] = c3 ? j[3] : (c4 ? jpos : j[4]);
        k[3] = c2 ? k[2] : (c3 ? key : k[3]);  j[3] = c2 ? j[2] : (c3 ? jpos : j[3]);
        k[2] = c1 ? k[1] : (c2 ? key : k[2]);  j[2] = c1 ? j[1] : (c2 ? jpos : j[2]);
        k[1] = c0 ? k[0] : (c1 ? key : k[1]);  j[1] = c0 ? j[0] : (c1 ? jpos : j[1]);
        k[0] = c0 ? key : k[0];                j[0] = c0 ? jpos : j[0];
    }
    __device__ __forceinline__ float worst() const { return __uint_as_float((unsigned)(k[4] >> 32)); }
    __device__ __forceinline__ bool final_tie() const { return false; }
    __device__ __forceinline__ unsigned worst_bits() const { return (unsigned)(k[4] >> 32); }
    __device__ __forceinline__ void to_top5(Top5& t) const {
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = __uint_as_float((unsigned)(k[s] >> 32)); t.j[s] = j[s]; }
        t.have = false;
    }
};

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {   // no clang builtin for the integer median
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Key selector (default fast tier): ONE 32-bit key per candidate and no payload moves.
//   key = (distance bits & ~63) | code        code = ((chunk & 7) << 2) | slot-in-chunk      (a "bucket" = 64 f32 ulps of d^2)
// The SIX smallest keys are kept sorted with five v_med3_u32 and one v_min_u32 per candidate — no comparisons, no selects,
// no wave-level branch.  Where a key came from is recovered afterwards from a small per-lane LDS table
// (T[0..7] = first array position of the last eight chunks, T[8..13] = positions resolved so far, T[15] = -4 for the
// sentinel code 63): position = T[code >> 2] + (code & 3); every eight chunks the six held keys are re-pointed to T[8..13].
// Exactness: buckets are monotone in the distance, so every candidate outside the held six has a bucket >= the sixth's.
// If the sixth key's bucket differs from the fifth's, the five best are exactly the five smallest distances (as a set);
// their exact f32 distances are then recomputed from the re-loaded points and put into the oracle's (d2, index) order.
// If the buckets coincide, or the fifth lies in the bucket of the search bound, the query is repeated with the exact
// selector (about once in 1e5 queries on voxel-filtered maps; always on lattice ties).
struct Sel5K {
    unsigned k[6];
    int tc;          // chunks processed by this lane
    int* T;          // this lane's column of the chunk table (row stride ts ints)
    int ts;
    unsigned bb;     // bucket of the bound
    float bnd;
    __device__ __forceinline__ void init(float bound) {
        bnd = fminf(bound, 3.0e38f);
        bb = __float_as_uint(bnd) >> 6;
#pragma unroll
        for (int s = 0; s < 6; s++) k[s] = ((bb + 1u + (unsigned)s) << 6) | 63u;   // six distinct buckets above the bound
        tc = 0; T = nullptr; ts = 0;
    }
    __device__ __forceinline__ void attach(int* col, int stride) { T = col; ts = stride; T[15 * stride] = -4; }
    __device__ __forceinline__ float worst() const { return __uint_as_float(k[4] | 63u); }   // upper end of the bucket: pruning stays conservative
    __device__ __forceinline__ unsigned worst_bits() const { return k[4] | 63u; }
    __device__ __forceinline__ void to_top5(Top5& t) const {   // only meaningful right after init (early exits)
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = bnd; t.j[s] = -1; }
        t.have = false;
    }
    __device__ __forceinline__ void push(unsigned key) {
        const unsigned m5 = umed3(k[4], k[5], key), m4 = umed3(k[3], k[4], key), m3 = umed3(k[2], k[3], key);
        const unsigned m2 = umed3(k[1], k[2], key), m1 = umed3(k[0], k[1], key);
        k[0] = min(k[0], key); k[1] = m1; k[2] = m2; k[3] = m3; k[4] = m4; k[5] = m5;
    }
    __device__ __forceinline__ int where(unsigned key) const { const unsigned c = key & 63u; return T[(int)(c >> 2) * ts] + (int)(c & 3u); }
    __device__ __forceinline__ void repoint() {
        int jr[6];
#pragma unroll
        for (int s = 0; s < 6; s++) jr[s] = where(k[s]);
#pragma unroll
        for (int s = 0; s < 6; s++) { T[(8 + s) * ts] = jr[s]; k[s] = (k[s] & ~63u) | (unsigned)(32 + 4 * s); }
    }
    __device__ __forceinline__ void chunk(float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
        if (tc >= 8 && (tc & 7) == 0) repoint();
        T[(tc & 7) * ts] = j;
        const unsigned code = (unsigned)(tc & 7) << 2;
        tc++;
        const unsigned u0 = __float_as_uint(dist2(p0, qx, qy, qz));
        const unsigned u1 = j + 1 < end ? __float_as_uint(dist2(p1, qx, qy, qz)) : 0x7f800000u;
        const unsigned u2 = j + 2 < end ? __float_as_uint(dist2(p2, qx, qy, qz)) : 0x7f800000u;
        const unsigned u3 = j + 3 < end ? __float_as_uint(dist2(p3, qx, qy, qz)) : 0x7f800000u;
        push((u0 & ~63u) | code); push((u1 & ~63u) | (code + 1u)); push((u2 & ~63u) | (code + 2u)); push((u3 & ~63u) | (code + 3u));
    }
    // Resolves the five best, recomputes their exact distances and orders them by (d2, original index).
    // Returns true if the query has to be repeated with the exact selector.
    __device__ __forceinline__ bool finish(const GridView& g, float qx, float qy, float qz, Top5& t) const {
        const bool redo = ((k[5] ^ k[4]) < 64u) || ((k[4] >> 6) == bb);
        unsigned long long e[5];
        int jr[5];
        float4 pp[5];
#pragma unroll
        for (int s = 0; s < 5; s++) {
            jr[s] = where(k[s]);
            const bool real = jr[s] >= 0;
            const float4 p = g.pts[real ? jr[s] : 0];
            pp[s] = p;
            const unsigned du = real ? __float_as_uint(dist2(p, qx, qy, qz)) : __float_as_uint(bnd);
            const unsigned lo = real ? (unsigned)__float_as_int(p.w) : 0x7fffffffu;
            e[s] = ((unsigned long long)du << 32) | lo;
        }
        const bool unsorted = !(e[0] <= e[1] && e[1] <= e[2] && e[2] <= e[3] && e[3] <= e[4]);   // real keys are unique (distinct indices); equal keys are sentinels
        if (__any(unsorted)) {   // within-bucket inversion somewhere in the wave (rare): 9 compare-exchanges
#define LILI_CE(a, b) { const bool sw = e[b] < e[a]; const unsigned long long ea = e[a], eb = e[b]; const int ja = jr[a], jb = jr[b]; \
                        const float4 pa_ = pp[a], pb_ = pp[b]; \
                        e[a] = sw ? eb : ea; e[b] = sw ? ea : eb; jr[a] = sw ? jb : ja; jr[b] = sw ? ja : jb; \
                        pp[a].x = sw ? pb_.x : pa_.x; pp[a].y = sw ? pb_.y : pa_.y; pp[a].z = sw ? pb_.z : pa_.z; pp[a].w = sw ? pb_.w : pa_.w; \
                        pp[b].x = sw ? pa_.x : pb_.x; pp[b].y = sw ? pa_.y : pb_.y; pp[b].z = sw ? pa_.z : pb_.z; pp[b].w = sw ? pa_.w : pb_.w; }
            LILI_CE(0, 1) LILI_CE(3, 4) LILI_CE(2, 4) LILI_CE(2, 3) LILI_CE(0, 3) LILI_CE(0, 2) LILI_CE(1, 4) LILI_CE(1, 3) LILI_CE(1, 2)
#undef LILI_CE
        }
#pragma unroll
        for (int s = 0; s < 5; s++) { t.d[s] = __uint_as_float((unsigned)(e[s] >> 32)); t.j[s] = jr[s]; t.p[s] = pp[s]; }
        t.aux = tc;
        t.have = !redo;
        return redo;
    }
};
__device__ __forceinline__ void process_chunk(Sel5K& sel, float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
    asm volatile("" : "+v"(p0.w), "+v"(p1.w), "+v"(p2.w), "+v"(p3.w));
    sel.chunk(p0, p1, p2, p3, j, end, qx, qy, qz);
}

// Lower bound (conservative by 0.1 %) of the f32 squared distance from the query to any point of the cell row
// (cy+dy, cz+dz): the gap to the own cell's boundary in y and z.  Rows whose bound exceeds the current 5th best
// cannot contribute (a candidate enters only with d <= that value) — skipping them keeps the search exact.
__device__ __forceinline__ float row_lower_bound(const GridView& g, float qy, float qz, int cy, int cz, int dy, int dz) {
    const double c = g.cell;
    double gy = dy == 0 ? 0.0 : (dy < 0 ? (double)qy - (g.oy + (double)cy * c) : (g.oy + (double)(cy + 1) * c) - (double)qy);
    double gz = dz == 0 ? 0.0 : (dz < 0 ? (double)qz - (g.oz + (double)cz * c) : (g.oz + (double)(cz + 1) * c) - (double)qz);
    gy = fmax(gy, 0.0); gz = fmax(gz, 0.0);
    return (float)(0.999 * (gy * gy + gz * gz));
}
// visiting order of the 9 (dy,dz) rows: centre, faces, diagonals (w = (dy+1)*3 + (dz+1))
__device__ __forceinline__ int row_order(int n) { return n == 0 ? 4 : n == 1 ? 1 : n == 2 ? 3 : n == 3 ? 5 : n == 4 ? 7 : n == 5 ? 0 : n == 6 ? 2 : n == 7 ? 6 : 8; }

// One run of consecutive cell-sorted map points.  Four independent loads are in flight per trip (the search is bound
// by the length of its dependent-load chain, not by bandwidth); lanes past their run end re-load the run's last point
// and give it a NaN distance, whose key can never enter the selection.
__device__ __forceinline__ float4 load_pt(const GridView& g, int j) {   // 32-bit byte offset from the uniform base (map < 2^28 points)
    return *(const float4*)((const char*)g.pts + ((unsigned)j << 4));
}
// Four consecutive candidates [j, j+4) of a run ending at `end` (slots past the end were loaded from the run's last point
// and get +inf, which no selector accepts; NaN distances of non-finite map points likewise: fminf).
template <class SEL>
__device__ __forceinline__ void process_chunk(SEL& sel, float4 p0, float4 p1, float4 p2, float4 p3, int j, int end, float qx, float qy, float qz) {
    const int last = end - 1;
    asm volatile("" : "+v"(p0.w), "+v"(p1.w), "+v"(p2.w), "+v"(p3.w));   // keep each point ONE 16-byte load (no re-load of .w inside the branches)
    // distances as bit patterns (non-negative floats order like unsigned integers; a NaN distance — non-finite map point —
    // sorts above every bound and is never accepted); slots past the run end get +inf
    const unsigned u0 = __float_as_uint(dist2(p0, qx, qy, qz));
    const unsigned u1 = j + 1 < end ? __float_as_uint(dist2(p1, qx, qy, qz)) : 0x7f800000u;
    const unsigned u2 = j + 2 < end ? __float_as_uint(dist2(p2, qx, qy, qz)) : 0x7f800000u;
    const unsigned u3 = j + 3 < end ? __float_as_uint(dist2(p3, qx, qy, qz)) : 0x7f800000u;
    // each test is a wave-level skip of the selection code (taken if any lane qualifies)
    if (u0 <= sel.worst_bits()) sel.insert(__uint_as_float(u0), p0, j);
    if (u1 <= sel.worst_bits()) sel.insert(__uint_as_float(u1), p1, min(j + 1, last));
    if (u2 <= sel.worst_bits()) sel.insert(__uint_as_float(u2), p2, min(j + 2, last));
    if (u3 <= sel.worst_bits()) sel.insert(__uint_as_float(u3), p3, min(j + 3, last));
}
// One run of consecutive cell-sorted map points, four independent loads in flight per trip (shell phase).
template <class SEL>
__device__ __forceinline__ void scan_run(const GridView& g, SEL& sel, int beg, int end, float qx, float qy, float qz) {
    const int last = end - 1;
    for (int j = beg; j < end; j += 4) {
        float4 p0 = load_pt(g, j), p1 = load_pt(g, min(j + 1, last)), p2 = load_pt(g, min(j + 2, last)), p3 = load_pt(g, min(j + 3, last));
        process_chunk(sel, p0, p1, p2, p3, j, end, qx, qy, qz);
    }
}

// Per-thread table of the non-empty rows of the inner 3x3 block (LDS, one column per thread): run begin / end in the
// cell-sorted array and the row's lower distance bound.
template <int BS>
struct RowTabT {
    int b[9][BS];
    int e[9][BS];
    float lb[9][BS];
    int cj[16][BS];   // Sel5K: array positions of the last eight chunks / resolved positions / sentinel
};

// Exact 5-NN among the map points of the (2*reach+1)^3 cells around the query.  reach = 1: the 27 cells (9 runs).
// reach = 2 (cells of half the size): the inner 27 cells first — about 2.4x fewer candidates than 27 full-size
// cells — and the outer shell of the 5x5x5 block only if the current 5th-best distance does not rule it out:
// every point outside the inner block is at least `margin` away (margin = one cell + the query's gap to the nearest
// face of its own cell), so worst < 0.999 * margin^2 makes the shell irrelevant.  All bounds are conservative by
// 0.1 % against f32 rounding of the distances; ties (d == worst) never skip.
// Profiling aid, compiled only with -DLILI_PHASE_PROBE (tools/assoc_phases.sh builds such a copy of the library; the product
// build has none of it): s_memrealtime stamps (100 MHz) of one wave at the phase boundaries of the association.  `dep` is a value
// the phase produced, so that the stamp cannot be taken before it exists.
struct PhaseProbe { long long t[8]; };
#ifdef LILI_PHASE_PROBE
#define PHASE_STAMP(pp, k, dep) do { if (pp) { long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); (pp)->t[k] = t_; } } while (0)
#else
#define PHASE_STAMP(pp, k, dep) do {} while (0)
#endif
__device__ __forceinline__ float gate_bound(double gate) {   // smallest f32 >= gate
    float gf = (float)gate;
    if ((double)gf < gate) gf = __uint_as_float(__float_as_uint(gf) + 1u);
    return gf;
}
// the 16 (dy, dz) rows of the 5x5x5 shell, nearest first (faces, then the rows next to them, then the corners), two batches of eight
__device__ constexpr int kShellDy[16] = {0, 0, -2, 2, -1, 1, -1, 1, -2, -2, 2, 2, -2, -2, 2, 2};
__device__ constexpr int kShellDz[16] = {-2, 2, 0, 0, -2, -2, 2, 2, -1, 1, -1, 1, -2, 2, -2, 2};
template <class SEL, class TAB>
__device__ __forceinline__ bool knn5_grid_sel(const GridView& g, TAB& tab, float qx, float qy, float qz, float bound, Top5& best, PhaseProbe* pp = nullptr) {
    SEL sel; sel.init(bound);
    sel.to_top5(best);
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return false;
    if constexpr (std::is_same<SEL, Sel5K>::value) sel.attach(&tab.cj[0][threadIdx.x], (int)(sizeof(tab.cj[0]) / sizeof(int)));
    const int R = g.reach;
    int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
    // queries more than `reach` cells outside the grid cannot have a neighbour within the gate radius
    if (cx < -R || cx > g.nx - 1 + R || cy < -R || cy > g.ny - 1 + R || cz < -R || cz > g.nz - 1 + R) return false;
    // the query's inner block lies in the box that has super-rows (x0 > x1: nothing to search either way)
    const bool inner9 = g.cell_start9 && cy >= g.by0 && cy < g.by0 + g.bny && cz >= g.bz0 && cz < g.bz0 + g.bnz &&
                        max(cx - 1, 0) >= g.bx0 && min(cx + 1, g.nx - 1) < g.bx0 + g.bnx;
    if (inner9) {
        // Super-row layout: the inner 27 cells are ONE run of the unified array (two range words, full chunks, no row table).
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        if (x0 <= x1) {
            const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
            int cj = row[x0];
            const int ce = row[x1 + 1];
            PHASE_STAMP(pp, 2, ce);
            auto fetch = [&](float4& p0, float4& p1, float4& p2, float4& p3, int& pj) {
                // unconditional: no branch around the loads, so the waits the compiler inserts are exact.  Slots past the run's end — also the
                // whole chunk requested after the last one — read the following entries (the array has 8 entries of slack) and are masked by
                // position / never processed.
                pj = cj;
                const float4* q = (const float4*)((const char*)g.pts + ((unsigned)cj << 4));
                p0 = q[0]; p1 = q[1]; p2 = q[2]; p3 = q[3];
                cj += 4;
            };
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            int aj = 0, bj = 0;
            fetch(a0, a1, a2, a3, aj);
            for (;;) {
                if (!(aj < ce)) break;
                fetch(b0, b1, b2, b3, bj);
                process_chunk(sel, a0, a1, a2, a3, aj, ce, qx, qy, qz);
                if (!(bj < ce)) break;
                fetch(a0, a1, a2, a3, aj);
                process_chunk(sel, b0, b1, b2, b3, bj, ce, qx, qy, qz);
            }
        }
    } else {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
        if (x0 <= x1) {
            // All nine row ranges are fetched at once (18 independent loads); the non-empty rows go to this thread's
            // column of the LDS table in visiting order (centre, faces, diagonals) with their lower bounds.
            int rb[9], re[9];
#pragma unroll
            for (int n = 0; n < 9; n++) {
                const int w = row_order(n);
                const int y = cy + w / 3 - 1, z = cz + w % 3 - 1;
                const bool in = z >= 0 && z < g.nz && y >= 0 && y < g.ny;
                const int* cs = g.cell_start + (size_t)(min(max(z, 0), g.nz - 1) * g.ny + min(max(y, 0), g.ny - 1)) * g.nx;
                const int b = cs[x0], e = cs[x1 + 1];
                rb[n] = b; re[n] = in ? e : b;
            }
            const int tid = threadIdx.x;
            int cnt = 0;
#pragma unroll
            for (int n = 0; n < 9; n++) {
                const int w = row_order(n);
                if (rb[n] < re[n]) {
                    tab.b[cnt][tid] = rb[n]; tab.e[cnt][tid] = re[n];
                    tab.lb[cnt][tid] = row_lower_bound(g, qy, qz, cy, cz, w / 3 - 1, w % 3 - 1);
                    cnt++;
                }
            }
            // Each lane walks ITS OWN rows (a wave takes max-over-lanes of the summed trips, not the sum of per-row maxima)
            // and the four loads of the next chunk are issued before the current chunk is processed.  The row for the
            // next chunk is chosen with the 5th-best distance of one chunk ago: a stale (larger) value can only keep a
            // row that the fresh one would prune — extra candidates, never a missing one.
            PHASE_STAMP(pp, 2, cnt);                                        // row ranges loaded, table written
            int n = 0, cj = 0, ce = 0;
            auto fetch = [&](float wv, float4& p0, float4& p1, float4& p2, float4& p3, int& pj, int& pe) {
                while (cj >= ce && n < cnt) {
                    const int b = tab.b[n][tid], e = tab.e[n][tid];
                    const float lb = tab.lb[n][tid];
                    n++;
                    if (!(lb > wv)) { cj = b; ce = e; }
                }
                pj = cj; pe = ce;
                if (cj < ce) {
                    const int last = ce - 1;
                    p0 = load_pt(g, cj); p1 = load_pt(g, min(cj + 1, last)); p2 = load_pt(g, min(cj + 2, last)); p3 = load_pt(g, min(cj + 3, last));
                    cj += 4;
                }
            };
            // two chunk buffers in ping-pong, so that the loads of one are in flight while the other is processed
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            int aj = 0, ae = 0, bj = 0, be = 0;
            fetch(sel.worst(), a0, a1, a2, a3, aj, ae);
            for (;;) {
                if (!(aj < ae)) break;
                fetch(sel.worst(), b0, b1, b2, b3, bj, be);
                process_chunk(sel, a0, a1, a2, a3, aj, ae, qx, qy, qz);
                if (!(bj < be)) break;
                fetch(sel.worst(), a0, a1, a2, a3, aj, ae);
                process_chunk(sel, b0, b1, b2, b3, bj, be, qx, qy, qz);
            }
        }
    }
    PHASE_STAMP(pp, 3, sel.worst());                                        // inner 3x3x3 block walked
    if (R == 2) {
        const double c = g.cell;
        const double fxm = (double)qx - (g.ox + (double)cx * c), fxp = (g.ox + (double)(cx + 1) * c) - (double)qx;
        const double fym = (double)qy - (g.oy + (double)cy * c), fyp = (g.oy + (double)(cy + 1) * c) - (double)qy;
        const double fzm = (double)qz - (g.oz + (double)cz * c), fzp = (g.oz + (double)(cz + 1) * c) - (double)qz;
        const double margin = c + fmax(fmin(fmin(fmin(fxm, fxp), fmin(fym, fyp)), fmin(fzm, fzp)), 0.0);
        if (!(sel.worst() < (float)(0.999 * margin * margin))) {
            // super-row layout: the 18 single-cell runs x = cx -+ 2 of the nine inner rows are two runs (one super cell each)
            const bool side9 = inner9 && (cx - 2 < 0 || cx - 2 >= g.bx0) && (cx + 2 >= g.nx || cx + 2 < g.bx0 + g.bnx);
            if (side9) {
                const int* row = g.cell_start9 + srow_index(g, g.bx0, cy, cz) - g.bx0;
                const int xl = cx - 2, xr = cx + 2;
                if (xl >= 0 && xl < g.nx) { const double gx = fmax(fxm + c, 0.0); if (!((float)(0.999 * gx * gx) > sel.worst())) scan_run(g, sel, row[xl], row[xl + 1], qx, qy, qz); }
                if (xr >= 0 && xr < g.nx) { const double gx = fmax(fxp + c, 0.0); if (!((float)(0.999 * gx * gx) > sel.worst())) scan_run(g, sel, row[xr], row[xr + 1], qx, qy, qz); }
            }
            // How many lanes of the wave are here?  A handful (a converged pose: one lane in a few waves) is bound by the dependent round trips
            // of the 16 shell rows — the batched form below; many (the first iterations of a registration) are bound by instruction issue on
            // mostly idle lanes, where the row-by-row walk with its progressive pruning and x-trimming does less work.
            const bool few = __popcll(__ballot(1)) <= 2;
            if (side9 && few) {
                // The 16 rows of the shell, nearest first, in two batches of eight: the range words of a batch are requested TOGETHER (one
                // round trip instead of eight dependent ones; pruned and x-trimmed with the 5th best at that moment), parked in the lane's
                // columns of the row table (idle in this layout), and only the non-empty rows that still matter are scanned.  The second
                // batch sees the 5th best the first one left.
                const double g1m = fmax(fxm, 0.0), g1p = fmax(fxp, 0.0), g2m = fmax(fxm + c, 0.0), g2p = fmax(fxp + c, 0.0);
                const int tid = threadIdx.x;
#pragma unroll
                for (int batch = 0; batch < 2; batch++) {
                    const float wv = sel.worst();
                    int rb[8], re[8]; float rl[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int dy = kShellDy[batch * 8 + i], dz = kShellDz[batch * 8 + i];
                        const int y = cy + dy, z = cz + dz;
                        const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym + (double)(-dy - 1) * c : fyp + (double)(dy - 1) * c, 0.0);
                        const double gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm + (double)(-dz - 1) * c : fzp + (double)(dz - 1) * c, 0.0);
                        const double lbr = 0.999 * (gy * gy + gz * gz);
                        const int dl = (float)(lbr + 0.999 * g2m * g2m) > wv ? ((float)(lbr + 0.999 * g1m * g1m) > wv ? 0 : 1) : 2;
                        const int dr = (float)(lbr + 0.999 * g2p * g2p) > wv ? ((float)(lbr + 0.999 * g1p * g1p) > wv ? 0 : 1) : 2;
                        const int x0 = max(cx - dl, 0), x1 = min(cx + dr, g.nx - 1);
                        const bool keep = y >= 0 && y < g.ny && z >= 0 && z < g.nz && !((float)lbr > wv) && x0 <= x1;
                        const int* cs = g.cell_start + (size_t)(keep ? z * g.ny + y : 0) * g.nx;
                        const int b = cs[keep ? x0 : 0], e = cs[keep ? x1 + 1 : 0];
                        rb[i] = b; re[i] = keep ? e : b; rl[i] = (float)lbr;
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) { tab.b[i][tid] = rb[i]; tab.e[i][tid] = re[i]; tab.lb[i][tid] = rl[i]; }
                    for (int i = 0; i < 8; i++) {
                        const int b = tab.b[i][tid], e = tab.e[i][tid];
                        if (b < e && !(tab.lb[i][tid] > sel.worst())) scan_run(g, sel, b, e, qx, qy, qz);
                    }
                }
            } else
            for (int dz = -2; dz <= 2; dz++) {
                const int z = cz + dz;
                if (z < 0 || z >= g.nz) continue;
                const double gz = dz == 0 ? 0.0 : fmax(dz < 0 ? fzm + (double)(-dz - 1) * c : fzp + (double)(dz - 1) * c, 0.0);
                for (int dy = -2; dy <= 2; dy++) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.ny) continue;
                    const double gy = dy == 0 ? 0.0 : fmax(dy < 0 ? fym + (double)(-dy - 1) * c : fyp + (double)(dy - 1) * c, 0.0);
                    const double lbr = 0.999 * (gy * gy + gz * gz);
                    if ((float)lbr > sel.worst()) continue;
                    const int* cs = g.cell_start + (size_t)(z * g.ny + y) * g.nx;
                    if (dy == -2 || dy == 2 || dz == -2 || dz == 2) {            // a row of the shell: up to 5 cells,
                        // trimmed to the cells whose box distance (row gap + x gap) can still beat the 5th best
                        const float wv = sel.worst();
                        const double g1m = fmax(fxm, 0.0), g1p = fmax(fxp, 0.0), g2m = fmax(fxm + c, 0.0), g2p = fmax(fxp + c, 0.0);
                        const int dl = (float)(lbr + 0.999 * g2m * g2m) > wv ? ((float)(lbr + 0.999 * g1m * g1m) > wv ? 0 : 1) : 2;
                        const int dr = (float)(lbr + 0.999 * g2p * g2p) > wv ? ((float)(lbr + 0.999 * g1p * g1p) > wv ? 0 : 1) : 2;
                        const int x0 = max(cx - dl, 0), x1 = min(cx + dr, g.nx - 1);
                        if (x0 <= x1) scan_run(g, sel, cs[x0], cs[x1 + 1], qx, qy, qz);
                    } else if (!side9) {                                         // inner row: only its two outer cells are new
                        const int xl = cx - 2, xr = cx + 2;
                        if (xl >= 0 && xl < g.nx) { double gx = fmax(fxm + c, 0.0); if (!((float)(lbr + 0.999 * gx * gx) > sel.worst())) scan_run(g, sel, cs[xl], cs[xl + 1], qx, qy, qz); }
                        if (xr >= 0 && xr < g.nx) { double gx = fmax(fxp + c, 0.0); if (!((float)(lbr + 0.999 * gx * gx) > sel.worst())) scan_run(g, sel, cs[xr], cs[xr + 1], qx, qy, qz); }
                    }
                }
            }
        }
    }
    PHASE_STAMP(pp, 4, sel.worst());                                        // shell decided / walked
    if constexpr (std::is_same<SEL, Sel5K>::value) return sel.finish(g, qx, qy, qz, best);
    else { sel.to_top5(best); return sel.final_tie(); }
}
// Fast selection first; the rare queries with an exact distance tie that could matter are repeated with the exact
// (distance, original index) selector, so the result is always the oracle's.
template <class TAB>
__device__ __forceinline__ void knn5_grid(const GridView& g, TAB& tab, float qx, float qy, float qz, float bound, Top5& best, int dbg = 0, PhaseProbe* pp = nullptr) {
    if (dbg & 32768) { knn5_grid_sel<Sel5>(g, tab, qx, qy, qz, bound, best); return; }   // A/B: exact selector only
    const bool redo = knn5_grid_sel<Sel5K>(g, tab, qx, qy, qz, bound, best, pp);
    PHASE_STAMP(pp, 5, best.d[4]);                                          // five winners resolved (exact distances, order)
    if (redo && !(dbg & 8192)) knn5_grid_sel<Sel5>(g, tab, qx, qy, qz, bound, best);   // bit 8192: profiling only (results then inexact on ties)
}

// Correspondence counting without atomics on a shared word (3128 same-address atomics cost ~40 us on
// MI355X): each block stores its own count; consumers add the <= few-thousand block counts themselves.
template <int BS>
__device__ __forceinline__ void store_block_count(bool ok, int* __restrict__ block_counts, int bid) {
    __shared__ int wave_cnt[BS / 64];
    unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < BS / 64; w++) s += wave_cnt[w];
        block_counts[bid] = s;
    }
}
// Sum of the per-block counts of one association launch (every thread of the block gets the total).
__device__ __forceinline__ int sum_block_counts(const int* __restrict__ block_counts, int nb) {
    __shared__ int part[16];
    __shared__ int total;
    int s = 0;
    // four independent loads per trip (the plain strided loop serialises one L2 round trip per element)
    const int bd = blockDim.x;
    for (int b0 = threadIdx.x; b0 < nb; b0 += 4 * bd) {
        const int b1 = b0 + bd, b2 = b0 + 2 * bd, b3 = b0 + 3 * bd;
        const int v0 = block_counts[b0], v1 = b1 < nb ? block_counts[b1] : 0, v2 = b2 < nb ? block_counts[b2] : 0, v3 = b3 < nb ? block_counts[b3] : 0;
        s += (v0 + v1) + (v2 + v3);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w]; total = t; }
    __syncthreads();
    return total;
}
// A kind without records counts 0.  `out` (optional): a caller-owned int[2] that receives the same totals (multi-GPU
// callers all-reduce it in place).
// `v.seq != 0`: the totals are all-reduced over the ranks inside this launch (lili_p2p_dev.h) — one launch instead of count kernel +
// collective: state / out then hold the GLOBAL counts, identical on every rank.
__global__ __launch_bounds__(kBlock) void k_sum_counts(const int* __restrict__ bc_surf, int nb_surf, const int* __restrict__ bc_edge, int nb_edge,
                                                      SlotState* __restrict__ state, int* __restrict__ out, P2PView v) {
    int t0 = bc_surf ? sum_block_counts(bc_surf, nb_surf) : 0;
    __syncthreads();
    int t1 = bc_edge ? sum_block_counts(bc_edge, nb_edge) : 0;
    if (threadIdx.x >= 64) return;
    if (v.seq) {
        unsigned long long s0, s1;
        const int mine = threadIdx.x == 0 ? t0 : t1;
        if (!p2p_exchange_wave<false>(v, 2, (unsigned long long)(unsigned)mine, 0ull, s0, s1)) { if (threadIdx.x == 0) state->gn_status = 2; return; }
        t0 = (int)(unsigned)__shfl((int)(unsigned)s0, 0); t1 = (int)(unsigned)__shfl((int)(unsigned)s0, 1);
    }
    if (threadIdx.x == 0) {
        state->n_res[0] = t0; state->n_res[1] = t1;
        if (out) { out[0] = t0; out[1] = t1; }
    }
}

__device__ __forceinline__ void load_assoc_pose(const PoseArg& pa, const MatchParams& P, dq& Q2, d3& T2) {
    if (pa.state) {
        const double* s = pa.state->pose;
        dq Q{s[3], s[4], s[5], s[6]};
        d3 T{s[0], s[1], s[2]};
        if (pa.derive_assoc) {   // L/src/BackendFusion.cpp:929-930
            Q2 = qmul(Q, dq{P.q_lb_inv[0], P.q_lb_inv[1], P.q_lb_inv[2], P.q_lb_inv[3]});
            T2 = T - qrot(Q2, d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});
        } else { Q2 = Q; T2 = T; }
    } else {
        Q2 = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]};
        T2 = d3{pa.t[0], pa.t[1], pa.t[2]};
    }
}

// ================================================================================================
// K5 / K6 — association.  The per-query fits are shared by the tiled (LDS) and the direct search path.
// surf records: rec_nd[i] = (w*nx, w*ny, w*nz, w*normInverse) as floats, rec_score[i] (f64), valid[i]
// edge records: rec_a[i] = (Ax, Ay, Az, s), rec_b[i] = (Bx, By, Bz, 0), valid[i]
// ================================================================================================
__device__ __forceinline__ void store_debug_nn(const GridView& g, const Top5& nn, int i, int* __restrict__ dbg_idx, float* __restrict__ dbg_d2) {
    if (!dbg_idx) return;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        dbg_idx[(size_t)i * 5 + k] = nn.j[k] >= 0 ? __float_as_int(g.pts[nn.j[k]].w) : -1;
        dbg_d2[(size_t)i * 5 + k] = nn.d[k];
    }
}

// Search bound of one query: the reference's gate, tightened by the query's 5 neighbours of the previous association
// of the same scan against the same map index (positions in the cell-sorted array, -1 = none).  Those are five real
// map points, so the true 5th-nearest distance cannot exceed their largest distance w at the new pose; everything
// farther is irrelevant and rows / shell cells beyond it are pruned from the first candidate on.  The result is the
// same exact 5-NN for any pose change — the cache only makes the bound tight when the pose moved little.
__device__ __forceinline__ float seeded_bound(const GridView& g, double gate, const int* __restrict__ nn_cache, int n_q, int i,
                                              float px, float py, float pz) {
    float bound = gate_bound(gate);
    if (nn_cache) {
        int c0 = nn_cache[i], c1 = nn_cache[(size_t)n_q + i], c2 = nn_cache[(size_t)2 * n_q + i], c3 = nn_cache[(size_t)3 * n_q + i],
            c4 = nn_cache[(size_t)4 * n_q + i];
        if ((c0 | c1 | c2 | c3 | c4) >= 0) {
            float w = dist2(load_pt(g, c0), px, py, pz);
            w = fmaxf(w, dist2(load_pt(g, c1), px, py, pz));
            w = fmaxf(w, dist2(load_pt(g, c2), px, py, pz));
            w = fmaxf(w, dist2(load_pt(g, c3), px, py, pz));
            w = fmaxf(w, dist2(load_pt(g, c4), px, py, pz));
            if (w < bound) bound = __uint_as_float(__float_as_uint(w) + 1u);   // strictly above w: the five seeds themselves must enter
        }
    }
    return bound;
}
__device__ __forceinline__ void store_nn_cache(int* __restrict__ nn_cache, int n_q, int i, const Top5& nn) {
    if (!nn_cache) return;
#pragma unroll
    for (int k = 0; k < 5; k++) nn_cache[(size_t)k * n_q + i] = nn.j[k];
}

// findCorrespondingSurfFeatures body after the kNN (L/src/BackendFusion.cpp:1613-1679 and variants)
__device__ __forceinline__ bool surf_fit(const GridView& g, const MatchParams& P, const Top5& nn, float4 ql, float px, float py, float pz,
                                         float4& rn, double& score) {
    rn = make_float4(0.f, 0.f, 0.f, 0.f);
    score = 0.0;
    if ((P.debug & 1) && nn.j[4] >= 0) { rn.x = nn.d[4]; return nn.d[4] < 0.5f; }
    if (!(nn.j[4] >= 0 && (double)nn.d[4] < P.kd_max_radius)) return false;   // L:1615
    float4 m[5];
#pragma unroll
    for (int k = 0; k < 5; k++) m[k] = nn.p[k];
    if (__any(!nn.have)) {           // exact-selector / tiled / debug paths: the points were not handed over
#pragma unroll
        for (int k = 0; k < 5; k++) if (!nn.have) m[k] = g.pts[nn.j[k]];
    }
    double sum_w = 0.0;
    double mx[5], my[5], mz[5], wk[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { mx[k] = (double)m[k].x; my[k] = (double)m[k].y; mz[k] = (double)m[k].z; wk[k] = 1.0; }
    if (P.variant == 0) {   // Livox reflectivity weighting, L:1617-1638
        double w[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            float diff = ql.w - g.aux[nn.j[k]];
            double tmp_w = (double)fabsf(diff);
            sum_w += tmp_w;
            w[k] = 1.0 / tmp_w;
        }
        if (sum_w > P.reflect_thres) return false;
#pragma unroll
        for (int k = 0; k < 5; k++) wk[k] = w[k] / sum_w;
    }
    double nv[3];
    bool fitted = false;
    if (!(P.debug & 16384)) {   // LILI_DEBUG bit 16384: always take the pivoted QR (A/B and parity of the two paths)
        if (P.variant == 0) {
            double w2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) w2[k] = wk[k] * wk[k];
            fitted = plane_fit_centered<true>(mx, my, mz, w2, nv, (P.debug & 2048) ? 0.0 : 1e-7);
        } else fitted = plane_fit_centered<false>(mx, my, mz, wk, nv, (P.debug & 2048) ? 0.0 : 1e-7);
    }
    if (!fitted) {            // ill-conditioned or rank-deficient: Eigen's rank-revealing procedure (rare, wave-divergent)
        col5 c0, c1, c2, b;
#pragma unroll
        for (int k = 0; k < 5; k++) { c0.v[k] = wk[k] * mx[k]; c1.v[k] = wk[k] * my[k]; c2.v[k] = wk[k] * mz[k]; b.v[k] = -1.0 * wk[k]; }
        lstsq53(c0, c1, c2, b, nv);
    }
    double nn_ = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
    double normInverse = 1.0 / nn_;
    nv[0] *= normInverse; nv[1] *= normInverse; nv[2] *= normInverse;
    bool planeValid = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (fabs(nv[0] * (double)m[k].x + nv[1] * (double)m[k].y + nv[2] * (double)m[k].z + normInverse) > P.surf_dist_thres) planeValid = false;
    }
    if (!planeValid) return false;
    // L:1661-1662: float pd, float weight; sqrt(sqrt()) on a float argument is the float overload
    float pd = (float)(nv[0] * (double)px + nv[1] * (double)py + nv[2] * (double)pz + normInverse);
    float r2 = px * px + py * py + pz * pz;
    float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)sqrtf(sqrtf(r2)));
    if (!((double)weight > P.surf_weight_min)) return false;
    rn.x = (float)((double)weight * nv[0]); rn.y = (float)((double)weight * nv[1]); rn.z = (float)((double)weight * nv[2]);
    rn.w = (float)((double)weight * normInverse);
    if (P.variant == 0) score = P.lidar_const * ((double)weight + exp(-sum_w));   // L:1676
    else if (P.variant == 1) score = P.lidar_const * (double)weight;                // R:1515
    else score = 1.0;
    return true;
}

// findCorrespondingCornerFeatures body after the kNN (L/src/BackendFusion.cpp:1543-1596, R:1404-1458)
__device__ __forceinline__ bool edge_fit(const GridView& g, const MatchParams& P, const Top5& nn, float px, float py, float pz,
                                         float4& ra, float4& rb) {
    ra = make_float4(0.f, 0.f, 0.f, 0.f); rb = ra;
    if (!(nn.j[4] >= 0 && (double)nn.d[4] < P.edge_gate)) return false;   // L:1543
    d3 m[5]; d3 c{0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) { float4 p = nn.p[k]; if (!nn.have) p = g.pts[nn.j[k]]; m[k] = d3{(double)p.x, (double)p.y, (double)p.z}; c = c + m[k]; }
    c = d3{c.x / 5.0, c.y / 5.0, c.z / 5.0};
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        d3 z = m[k] - c;
        a00 += z.x * z.x; a01 += z.x * z.y; a02 += z.x * z.z; a11 += z.y * z.y; a12 += z.y * z.z; a22 += z.z * z.z;
    }
    double ev[3]; d3 vmin, vmax;
    eig3_sym(a00, a01, a02, a11, a12, a22, ev, vmin, vmax);
    if (!(ev[2] > 3.0 * ev[1])) return false;   // L:1575
    d3 u = canon_sign(vmax);
    d3 A = c + 0.1 * u, B = c - 0.1 * u;
    if (P.edge_dist_max > 0) {   // R:1437-1443
        d3 lp{(double)px, (double)py, (double)pz};
        d3 nu = cross3(lp - A, lp - B);
        d3 de = A - B;
        double dist = sqrt(dot3(nu, nu)) / sqrt(dot3(de, de));
        if (!(dist < P.edge_dist_max)) return false;
    }
    ra = make_float4((float)A.x, (float)A.y, (float)A.z, (float)P.lidar_const);
    rb = make_float4((float)B.x, (float)B.y, (float)B.z, 0.f);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Tiled exact 5-NN: the 256 queries of a block (consecutive in the binned order, hence spatially
// compact) share one neighbourhood = bounding box of their cells +-1.  Each (y,z) row of that box is ONE
// contiguous run of the cell-sorted map (x-fastest cells), so the block stages the rows in LDS with
// coalesced 16-B loads and every thread then scans its own 3x3x3-cell window from LDS instead of
// issuing ~100 dependent, divergent global loads.  Rows are staged in batches of <= kTileCap points;
// degenerate tiles (too many rows / batches) use the direct path.  The candidate set per query is exactly
// the 27-cell set of knn5_grid, so results are identical.
// ------------------------------------------------------------------------------------------------
constexpr int kTileCap = 2048;        // float4 slots staged per batch (32 KiB)
constexpr int kTileHalf = kTileCap / 2;
constexpr int kTileRows = kBlock;     // one row descriptor per thread
constexpr int kTileBatches = 32;

struct TileLds {
    float4 pts[kTileCap];
    int row_gbeg[kTileRows];
    int row_len[kTileRows];
    int row_off[kTileRows];
    int batch_base[kTileBatches];
    int bbox[6];
    int scan[kBlock / 64 + 1];
};

template <class TAB>
__device__ __forceinline__ void knn5_tiled(const GridView& g, TileLds& L, TAB& tab, bool live, float qx, float qy, float qz, Top5& best, int dbg) {
    const int tid = threadIdx.x;
    Sel5 sel; sel.init();
    sel.to_top5(best);
    // ---- phase 0: cell of every query, block bounding box
    int cx = 0, cy = 0, cz = 0;
    bool inr = false;
    if (live && isfinite(qx) && isfinite(qy) && isfinite(qz)) {
        cx = cell_coord(qx, g.ox, g.inv_cell); cy = cell_coord(qy, g.oy, g.inv_cell); cz = cell_coord(qz, g.oz, g.inv_cell);
        inr = !(cx < -1 || cx > g.nx || cy < -1 || cy > g.ny || cz < -1 || cz > g.nz);
    }
    if (tid < 3) { L.bbox[tid] = 0x7fffffff; L.bbox[3 + tid] = -0x7fffffff; }
    if (tid < kTileBatches) L.batch_base[tid] = 0x7fffffff;
    __syncthreads();
    {
        int mnx = inr ? cx : 0x7fffffff, mny = inr ? cy : 0x7fffffff, mnz = inr ? cz : 0x7fffffff;
        int mxx = inr ? cx : -0x7fffffff, mxy = inr ? cy : -0x7fffffff, mxz = inr ? cz : -0x7fffffff;
        for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o)); mnz = min(mnz, __shfl_xor(mnz, o));
            mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o)); mxz = max(mxz, __shfl_xor(mxz, o));
        }
        if ((tid & 63) == 0) {
            atomicMin(&L.bbox[0], mnx); atomicMin(&L.bbox[1], mny); atomicMin(&L.bbox[2], mnz);
            atomicMax(&L.bbox[3], mxx); atomicMax(&L.bbox[4], mxy); atomicMax(&L.bbox[5], mxz);
        }
    }
    __syncthreads();
    if (L.bbox[0] > L.bbox[3]) return;   // no query of this block can have a neighbour (uniform exit)
    const int xa = max(L.bbox[0] - 1, 0), xb = min(L.bbox[3] + 1, g.nx - 1);
    const int ya = max(L.bbox[1] - 1, 0), yb = min(L.bbox[4] + 1, g.ny - 1);
    const int za = max(L.bbox[2] - 1, 0), zb = min(L.bbox[5] + 1, g.nz - 1);
    const int nyt = yb - ya + 1, nzt = zb - za + 1;
    const long long nrows_ll = (long long)nyt * (long long)nzt;
    bool direct = xa > xb || nyt <= 0 || nzt <= 0 || nrows_ll > kTileRows;
    int total = 0;
    if (!direct) {
        // ---- phase 1: row descriptors (one per thread) + exclusive scan of the row lengths
        const int nrows = (int)nrows_ll;
        int len = 0, gbeg = 0;
        if (tid < nrows) {
            int y = ya + tid / nzt, z = za + tid % nzt;
            int row = (z * g.ny + y) * g.nx;
            gbeg = g.cell_start[row + xa];
            len = g.cell_start[row + xb + 1] - gbeg;
        }
        int off = block_exclusive_scan(len, L.scan, total);
        if (tid < nrows) {
            L.row_gbeg[tid] = gbeg; L.row_len[tid] = len; L.row_off[tid] = off;
            if (len > 0 && len <= kTileHalf) {
                int bq = (off + len - 1) / kTileHalf;
                if (bq < kTileBatches) atomicMin(&L.batch_base[bq], off);
            }
        }
        if ((total + kTileHalf - 1) / kTileHalf > kTileBatches) direct = true;   // uniform: `total` is block-wide
        __syncthreads();
    }
    if (direct) {   // degenerate tile: per-thread search in global memory (identical candidate set)
        if (inr) {
            Top5 t; knn5_grid(g, tab, qx, qy, qz, __uint_as_float(0x7f800000u), t);
            best = t;
        }
        return;
    }
    if (dbg & 16) return;
    // ---- phase 2: this thread's 9 windows (global index ranges) and the tile rows they live in
    int wbeg[9], wend[9], wrow[9];
    float wlb[9];
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
#pragma unroll
    for (int n = 0; n < 9; n++) {
        const int w = n;   // slot n holds row row_order(n): centre first, so later rows can be pruned
        const int ro = row_order(n);
        int y = cy + (ro / 3 - 1), z = cz + (ro % 3 - 1);
        wlb[n] = row_lower_bound(g, qy, qz, cy, cz, ro / 3 - 1, ro % 3 - 1);
        bool okw = inr && x0 <= x1 && y >= 0 && y < g.ny && z >= 0 && z < g.nz;
        wrow[w] = okw ? (y - ya) * nzt + (z - za) : -1;
        int row = (z * g.ny + y) * g.nx;
        wbeg[w] = okw ? g.cell_start[row + x0] : 0;
        wend[w] = okw ? g.cell_start[row + x1 + 1] : 0;
    }
    if (dbg & 32) return;
    // ---- phase 3: batches of rows through LDS
    const int nrows = (int)nrows_ll;
    const int nbatch = (total + kTileHalf - 1) / kTileHalf;
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < nbatch; k++) {
        const int base = L.batch_base[k];
        if (base != 0x7fffffff && !(dbg & 8)) {
            for (int r = wave; r < nrows; r += kBlock / 64) {
                int len = L.row_len[r];
                if (len <= 0 || len > kTileHalf) continue;
                int off = L.row_off[r];
                if ((off + len - 1) / kTileHalf != k) continue;
                const float4* src = g.pts + L.row_gbeg[r];
                float4* dst = L.pts + (off - base);
                for (int l = lane; l < len; l += 64) dst[l] = src[l];
            }
        }
        __syncthreads();
        if (base != 0x7fffffff && !(dbg & 4)) {
#pragma unroll
            for (int w = 0; w < 9; w++) {
                int r = wrow[w];
                if (r < 0 || wbeg[w] >= wend[w]) continue;
                int len = L.row_len[r], off = L.row_off[r];
                if (len > kTileHalf || (off + len - 1) / kTileHalf != k) continue;
                if (wlb[w] > sel.worst()) continue;
                const int shift = (off - base) - L.row_gbeg[r];   // LDS slot = global index + shift
                int j = wbeg[w];
                for (; j + 1 < wend[w]; j += 2) {
                    float4 p0 = L.pts[j + shift], p1 = L.pts[j + 1 + shift];
                    float d0 = dist2(p0, qx, qy, qz), d1 = dist2(p1, qx, qy, qz);
                    sel.insert(d0, p0, j);
                    sel.insert(d1, p1, j + 1);
                }
                if (j < wend[w]) { float4 p0 = L.pts[j + shift]; sel.insert(dist2(p0, qx, qy, qz), p0, j); }
            }
        }
        __syncthreads();
    }
    // ---- phase 4: rows longer than half a batch are scanned straight from global memory
#pragma unroll
    for (int w = 0; w < 9; w++) {
        int r = wrow[w];
        if (r < 0 || wbeg[w] >= wend[w]) continue;
        if (L.row_len[r] <= kTileHalf) continue;
        if (wlb[w] > sel.worst()) continue;
        for (int j = wbeg[w]; j < wend[w]; j++) {
            float4 p = g.pts[j];
            sel.insert(dist2(p, qx, qy, qz), p, j);
        }
    }
    sel.to_top5(best);
}

// What a lane of the association found, handed on in registers to the launch that linearises on the fly (k_associate_lin): the values
// are the ROUNDED ones the record arrays receive, so the two-launch path sees the same numbers.
struct LaneRec { bool ok; float4 ql, r0, r1; double score; };
template <bool TILED, int BS, class TILE, class TAB>
__device__ __forceinline__ void assoc_surf_body(
        const float4* __restrict__ queries, const int* __restrict__ perm, const int2* __restrict__ tiles, int n_q, const GridView& g, const PoseArg& pa, const MatchParams& P,
        float4* __restrict__ rec_nd, double* __restrict__ rec_score, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int* __restrict__ nn_cache, const AssocSched& sched,
        int vbid, TILE& L, TAB& tab, LaneRec* rec_out = nullptr) {
    const long long t_begin = (P.debug & 4096) ? (long long)__builtin_amdgcn_s_memrealtime() : 0ll;   // profiling aid (tools/assoc_blocks.py)
    const int bid = sched.order ? sched.order[vbid] : vbid;
    const int2 tile = tiles ? tiles[bid] : make_int2(bid * BS, min(BS, n_q - bid * BS));
    const bool live = (int)threadIdx.x < tile.y;
    int t = tile.x + threadIdx.x;
    int i = live ? (perm ? perm[t] : t) : 0;
    float4 ql = queries[live ? i : 0];     // requested before the (dependent, scalar) pose loads: the two latencies overlap
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
    if (P.debug & 2) {
#pragma unroll
        for (int k = 0; k < 5; k++) { nn.d[k] = 0.01f * (k + 1); nn.j[k] = (i * 7 + k) % g.n_points; }
    } else if constexpr (TILED) knn5_tiled(g, L, tab, live, px, py, pz, nn, P.debug);
    else if (live) { knn5_grid(g, tab, px, py, pz, seeded_bound(g, P.kd_max_radius, nn_cache, n_q, i, px, py, pz), nn, P.debug, pp); store_nn_cache(nn_cache, n_q, i, nn); }
    if (BS == 64 && !TILED && sched.block_cost) {   // cost of this block for the next launch's dispatch order
        int c = live ? nn.aux : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o));
        if (threadIdx.x == 0) sched.block_cost[bid] = c;
    }
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
    store_block_count<BS>(ok, block_counts, vbid);
#ifdef LILI_PHASE_PROBE
    if ((P.debug & 4096) && dbg_d2 && threadIdx.x == 0) {   // phase stamps as ticks since the block began, over the d2 debug rows of the block's first two queries
        for (int k = 1; k <= 6; k++) dbg_d2[(size_t)tile.x * 5 + k] = probe.t[k] ? (float)(probe.t[k] - t_begin) : -1.0f;
    }
#endif
    if ((P.debug & 4096) && dbg_idx && threadIdx.x == 0) {   // per-workgroup begin / end ticks (100 MHz) and hardware id into the debug rows of the block's first query
        long long* o = (long long*)(dbg_idx + (size_t)tile.x * 5);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = t_begin; o[1] = (long long)__builtin_amdgcn_s_memrealtime();
        dbg_idx[(size_t)tile.x * 5 + 4] = (int)((hw & 0xffffu) | (xcc << 16));
    }
    if ((P.debug & 4096) && dbg_idx && live && threadIdx.x != 0) dbg_idx[(size_t)i * 5] = nn.aux;
}
struct NoTile {};
template <bool TILED, int BS>
__global__ __launch_bounds__(BS) void k_associate_surf(
        const float4* __restrict__ queries, const int* __restrict__ perm, const int2* __restrict__ tiles, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_nd, double* __restrict__ rec_score, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int* __restrict__ nn_cache, AssocSched sched) {
    __shared__ typename std::conditional<TILED, TileLds, NoTile>::type L;
    __shared__ RowTabT<BS> tab;
    assoc_surf_body<TILED, BS>(queries, perm, tiles, n_q, g, pa, P, rec_nd, rec_score, valid, dbg_idx, dbg_d2, block_counts, nn_cache, sched, (int)blockIdx.x, L, tab);
}

template <bool TILED, int BS, class TILE, class TAB>
__device__ __forceinline__ void assoc_edge_body(
        const float4* __restrict__ queries, const int* __restrict__ perm, const int2* __restrict__ tiles, int n_q, const GridView& g, const PoseArg& pa, const MatchParams& P,
        float4* __restrict__ rec_a, float4* __restrict__ rec_b, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int* __restrict__ nn_cache, const AssocSched& sched,
        int vbid, TILE& L, TAB& tab, LaneRec* rec_out = nullptr) {
    const int bid = sched.order ? sched.order[vbid] : vbid;
    const int2 tile = tiles ? tiles[bid] : make_int2(bid * BS, min(BS, n_q - bid * BS));
    const bool live = (int)threadIdx.x < tile.y;
    int t = tile.x + threadIdx.x;
    int i = live ? (perm ? perm[t] : t) : 0;
    float4 ql = queries[live ? i : 0];     // requested before the (dependent, scalar) pose loads: the two latencies overlap
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
    float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    Top5 nn; nn.have = false;
    if constexpr (TILED) knn5_tiled(g, L, tab, live, px, py, pz, nn, P.debug);
    else if (live) { knn5_grid(g, tab, px, py, pz, seeded_bound(g, P.edge_gate, nn_cache, n_q, i, px, py, pz), nn); store_nn_cache(nn_cache, n_q, i, nn); }
    bool ok = false;
    if (live) {
        store_debug_nn(g, nn, i, dbg_idx, dbg_d2);
        float4 ra, rb;
        ok = edge_fit(g, P, nn, px, py, pz, ra, rb);
        rec_a[i] = ra; rec_b[i] = rb; valid[i] = ok ? 1 : 0;
        if (rec_out) { rec_out->ql = ql; rec_out->r0 = ra; rec_out->r1 = rb; }
    }
    if (rec_out) rec_out->ok = ok;
    store_block_count<BS>(ok, block_counts, vbid);
}
template <bool TILED, int BS>
__global__ __launch_bounds__(BS) void k_associate_edge(
        const float4* __restrict__ queries, const int* __restrict__ perm, const int2* __restrict__ tiles, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_a, float4* __restrict__ rec_b, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts, int* __restrict__ nn_cache, AssocSched sched) {
    __shared__ typename std::conditional<TILED, TileLds, NoTile>::type L;
    __shared__ RowTabT<BS> tab;
    assoc_edge_body<TILED, BS>(queries, perm, tiles, n_q, g, pa, P, rec_a, rec_b, valid, dbg_idx, dbg_d2, block_counts, nn_cache, sched, (int)blockIdx.x, L, tab);
}

// Both kinds of one keyframe in ONE launch (the reference back-end associates corners and planes of a keyframe back to back,
// L/src/BackendFusion.cpp:935-936): workgroups [0, E.nb) take the (few) edge queries, the rest the surf queries — one wave per
// workgroup, direct search path.  Saves a kernel boundary and the fill / drain of a second grid per outer iteration.
__global__ __launch_bounds__(kAssocBlock) void k_associate_both(AssocArgs S, AssocArgs E, PoseArg pa, MatchParams P) {
    __shared__ NoTile L;
    __shared__ RowTabT<kAssocBlock> tab;
    const AssocSched sched{nullptr, nullptr};
    const int b = (int)blockIdx.x;
    if (b < E.nb) assoc_edge_body<false, kAssocBlock>(E.queries, nullptr, nullptr, E.n_q, E.g, pa, P, E.rec0, reinterpret_cast<float4*>(E.rec1), E.valid, E.dbg_idx, E.dbg_d2,
                                                      E.block_counts, E.nn_cache, sched, b, L, tab);
    else assoc_surf_body<false, kAssocBlock>(S.queries, nullptr, nullptr, S.n_q, S.g, pa, P, S.rec0, reinterpret_cast<double*>(S.rec1), S.valid, S.dbg_idx, S.dbg_d2,
                                             S.block_counts, S.nn_cache, sched, b - E.nb, L, tab);
}
// Association on a map that is much denser than the gate radius (SURVEY §8d Config 2, variant B: 5 M points at a 0.05 m leaf — ~170
// points per gate-sized cell, ~1500 candidates in the inner 27 cells).  The map then carries a SECOND index with cells sized from the
// measured point density (lili_map_set: ~3 points per cell), searched first with the selection bounded by fbound = (reach * fine cell /
// 1.01)^2: everything within sqrt(fbound) of the query lies inside the fine 5x5x5 block, so if five points are found strictly inside
// fbound no unseen point can beat the fifth — they ARE the global 5 nearest.  Only queries that do not find five (map borders, holes)
// repeat the search on the gate-sized index with the reference's gate; both searches are the exact knn5_grid.  `kind`: 0 surf, 1 edge.
__global__ __launch_bounds__(kAssocBlock) void k_associate_fine(AssocArgs A, GridView gf, float fbound, int kind, PoseArg pa, MatchParams P) {
    __shared__ RowTabT<kAssocBlock> tab;
    const int i = (int)blockIdx.x * kAssocBlock + (int)threadIdx.x;
    const bool live = i < A.n_q;
    const float4 ql = A.queries[live ? i : 0];
    dq Q2; d3 T2;
    load_assoc_pose(pa, P, Q2, T2);
    const d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
    const float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
    const float gate = gate_bound(kind == 0 ? P.kd_max_radius : P.edge_gate);
    Top5 nn; nn.aux = 0; nn.have = false;
    bool fine_hit = false;
    if (live) {
        knn5_grid(gf, tab, px, py, pz, fminf(fbound, gate), nn, P.debug);
        fine_hit = nn.j[4] >= 0 && nn.d[4] < fbound;
    }
    if (live && !fine_hit) knn5_grid(A.g, tab, px, py, pz, gate, nn, P.debug);
    bool ok = false;
    if (live) {
        const GridView& u = fine_hit ? gf : A.g;
        store_debug_nn(u, nn, i, A.dbg_idx, A.dbg_d2);
        if (kind == 0) {
            float4 rn; double score;
            ok = surf_fit(u, P, nn, ql, px, py, pz, rn, score);
            A.rec0[i] = rn; reinterpret_cast<double*>(A.rec1)[i] = score; A.valid[i] = ok ? 1 : 0;
        } else {
            float4 ra, rb;
            ok = edge_fit(u, P, nn, px, py, pz, ra, rb);
            A.rec0[i] = ra; reinterpret_cast<float4*>(A.rec1)[i] = rb; A.valid[i] = ok ? 1 : 0;
        }
    }
    store_block_count<kAssocBlock>(ok, A.block_counts, (int)blockIdx.x);
}
#define LILI_ASSOC_ARGS_SURF const float4*, const int*, const int2*, int, GridView, PoseArg, MatchParams, float4*, double*, unsigned char*, int*, float*, int*, int*, AssocSched
#define LILI_ASSOC_ARGS_EDGE const float4*, const int*, const int2*, int, GridView, PoseArg, MatchParams, float4*, float4*, unsigned char*, int*, float*, int*, int*, AssocSched
template __global__ void k_associate_surf<true, kBlock>(LILI_ASSOC_ARGS_SURF);
template __global__ void k_associate_surf<false, kBlock>(LILI_ASSOC_ARGS_SURF);
template __global__ void k_associate_surf<false, kAssocBlock>(LILI_ASSOC_ARGS_SURF);
template __global__ void k_associate_edge<true, kBlock>(LILI_ASSOC_ARGS_EDGE);
template __global__ void k_associate_edge<false, kBlock>(LILI_ASSOC_ARGS_EDGE);
template __global__ void k_associate_edge<false, kAssocBlock>(LILI_ASSOC_ARGS_EDGE);

// Dispatch order for the next association launch of the same scan (see AssocSched).  One workgroup: bitonic sort of
// (cost descending, block index) keys in LDS, then the assignment
//   * the r = n mod S SIMDs that receive one wave more than the others (slots s, S+s, ... with s < r) take the (q+1) r
//     LIGHTEST blocks (q = n / S),
//   * the other S - r SIMDs take q blocks each from the rest in snake order (heaviest with lightest).
// n <= kMaxOrderBlocks.  Cost model: a block's work ~ kCostBase + its chunk count (skeleton + fit + chunks).
constexpr int kMaxOrderBlocks = 8192;
__global__ __launch_bounds__(1024) void k_block_order(const int* __restrict__ block_cost, int n, int S, int* __restrict__ order) {
    __shared__ unsigned key[kMaxOrderBlocks];
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += blockDim.x)
        key[i] = i < n ? ((unsigned)(255 - min(max(block_cost[i], 0), 255)) << 16) | (unsigned)i : 0xffffffffu;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned a = key[i], b = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[l] = a; }
                }
            }
            __syncthreads();
        }
    const int q = n / S, r = n % S;
    const int nl = (q + 1) * r, nh = n - nl;          // light blocks: ranks nh .. n-1 of the descending order
    for (int rank = threadIdx.x; rank < n; rank += blockDim.x) {
        const int blk = (int)(key[rank] & 0xffffu);
        int slot;
        if (rank >= nh) { const int t = rank - nh; slot = (t / r) * S + (t % r); }
        else {
            const int w = S - r, tier = rank / w;
            int pos = rank % w;
            if (tier & 1) pos = w - 1 - pos;
            slot = tier * S + r + pos;
        }
        order[slot] = blk;
    }
}

// ================================================================================================
// Linearisation: residual + 1x7 global Jacobian per record, loss corrector, Gram reduction.
//
// Reduction scheme (deterministic, no float atomics): every wave stages its 64 rows [J0..J6, r, cost]
// in LDS; lane l < 36 owns Gram entry (a,b) of the upper triangle and sums row[q][a]*row[q][b] over
// q = 0..63 in order; lane 36 sums the cost column.  Waves of a block are then added in order and the
// block writes one 40-double partial; k_reduce_gn adds the partials in a fixed order.
// ================================================================================================
constexpr int kRow = 12;          // LDS row: [J0..J6, r | 1, cost, 0, 0]
constexpr int kLinBlock = 1024;   // linearisation block (16 waves; the launch covers the queries with <= 256 blocks)
// Gram accumulation on the f64 matrix cores.  Per wave, G += V^T V over its 64 rows v = [a | b | e] with a = (J0..J3),
// b = (J4, J5, J6, r), e = (1, cost, 0, 0), issued as 16 x v_mfma_f64_4x4x4_4b_f64: ONE instruction contracts four rows (k)
// into four independent 4x4 blocks — block 0: a a^T, block 1: a b^T, block 2: b b^T, block 3: e e^T (count and cost sum) —
// i.e. exactly the 36 + 2 numbers of the upper triangle, where round 1's 16x16x4 form computed a 16x16 tile of which 55
// entries were used (measured on MI355X, tools/probe_mfma.hip: 64 clocks per 16x16x4 against 20 per 4x4x4_4b, and the
// 16 operand reads of a wave were issued one by one in front of their MFMA).  Lane map of the instruction (probed, same file):
// operand lane l feeds row i (A) / column j (B) = l & 3 of block (l >> 2) & 3 at k = l >> 4; result lane o holds
// D_block[o >> 4][o & 3] of block (o >> 2) & 3.  This is a reduction, not a GEMM re-shaping of the path; the summation
// order is fixed by the instruction sequence, so results stay deterministic.
// profiling aid: thread 0 of one probe block stamps the constant 100 MHz clock into SlotState::tprof[slot]
__device__ __forceinline__ void tstamp(const SlotState* state, int debug, int probe_block, int slot) {
    if ((debug & 256) && (int)blockIdx.x == probe_block && threadIdx.x == 0)
        const_cast<SlotState*>(state)->tprof[slot] = (long long)__builtin_amdgcn_s_memrealtime();
}
// 16-byte granule {value, value ^ key}: one write-through store / two relaxed agent-scope (sc1) loads
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long launch_key(unsigned long long epoch) { return (epoch + 1ull) * 0x9E3779B97F4A7C15ull; }   // never 0 for epoch < 2^64 - 1
__device__ __forceinline__ void store_granule(double* g, double v, unsigned long long key) {
    const unsigned long long lo = (unsigned long long)__double_as_longlong(v), hi = lo ^ key;
    const u32x4 d = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g), "v"(d) : "memory");
}
__device__ __forceinline__ void load_granule(const double* g, unsigned long long& lo, unsigned long long& hi) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(g);
    lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// result lane of partial entry e: e < 36 = upper triangle of the 8x8 Gram (row-major), 36 = cost, 37 = count, 38 / 39 = always zero
__device__ __forceinline__ int gram_lane(int e) {
    if (e == 36) return 16 + 12;        // block 3, [1][0] = sum cost * 1
    if (e == 37) return 12;             // block 3, [0][0] = sum 1 * 1
    if (e >= 38) return 2 * 16 + 12 + 2;   // block 3, [2][2] = 0
    int a = 0, l = e;
    while (l >= 8 - a) { l -= 8 - a; a++; }
    const int b = a + l;
    if (b < 4) return 16 * a + b;                       // a a^T
    if (a < 4) return 16 * a + 4 + (b - 4);             // a b^T
    return 16 * (a - 4) + 8 + (b - 4);                  // b b^T
}
struct GramAcc {
    double acc;
    __device__ __forceinline__ void init() { acc = 0.0; }
    // Every wave stages and consumes ITS OWN 64 rows, so only wave-level ordering is needed here (LDS operations of
    // one wave execute in order; the fences keep the compiler from moving them) — no block barrier: fast waves do
    // their MFMAs while slow ones still wait for their records.  All lanes of the wave must call this.
    __device__ __forceinline__ void add_rows(const double Jr[8], double cost, bool ok, double* lds) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double* rows = lds + wave * 64 * kRow;
        double* myrow = rows + lane * kRow;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous tile's reads are done before the rows are overwritten
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; k++) myrow[k] = ok ? Jr[k] : 0.0;
        myrow[8] = ok ? 1.0 : 0.0;
        myrow[9] = ok ? cost : 0.0;
        myrow[10] = 0.0; myrow[11] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int kq = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3;
        const int ia = blk < 2 ? c : (blk == 2 ? 4 + c : 8 + c);
        const int ib = blk == 0 ? c : (blk == 3 ? 8 + c : 4 + c);
        const double* pa = rows + kq * kRow + ia;
        const double* pb = rows + kq * kRow + ib;
        double a[16], b[16];
#pragma unroll
        for (int s = 0; s < 16; s++) { a[s] = pa[4 * s * kRow]; b[s] = pb[4 * s * kRow]; }   // all 32 operand reads in flight before the first MFMA
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[s], b[s], acc, 0, 0, 0);
    }
    // block partial: 36 upper-triangle entries of the 8x8 Gram, [36] = cost, [37] = count
    // `key` != 0: the partial is PUBLISHED for the reducer block of the same launch (fused_tail) as 40 granules of 16 bytes,
    // {value bits, value bits ^ key}, each written by ONE write-through (sc1) 16-byte store.  The data is its own flag: a granule
    // whose halves satisfy hi == lo ^ key was written by THIS launch (key is unique per launch), so the reducer needs no ticket, no
    // fence and no drained-store wait on the producer side (MI355X_MICROARCH.md, inter-workgroup visibility, form R2).
    __device__ __forceinline__ void finish(double* lds, double* slot, unsigned long long key = 0ull) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();                 // every wave is done with its row area before the LDS is reused for the wave results
        lds[wave * 64 + lane] = acc;     // this wave's four 4x4 result blocks
        __syncthreads();
        if (threadIdx.x < 40) {
            const int src = gram_lane(threadIdx.x);
            double s = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += lds[w * 64 + src];
            if (key) store_granule(slot + 2 * threadIdx.x, s, key);
            else slot[threadIdx.x] = s;
        }
    }
};

// ---- fused tail: the whole inner iteration (linearise + reduce + solve + pose update) is ONE launch.  The block with the highest
// index is the REDUCER: after its own tile it sweeps the granules of all block partials until every one carries this launch's key
// (GramAcc::finish), adds them in the fixed order of reduce_partials_block — so the record does not depend on timing — and applies
// the Gauss-Newton update.  The other blocks just store and leave: no ticket, no fence, no drained-store wait
// (round 1's __threadfence pair, and a first round-2 version with write-through partials + sharded tickets, both cost more than
// the kernel boundary they saved: 46.1 / 39.9 vs 43.0 / 38.6 us per iteration — the chain store -> ack -> atomic -> atomic -> load is
// four memory round trips; the granule sweep is one).  key = launch_key(state->epoch); the reducer advances the epoch.
template <bool XCHG>
__device__ __forceinline__ void reduce_partials_block(const double* part_surf, int nb_surf, const double* part_edge, int nb_edge,
                                                      double* __restrict__ out, SlotState* __restrict__ state, int do_gn, unsigned long long key, const P2PView& xv);   // defined below
__device__ __forceinline__ void fused_tail(const FuseTail& fz, unsigned long long key) {
    if (fz.mode != 1 && fz.mode != 2) return;      // 0: plain partials for k_reduce_partials; 3: publish only (the per-kind launches of merge_kinds = 0: the second launch reduces)
    if (blockIdx.x != gridDim.x - 1) return;
    P2PView none{};       // the fused tail is the single-GPU structure: no exchange (seq = 0)
    reduce_partials_block<false>(fz.part_surf, fz.nb_surf, fz.part_edge, fz.nb_edge, fz.out, fz.state, (fz.mode == 2 ? 1 : 0) | (fz.debug & 256), key, none);
}

__device__ __forceinline__ void load_body_pose(const PoseArg& pa, dq& Q, d3& T) {
    if (pa.state) { const double* s = pa.state->pose; T = d3{s[0], s[1], s[2]}; Q = dq{s[3], s[4], s[5], s[6]}; }
    else { T = d3{pa.t[0], pa.t[1], pa.t[2]}; Q = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]}; }
}

// Residual, 1x7 Jacobian row and loss corrector of ONE correspondence (Jr[0..6] = robustified Jacobian, Jr[7] = residual; returns the robust
// cost) — shared by the linearisation launch and by the association launch that linearises on the fly (k_associate_lin).
//   surf: LidarPlaneNormFactor / LidarPlaneNormIncreFactor, L/include/factors/LidarKeyframeFactor.h:86-90, 118-128; `score` is the (count-scaled) weight
__device__ __forceinline__ double surf_lin_row(const MatchParams& P, const dq& Q, const d3& T, const dq& qlb_inv, float4 ql, float4 nd, double score, double Jr[8]) {
    d3 cp{(double)ql.x, (double)ql.y, (double)ql.z};
    d3 n{(double)nd.x, (double)nd.y, (double)nd.z};
    d3 v;
    if (P.variant == 2) { v = cp; score = 1.0; }   // LidarPlaneNormIncreFactor, LidarKeyframeFactor.h:118-128
    else v = qrot(qlb_inv, cp - d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});                                  // :86
    d3 pw = qrot(Q, v) + T;                                                                              // :87
    double r = score * (dot3(n, pw) + (double)nd.w);                                                    // :90
    double jq[4];
    qrot_jac_row(Q, v, n, jq);
    double J[7] = {score * n.x, score * n.y, score * n.z, score * jq[0], score * jq[1], score * jq[2], score * jq[3]};
    const double cost = robustify(P.loss, P.loss_a, J, r, P.no_cost == 0);
#pragma unroll
    for (int k = 0; k < 7; k++) Jr[k] = J[k];
    Jr[7] = r;
    return cost;
}
//   edge: LidarEdgeFactor, LidarKeyframeFactor.h:38-44 (no extrinsic: SURVEY F6); `s` is the (count-scaled) weight
__device__ __forceinline__ double edge_lin_row(const MatchParams& P, const dq& Q, const d3& T, float4 ql, float4 fa, float4 fb, double s, double Jr[8]) {
    d3 cp{(double)ql.x, (double)ql.y, (double)ql.z};
    d3 Av{(double)fa.x, (double)fa.y, (double)fa.z}, B{(double)fb.x, (double)fb.y, (double)fb.z};
    d3 lp = qrot(Q, cp) + T;                    // :38
    d3 nu = cross3(lp - Av, lp - B);            // :40
    d3 de = Av - B;                             // :41
    double nn = sqrt(dot3(nu, nu)), dn = sqrt(dot3(de, de));
    double r = s * (nn / dn);                   // :43-44
    // d|nu|/dlp = nu^T [a-b]x / |nu| = (nu x (B - A))^T / |nu|
    d3 g = cross3(nu, B - Av);
    double k = s / (nn * dn);
    g = k * g;
    double jq[4];
    qrot_jac_row(Q, cp, g, jq);
    double J[7] = {g.x, g.y, g.z, jq[0], jq[1], jq[2], jq[3]};
    const double cost = robustify(P.loss, P.loss_a, J, r, P.no_cost == 0);
#pragma unroll
    for (int kk = 0; kk < 7; kk++) Jr[kk] = J[kk];
    Jr[7] = r;
    return cost;
}

// Linearisation bodies: `bid` of `nb` virtual blocks of one kind (the combined surf + edge launch maps its grid onto both).
__device__ __forceinline__ void lin_surf_body(const LinArgs& A, int bid, const PoseArg& pa, const MatchParams& P, const SlotState* __restrict__ state,
                                              const int* __restrict__ n_global, double* lds, unsigned long long key) {
    const float4* __restrict__ queries = A.queries; const float4* __restrict__ rec_nd = A.rec0;
    const double* __restrict__ rec_score = reinterpret_cast<const double*>(A.rec1);
    const unsigned char* __restrict__ valid = A.valid;
    const int n_q = A.n_q;
    tstamp(state, P.debug, 100, 0);
    if ((P.debug & 512) && bid == 100 && threadIdx.x == 0) const_cast<SlotState*>(state)->tprof[15] = (long long)__builtin_amdgcn_s_memrealtime();
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    const dq qlb_inv{P.q_lb_inv_jet[0], P.q_lb_inv_jet[1], P.q_lb_inv_jet[2], P.q_lb_inv_jet[3]};
    // N of R:861: this rank's count (sum of the association's block counts) or, when a multi-GPU caller has
    // all-reduced it, the global count in state->n_res
    // the first tile's records are requested before the count reduction below (which synchronises the block twice), and
    // unconditionally — one memory round trip instead of valid -> record
    const int BS = blockDim.x;
    const int i0 = bid * BS + threadIdx.x;
    const int i0c = min(i0, n_q - 1);
    unsigned char v0 = valid[i0c];
    float4 ql0 = queries[i0c], nd0 = rec_nd[i0c];
    double sc0 = rec_score[i0c];
    // ROT count scaling exactly as the reference writes it (R/src/BackendFusion.cpp:861, pinned by tests/test_reference_*.py against the reference text):
    // vec_surf_scores[i] * 1000 / vec_surf_res_cnt  =  (score * 1000.0) / (double)N — a multiply, then a true division
    double n_den = 1.0;
    if (P.debug & 128) n_den = 190000.0;
    else if (P.scale_surf_num > 0) n_den = (double)(A.block_counts ? sum_block_counts(A.block_counts, A.n_bc) : (n_global ? n_global[0] : state->n_res[0]));
    tstamp(state, P.debug, 100, 1);
    for (int base = bid * BS; base < n_q; base += A.nb * BS) {
        int i = base + threadIdx.x;
        const bool first = base == bid * BS;
        const int ic = min(i, n_q - 1);
        bool ok = i < n_q && (first ? v0 : valid[ic]);
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            float4 ql = first ? ql0 : queries[i]; float4 nd = first ? nd0 : rec_nd[i];
            double score = first ? sc0 : rec_score[i];
            if (P.scale_surf_num > 0) score = score * P.scale_surf_num / n_den;
            cost = surf_lin_row(P, Q, T, qlb_inv, ql, nd, score, Jr);
        }
        tstamp(state, P.debug, 100, 2);
        ga.add_rows(Jr, cost, ok, lds);
        tstamp(state, P.debug, 100, 3);
        if ((P.debug & 512) && bid == 100 && (threadIdx.x & 63) == 0) const_cast<SlotState*>(state)->tprof[threadIdx.x >> 6] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    ga.finish(lds, A.partials + (size_t)bid * kPartialStride, key);
    tstamp(state, P.debug, 100, 4);
}

__device__ __forceinline__ void lin_edge_body(const LinArgs& A, int bid, const PoseArg& pa, const MatchParams& P, const SlotState* __restrict__ state,
                                              const int* __restrict__ n_global, double* lds, unsigned long long key) {
    const float4* __restrict__ queries = A.queries; const float4* __restrict__ rec_a = A.rec0;
    const float4* __restrict__ rec_b = reinterpret_cast<const float4*>(A.rec1);
    const unsigned char* __restrict__ valid = A.valid;
    const int n_q = A.n_q;
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    // R:843: points[i].intensity * 200 / vec_edge_res_cnt — float * int / int, i.e. FLOAT arithmetic (pinned by tests/test_reference_*.py against the reference text)
    float n_den = 1.0f;
    if (P.scale_edge_num > 0) n_den = (float)(A.block_counts ? sum_block_counts(A.block_counts, A.n_bc) : (n_global ? n_global[1] : state->n_res[1]));
    const float n_num = (float)P.scale_edge_num;
    for (int base = bid * blockDim.x; base < n_q; base += A.nb * blockDim.x) {
        int i = base + threadIdx.x;
        bool ok = i < n_q && valid[i];
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            float4 ql = queries[i]; float4 fa = rec_a[i], fb = rec_b[i];
            double s = (double)fa.w;
            if (P.scale_edge_num > 0) s = (double)__fdiv_rn(__fmul_rn(fa.w, n_num), n_den);
            cost = edge_lin_row(P, Q, T, ql, fa, fb, s, Jr);
        }
        ga.add_rows(Jr, cost, ok, lds);
    }
    ga.finish(lds, A.partials + (size_t)bid * kPartialStride, key);
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

// ================================================================================================
// Final reduction of block partials (fixed order) -> 72-double record, optionally followed by the
// Gauss-Newton update in the same launch (single-GPU path; multi-GPU callers all-reduce in between).
// out: [0..63] full symmetric 8x8 Gram (row-major), [64] cost, [65] n_surf, [66] n_edge.
//
// GN step (device mirror of ceres::QuaternionParameterization):
//   P = blockdiag(I3, plusJacobian(q) 4x3); H = P^T G77 P, g = P^T G7r; solve H d = -g (Cholesky);
//   t += d[0:3]; q = [cos|dq|, sin|dq|/|dq| dq] (x) q
// ================================================================================================
// wave-level ordering of LDS traffic inside ONE wave (LDS operations of a wave execute in order; the fence keeps the compiler from
// moving them) — the tail of the reduction and the GN update run in a single wave, without s_barrier
#define LILI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
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
    __shared__ NoTile L;
    __shared__ __attribute__((aligned(16))) RowTabT<BS> tab;
    static_assert(sizeof(RowTabT<BS>) >= (size_t)BS * kRow * sizeof(double), "Gram staging rows must fit the row table");
    const AssocSched sched{nullptr, nullptr};
    const int b = (int)blockIdx.x;
    const bool edge = b < E.nb;
    LaneRec rec{};
    if (edge) assoc_edge_body<false, BS>(E.queries, nullptr, nullptr, E.n_q, E.g, pa, P, E.rec0, reinterpret_cast<float4*>(E.rec1), E.valid, E.dbg_idx, E.dbg_d2,
                                         E.block_counts, E.nn_cache, sched, b, L, tab, &rec);
    else assoc_surf_body<false, BS>(S.queries, nullptr, nullptr, S.n_q, S.g, pa, P, S.rec0, reinterpret_cast<double*>(S.rec1), S.valid, S.dbg_idx, S.dbg_d2,
                                    S.block_counts, S.nn_cache, sched, b - E.nb, L, tab, &rec);
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
// sin(x)/x and cos(x) from x^2 for the small rotation of a Gauss-Newton step (|x| < 0.5: the series are truncated below 1e-19
// relative): 16 fused multiply-adds instead of two libm calls with argument reduction (~120 dependent f64 instructions at the
// end of the latency-bound update chain).  ceres::QuaternionParameterization::Plus takes sin / cos from libm, which is not
// correctly rounded either; the two agree to 1-2 ulp.
__device__ __forceinline__ void sinc_cos_small(double x2, double& sinc, double& c) {
    double s = -1.0 / 355687428096000.0;            // -1/17!
    s = __fma_rn(s, x2, 1.0 / 1307674368000.0);      //  1/15!
    s = __fma_rn(s, x2, -1.0 / 6227020800.0);        // -1/13!
    s = __fma_rn(s, x2, 1.0 / 39916800.0);           //  1/11!
    s = __fma_rn(s, x2, -1.0 / 362880.0);            // -1/9!
    s = __fma_rn(s, x2, 1.0 / 5040.0);
    s = __fma_rn(s, x2, -1.0 / 120.0);
    s = __fma_rn(s, x2, 1.0 / 6.0);
    sinc = __fma_rn(-s, x2, 1.0);
    double k = 1.0 / 6402373705728000.0;             //  1/18!
    k = __fma_rn(k, x2, -1.0 / 20922789888000.0);    // -1/16!
    k = __fma_rn(k, x2, 1.0 / 87178291200.0);        //  1/14!
    k = __fma_rn(k, x2, -1.0 / 479001600.0);         // -1/12!
    k = __fma_rn(k, x2, 1.0 / 3628800.0);            //  1/10!
    k = __fma_rn(k, x2, -1.0 / 40320.0);
    k = __fma_rn(k, x2, 1.0 / 720.0);
    k = __fma_rn(k, x2, -1.0 / 24.0);
    k = __fma_rn(k, x2, 0.5);
    c = __fma_rn(-k, x2, 1.0);
}
__device__ void gn_update_block(const double* gram /*LDS or global, 64+*/, SlotState* __restrict__ state, const double xq[4]) {
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
        if (okc) {
            state->pose[0] += d[0]; state->pose[1] += d[1]; state->pose[2] += d[2];
            const double nd2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
            if (nd2 > 0.0) {
                double sbd, cw;
                if (nd2 < 0.25) sinc_cos_small(nd2, sbd, cw);
                else { const double nd = sqrt(nd2); sbd = sin(nd) / nd; cw = cos(nd); }
                dq qd{cw, sbd * d[3], sbd * d[4], sbd * d[5]};
                dq r = qmul(qd, dq{x0, x1, x2, x3});
                state->pose[3] = r.w; state->pose[4] = r.x; state->pose[5] = r.y; state->pose[6] = r.z;
            }
#pragma unroll
            for (int i = 0; i < 6; i++) state->last_delta[i] = d[i];
            state->gn_status = 0;
        } else state->gn_status = 1;
        state->iters += 1;
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
                                                      double* __restrict__ out, SlotState* __restrict__ state, int do_gn, unsigned long long key, const P2PView& xv) {
    tstamp(state, do_gn, (int)blockIdx.x, 8);
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
        if (!p2p_exchange_wave<true>(xv, 72, w0, w1, s0, s1)) { if (lane == 0) state->gn_status = 2; return; }
        full[lane] = __longlong_as_double((long long)s0);
        if (lane < 8) full[64 + lane] = __longlong_as_double((long long)s1);
        LILI_WAVE_SYNC();
    }
    out[lane] = full[lane];
    if (lane < 8) out[64 + lane] = full[64 + lane];
    if (key && lane == 0) state->epoch = state->epoch + 1ull;      // the next fused launch of this slot gets a new key (stream order)
    tstamp(state, do_gn, (int)blockIdx.x, 10);
    if (do_gn & 1) gn_update_block(full, state, xq);
    tstamp(state, do_gn, (int)blockIdx.x, 11);
}

__global__ __launch_bounds__(kReduceThreads) void k_reduce_partials(const double* __restrict__ part_surf, int nb_surf,
                                                            const double* __restrict__ part_edge, int nb_edge,
                                                            double* __restrict__ out, SlotState* __restrict__ state, int do_gn, P2PView v) {
    reduce_partials_block<true>(part_surf, nb_surf, part_edge, nb_edge, out, state, do_gn, 0ull, v);
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
