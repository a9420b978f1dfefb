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
#include "lili_kernels.h"
#include "lili_device_math.h"

namespace lili {

// ================================================================================================
// cloud ingestion: AoS points (stride 32 / 48 B ...) -> float4 (x, y, z, aux)
// ================================================================================================
__global__ void k_cloud_to_f4(const unsigned char* __restrict__ raw, int n, int stride, int aux_off, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(raw + (size_t)i * stride);
    float4 v;
    v.x = p[0]; v.y = p[1]; v.z = p[2];
    v.w = aux_off >= 0 ? *reinterpret_cast<const float*>(raw + (size_t)i * stride + aux_off) : 0.f;
    out[i] = v;
}

// ================================================================================================
// K7 — map index build: bounding box, cell histogram, exclusive scan, scatter
// ================================================================================================
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

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

__global__ void k_cell_count(const float4* __restrict__ pts, int n, GridView g, int* __restrict__ cell_count, int* __restrict__ pt_cell) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = cell_of(pts[i], g);
    pt_cell[i] = c;
    atomicAdd(&cell_count[c], 1);
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
__global__ void k_scan_apply(const int* __restrict__ in, int64_t n, const int* __restrict__ block_offs, int* __restrict__ out /*[n+1]*/) {
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

__global__ void k_scatter(const float4* __restrict__ pts, int n, const int* __restrict__ pt_cell, const int* __restrict__ cell_start,
                          int* __restrict__ cell_fill, float4* __restrict__ sorted, float* __restrict__ aux_sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = pt_cell[i];
    int pos = cell_start[c] + atomicAdd(&cell_fill[c], 1);
    float4 p = pts[i];
    if (aux_sorted) aux_sorted[pos] = p.w;
    p.w = __int_as_float(i);
    sorted[pos] = p;
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

// ================================================================================================
// exact 5-NN inside the 27-cell neighbourhood
// ================================================================================================
struct Top5 {
    float d[5];
    int j[5];     // position in the cell-sorted array
};
// FLANN L2_Simple on 3 floats (f32, x then y then z, no FMA)
__device__ __forceinline__ float dist2(float4 p, float qx, float qy, float qz) {
    float r = 0.f;
    float dx = qx - p.x; r += dx * dx;
    float dy = qy - p.y; r += dy * dy;
    float dz = qz - p.z; r += dz * dz;
    return r;
}
// ascending (d2, original index): ties on d2 are resolved by the map point's original index so that
// the result does not depend on the traversal order (FLANN's own tie order is unspecified, App. B1).
// The index is only fetched on an exact tie.
__device__ __forceinline__ bool less_dj(float da, int ja, float db, int jb, const float4* __restrict__ pts) {
    if (da < db) return true;
    if (!(da == db)) return false;
    int ia = ja < 0 ? 0x7fffffff : __float_as_int(pts[ja].w);
    int ib = jb < 0 ? 0x7fffffff : __float_as_int(pts[jb].w);
    return ia < ib;
}
__device__ __forceinline__ void top5_insert(Top5& t, float d, int j, const float4* __restrict__ pts) {
    if (!less_dj(d, j, t.d[4], t.j[4], pts)) return;
    t.d[4] = d; t.j[4] = j;
#pragma unroll
    for (int k = 4; k > 0; k--) {
        if (!less_dj(t.d[k], t.j[k], t.d[k - 1], t.j[k - 1], pts)) break;
        float td = t.d[k]; t.d[k] = t.d[k - 1]; t.d[k - 1] = td;
        int tj = t.j[k]; t.j[k] = t.j[k - 1]; t.j[k - 1] = tj;
    }
}

__device__ __forceinline__ void knn5_grid(const GridView& g, float qx, float qy, float qz, Top5& best) {
#pragma unroll
    for (int k = 0; k < 5; k++) { best.d[k] = INFINITY; best.j[k] = -1; }
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz))) return;
    int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
    // queries more than one cell outside the grid cannot have a neighbour within the gate radius
    if (cx < -1 || cx > g.nx || cy < -1 || cy > g.ny || cz < -1 || cz > g.nz) return;
    int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    if (x0 > x1) return;
    for (int dz = -1; dz <= 1; dz++) {
        int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; dy++) {
            int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            int row = (z * g.ny + y) * g.nx;
            int beg = g.cell_start[row + x0], end = g.cell_start[row + x1 + 1];
            for (int j = beg; j < end; j++) {
                float4 p = g.pts[j];
                float d = dist2(p, qx, qy, qz);
                if (d <= best.d[4]) top5_insert(best, d, j, g.pts);
            }
        }
    }
}

// Correspondence counting without atomics on a shared word (3128 same-address atomics cost ~40 us on
// MI355X): each block stores its own count; consumers add the <= few-thousand block counts themselves.
__device__ __forceinline__ void store_block_count(bool ok, int* __restrict__ block_counts) {
    __shared__ int wave_cnt[kBlock / 64];
    unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) s += wave_cnt[w];
        block_counts[blockIdx.x] = s;
    }
}
// Sum of the per-block counts of one association launch (every thread of the block gets the total).
__device__ __forceinline__ int sum_block_counts(const int* __restrict__ block_counts, int nb) {
    __shared__ int part[kBlock / 64];
    __shared__ int total;
    int s = 0;
    for (int b = threadIdx.x; b < nb; b += kBlock) s += block_counts[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < kBlock / 64; w++) t += part[w]; total = t; }
    __syncthreads();
    return total;
}
__global__ __launch_bounds__(kBlock) void k_sum_counts(const int* __restrict__ bc_surf, int nb_surf, const int* __restrict__ bc_edge, int nb_edge,
                                                      SlotState* __restrict__ state) {
    if (bc_surf) { int t = sum_block_counts(bc_surf, nb_surf); if (threadIdx.x == 0) state->n_res[0] = t; }
    __syncthreads();
    if (bc_edge) { int t = sum_block_counts(bc_edge, nb_edge); if (threadIdx.x == 0) state->n_res[1] = t; }
}

__device__ __forceinline__ void load_assoc_pose(const PoseArg& pa, const MatchParams& P, dq& Q2, d3& T2) {
    if (pa.state) {
        const double* s = pa.state->pose;
        dq Q{s[3], s[4], s[5], s[6]};
        d3 T{s[0], s[1], s[2]};
        if (pa.derive_assoc) {   // L/src/BackendFusion.cpp:929-930
            Q2 = qmul(Q, qinv(dq{P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]}));
            T2 = T - qrot(Q2, d3{P.t_lb[0], P.t_lb[1], P.t_lb[2]});
        } else { Q2 = Q; T2 = T; }
    } else {
        Q2 = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]};
        T2 = d3{pa.t[0], pa.t[1], pa.t[2]};
    }
}

// ================================================================================================
// K5 — surf association.  One thread per query.
// records: rec_nd[i] = (w*nx, w*ny, w*nz, w*normInverse) as floats, rec_score[i] (f64), valid[i]
// ================================================================================================
__global__ __launch_bounds__(kBlock) void k_associate_surf(
        const float4* __restrict__ queries, const int* __restrict__ perm, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_nd, double* __restrict__ rec_score, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (t < n_q) {
        int i = perm ? perm[t] : t;
        dq Q2; d3 T2;
        load_assoc_pose(pa, P, Q2, T2);
        float4 ql = queries[i];
        d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;   // transformPoint, L:695-711
        float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
        Top5 nn;
        if (P.debug & 2) {
#pragma unroll
            for (int k = 0; k < 5; k++) { nn.d[k] = 0.01f * (k + 1); nn.j[k] = (i * 7 + k) % g.n_points; }
        } else knn5_grid(g, px, py, pz, nn);
        if (dbg_idx) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                dbg_idx[(size_t)i * 5 + k] = nn.j[k] >= 0 ? __float_as_int(g.pts[nn.j[k]].w) : -1;
                dbg_d2[(size_t)i * 5 + k] = nn.d[k];
            }
        }
        float4 rn = make_float4(0.f, 0.f, 0.f, 0.f);
        double score = 0.0;
        if ((P.debug & 1) && nn.j[4] >= 0) { rn.x = nn.d[4]; ok = nn.d[4] < 0.5f; }
        else if (nn.j[4] >= 0 && (double)nn.d[4] < P.kd_max_radius) {   // L:1615
            float4 m[5];
#pragma unroll
            for (int k = 0; k < 5; k++) m[k] = g.pts[nn.j[k]];
            col5 c0, c1, c2, b;
            double sum_w = 0.0;
            bool go = true;
            if (P.variant == 0) {   // Livox reflectivity weighting, L:1617-1638
                double w[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    float diff = ql.w - g.aux[nn.j[k]];
                    double tmp_w = (double)fabsf(diff);
                    sum_w += tmp_w;
                    w[k] = 1.0 / tmp_w;
                }
                if (sum_w > P.reflect_thres) go = false;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    double wk = w[k] / sum_w;
                    c0.v[k] = wk * (double)m[k].x; c1.v[k] = wk * (double)m[k].y; c2.v[k] = wk * (double)m[k].z;
                    b.v[k] = -1.0 * wk;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) { c0.v[k] = (double)m[k].x; c1.v[k] = (double)m[k].y; c2.v[k] = (double)m[k].z; b.v[k] = -1.0; }
            }
            if (go) {
                double nv[3];
                lstsq53(c0, c1, c2, b, nv);
                double nn_ = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                double normInverse = 1.0 / nn_;
                nv[0] /= nn_; nv[1] /= nn_; nv[2] /= nn_;
                bool planeValid = true;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    if (fabs(nv[0] * (double)m[k].x + nv[1] * (double)m[k].y + nv[2] * (double)m[k].z + normInverse) > P.surf_dist_thres) planeValid = false;
                }
                if (planeValid) {
                    // L:1661-1662: float pd, float weight; sqrt(sqrt()) on a float argument is the float overload
                    float pd = (float)(nv[0] * (double)px + nv[1] * (double)py + nv[2] * (double)pz + normInverse);
                    float r2 = px * px + py * py + pz * pz;
                    float weight = (float)(1.0 - 0.9 * (double)fabsf(pd) / (double)sqrtf(sqrtf(r2)));
                    if ((double)weight > P.surf_weight_min) {
                        ok = true;
                        rn.x = (float)((double)weight * nv[0]); rn.y = (float)((double)weight * nv[1]); rn.z = (float)((double)weight * nv[2]);
                        rn.w = (float)((double)weight * normInverse);
                        if (P.variant == 0) score = P.lidar_const * ((double)weight + exp(-sum_w));   // L:1676
                        else if (P.variant == 1) score = P.lidar_const * (double)weight;                // R:1515
                        else score = 1.0;
                    }
                }
            }
        }
        rec_nd[i] = rn;
        rec_score[i] = score;
        valid[i] = ok ? 1 : 0;
    }
    store_block_count(ok, block_counts);
}

// ================================================================================================
// K6 — edge association.  records: rec_a[i] = (Ax, Ay, Az, s), rec_b[i] = (Bx, By, Bz, 0), valid[i]
// ================================================================================================
__global__ __launch_bounds__(kBlock) void k_associate_edge(
        const float4* __restrict__ queries, const int* __restrict__ perm, int n_q, GridView g, PoseArg pa, MatchParams P,
        float4* __restrict__ rec_a, float4* __restrict__ rec_b, unsigned char* __restrict__ valid,
        int* __restrict__ dbg_idx, float* __restrict__ dbg_d2, int* __restrict__ block_counts) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (t < n_q) {
        int i = perm ? perm[t] : t;
        dq Q2; d3 T2;
        load_assoc_pose(pa, P, Q2, T2);
        float4 ql = queries[i];
        d3 pmd = qrot(Q2, d3{(double)ql.x, (double)ql.y, (double)ql.z}) + T2;
        float px = (float)pmd.x, py = (float)pmd.y, pz = (float)pmd.z;
        Top5 nn;
        knn5_grid(g, px, py, pz, nn);
        if (dbg_idx) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                dbg_idx[(size_t)i * 5 + k] = nn.j[k] >= 0 ? __float_as_int(g.pts[nn.j[k]].w) : -1;
                dbg_d2[(size_t)i * 5 + k] = nn.d[k];
            }
        }
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (nn.j[4] >= 0 && (double)nn.d[4] < P.edge_gate) {   // L:1543
            d3 m[5]; d3 c{0, 0, 0};
#pragma unroll
            for (int k = 0; k < 5; k++) { float4 p = g.pts[nn.j[k]]; m[k] = d3{(double)p.x, (double)p.y, (double)p.z}; c = c + m[k]; }
            c = d3{c.x / 5.0, c.y / 5.0, c.z / 5.0};
            double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                d3 z = m[k] - c;
                a00 += z.x * z.x; a01 += z.x * z.y; a02 += z.x * z.z; a11 += z.y * z.y; a12 += z.y * z.z; a22 += z.z * z.z;
            }
            double ev[3]; d3 vmin, vmax;
            eig3_sym(a00, a01, a02, a11, a12, a22, ev, vmin, vmax);
            if (ev[2] > 3.0 * ev[1]) {   // L:1575
                d3 u = canon_sign(vmax);
                d3 A = c + 0.1 * u, B = c - 0.1 * u;
                bool keep = true;
                if (P.edge_dist_max > 0) {   // R:1437-1443
                    d3 lp{(double)px, (double)py, (double)pz};
                    d3 nu = cross3(lp - A, lp - B);
                    d3 de = A - B;
                    double dist = sqrt(dot3(nu, nu)) / sqrt(dot3(de, de));
                    keep = dist < P.edge_dist_max;
                }
                if (keep) {
                    ok = true;
                    ra = make_float4((float)A.x, (float)A.y, (float)A.z, (float)P.lidar_const);
                    rb = make_float4((float)B.x, (float)B.y, (float)B.z, 0.f);
                }
            }
        }
        rec_a[i] = ra; rec_b[i] = rb; valid[i] = ok ? 1 : 0;
    }
    store_block_count(ok, block_counts);
}

// ================================================================================================
// Linearisation: residual + 1x7 global Jacobian per record, loss corrector, Gram reduction.
//
// Reduction scheme (deterministic, no float atomics): every wave stages its 64 rows [J0..J6, r, cost]
// in LDS; lane l < 36 owns Gram entry (a,b) of the upper triangle and sums row[q][a]*row[q][b] over
// q = 0..63 in order; lane 36 sums the cost column.  Waves of a block are then added in order and the
// block writes one 40-double partial; k_reduce_gn adds the partials in a fixed order.
// ================================================================================================
constexpr int kRow = 9;
// Lane l < 36 owns the upper-triangle entry (a, b); lane 36 the cost column; lane 37 the count.
struct GramAcc {
    double acc; int a, b;
    __device__ __forceinline__ void init() {
        int lane = threadIdx.x & 63;
        a = 0; int l = lane;
        while (a < 8 && l >= 8 - a) { l -= 8 - a; a++; }
        b = a + l;
        if (lane >= 36) { a = 8; b = 8; }
        acc = 0.0;
    }
    // all threads of the block must call this (it synchronises)
    __device__ __forceinline__ void add_rows(const double Jr[8], double cost, bool ok, double* lds) {
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double* rows = lds + wave * 64 * kRow;
        double* myrow = rows + lane * kRow;
#pragma unroll
        for (int k = 0; k < 8; k++) myrow[k] = ok ? Jr[k] : 0.0;
        myrow[8] = ok ? cost : 0.0;
        unsigned long long bal = __ballot(ok);
        __syncthreads();
        if (lane < 36) {
            double s = 0.0;
#pragma unroll 16
            for (int q = 0; q < 64; q++) s += rows[q * kRow + a] * rows[q * kRow + b];
            acc += s;
        } else if (lane == 36) {
            double s = 0.0;
#pragma unroll 16
            for (int q = 0; q < 64; q++) s += rows[q * kRow + 8];
            acc += s;
        } else if (lane == 37) {
            acc += (double)__popcll(bal);
        }
        __syncthreads();
    }
    __device__ __forceinline__ void finish(double* lds, double* __restrict__ partial_out) {
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double* wsum = lds + kBlock * kRow;
        if (lane < 40) wsum[wave * 40 + lane] = lane < 38 ? acc : 0.0;
        __syncthreads();
        if (threadIdx.x < 40) {
            double s = 0.0;
            for (int w = 0; w < kBlock / 64; w++) s += wsum[w * 40 + threadIdx.x];
            partial_out[threadIdx.x] = s;
        }
    }
};

__device__ __forceinline__ void load_body_pose(const PoseArg& pa, dq& Q, d3& T) {
    if (pa.state) { const double* s = pa.state->pose; T = d3{s[0], s[1], s[2]}; Q = dq{s[3], s[4], s[5], s[6]}; }
    else { T = d3{pa.t[0], pa.t[1], pa.t[2]}; Q = dq{pa.q[0], pa.q[1], pa.q[2], pa.q[3]}; }
}

__global__ __launch_bounds__(kBlock) void k_linearize_surf(
        const float4* __restrict__ queries, int n_q, const float4* __restrict__ rec_nd, const double* __restrict__ rec_score,
        const unsigned char* __restrict__ valid, PoseArg pa, MatchParams P, const SlotState* __restrict__ state,
        const int* __restrict__ block_counts, int n_bc, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    const dq qlb_inv = qinv(dq{P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]});
    // N of R:861: this rank's count (sum of the association's block counts) or, when a multi-GPU caller has
    // all-reduced it, the global count in state->n_res
    double nscale = 1.0;
    if (P.scale_surf_num > 0) nscale = P.scale_surf_num / (double)(block_counts ? sum_block_counts(block_counts, n_bc) : state->n_res[0]);
    for (int base = blockIdx.x * kBlock; base < n_q; base += gridDim.x * kBlock) {
        int i = base + threadIdx.x;
        bool ok = i < n_q && valid[i];
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            float4 ql = queries[i]; float4 nd = rec_nd[i];
            double score = rec_score[i];
            if (P.scale_surf_num > 0) score = score * nscale;
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
            cost = robustify(P.loss, P.loss_a, J, r);
#pragma unroll
            for (int k = 0; k < 7; k++) Jr[k] = J[k];
            Jr[7] = r;
        }
        ga.add_rows(Jr, cost, ok, lds);
    }
    ga.finish(lds, partials + (size_t)blockIdx.x * kPartialDoubles);
}

__global__ __launch_bounds__(kBlock) void k_linearize_edge(
        const float4* __restrict__ queries, int n_q, const float4* __restrict__ rec_a, const float4* __restrict__ rec_b,
        const unsigned char* __restrict__ valid, PoseArg pa, MatchParams P, const SlotState* __restrict__ state,
        const int* __restrict__ block_counts, int n_bc, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    GramAcc ga; ga.init();
    dq Q; d3 T;
    load_body_pose(pa, Q, T);
    double nscale = 1.0;   // R:843
    if (P.scale_edge_num > 0) nscale = P.scale_edge_num / (double)(block_counts ? sum_block_counts(block_counts, n_bc) : state->n_res[1]);
    for (int base = blockIdx.x * kBlock; base < n_q; base += gridDim.x * kBlock) {
        int i = base + threadIdx.x;
        bool ok = i < n_q && valid[i];
        double Jr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double cost = 0.0;
        if (ok) {
            float4 ql = queries[i]; float4 fa = rec_a[i], fb = rec_b[i];
            double s = (double)fa.w;
            if (P.scale_edge_num > 0) s = s * nscale;
            d3 cp{(double)ql.x, (double)ql.y, (double)ql.z};
            d3 A{(double)fa.x, (double)fa.y, (double)fa.z}, B{(double)fb.x, (double)fb.y, (double)fb.z};
            d3 lp = qrot(Q, cp) + T;                    // LidarKeyframeFactor.h:38 (no extrinsic: SURVEY F6)
            d3 nu = cross3(lp - A, lp - B);             // :40
            d3 de = A - B;                              // :41
            double nn = sqrt(dot3(nu, nu)), dn = sqrt(dot3(de, de));
            double r = s * (nn / dn);                   // :43-44
            // d|nu|/dlp = nu^T [a-b]x / |nu| = (nu x (B - A))^T / |nu|
            d3 g = cross3(nu, B - A);
            double k = s / (nn * dn);
            g = k * g;
            double jq[4];
            qrot_jac_row(Q, cp, g, jq);
            double J[7] = {g.x, g.y, g.z, jq[0], jq[1], jq[2], jq[3]};
            cost = robustify(P.loss, P.loss_a, J, r);
#pragma unroll
            for (int kk = 0; kk < 7; kk++) Jr[kk] = J[kk];
            Jr[7] = r;
        }
        ga.add_rows(Jr, cost, ok, lds);
    }
    ga.finish(lds, partials + (size_t)blockIdx.x * kPartialDoubles);
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
__device__ void gn_update_block(const double* gram /*LDS or global, 64+*/, SlotState* __restrict__ state) {
    __shared__ double Jq[4][3];   // plus-Jacobian rows: [-x1 -x2 -x3; x0 x3 -x2; -x3 x0 x1; x2 -x1 x0]
    __shared__ double M[7][6];    // M = G77 * P   (P = blockdiag(I3, Jq): only 4 terms per entry)
    __shared__ double H[6][6];
    __shared__ double gvec[6];
    int tid = threadIdx.x;
    const double x0 = state->pose[3], x1 = state->pose[4], x2 = state->pose[5], x3 = state->pose[6];
    if (tid < 12) {
        int rr = tid / 3, cc = tid % 3;
        double e0 = rr == 0 ? -x1 : rr == 1 ? x0 : rr == 2 ? -x3 : x2;
        double e1 = rr == 0 ? -x2 : rr == 1 ? x3 : rr == 2 ? x0 : -x1;
        double e2 = rr == 0 ? -x3 : rr == 1 ? -x2 : rr == 2 ? x1 : x0;
        Jq[rr][cc] = cc == 0 ? e0 : cc == 1 ? e1 : e2;
    }
    __syncthreads();
    if (tid < 42) {
        int i = tid / 6, b = tid % 6;
        double v;
        if (b < 3) v = gram[i * 8 + b];
        else v = ((gram[i * 8 + 3] * Jq[0][b - 3] + gram[i * 8 + 4] * Jq[1][b - 3]) + gram[i * 8 + 5] * Jq[2][b - 3]) + gram[i * 8 + 6] * Jq[3][b - 3];
        M[i][b] = v;
    }
    __syncthreads();
    if (tid < 36) {
        int a = tid / 6, b = tid % 6;
        double v;
        if (a < 3) v = M[a][b];
        else v = ((Jq[0][a - 3] * M[3][b] + Jq[1][a - 3] * M[4][b]) + Jq[2][a - 3] * M[5][b]) + Jq[3][a - 3] * M[6][b];
        H[a][b] = v;
    } else if (tid < 42) {
        int a = tid - 36;
        double v;
        if (a < 3) v = gram[a * 8 + 7];
        else v = ((Jq[0][a - 3] * gram[3 * 8 + 7] + Jq[1][a - 3] * gram[4 * 8 + 7]) + Jq[2][a - 3] * gram[5 * 8 + 7]) + Jq[3][a - 3] * gram[6 * 8 + 7];
        gvec[a] = -v;
    }
    __syncthreads();
    if (tid == 0) {
        // 6x6 Cholesky solve entirely in registers (all indices are compile-time constants after unrolling)
        double L[6][6], d[6];
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
            for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k];
            if (!(dj > 0)) okc = false;
            dj = sqrt(dj); L[j][j] = dj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
                double sv = L[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) sv -= L[i][k] * L[j][k];
                L[i][j] = sv / dj;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double sv = d[i];
#pragma unroll
            for (int k = 0; k < i; k++) sv -= L[i][k] * d[k];
            d[i] = sv / L[i][i];
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            double sv = d[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) sv -= L[k][i] * d[k];
            d[i] = sv / L[i][i];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) if (!(d[i] == d[i])) okc = false;
        if (okc) {
            state->pose[0] += d[0]; state->pose[1] += d[1]; state->pose[2] += d[2];
            double nd = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
            if (nd > 0.0) {
                double sbd = sin(nd) / nd;
                dq qd{cos(nd), sbd * d[3], sbd * d[4], sbd * d[5]};
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
__global__ __launch_bounds__(kReduceThreads) void k_reduce_partials(const double* __restrict__ part_surf, int nb_surf,
                                                            const double* __restrict__ part_edge, int nb_edge,
                                                            double* __restrict__ out, SlotState* __restrict__ state, int do_gn) {
    constexpr int kGroups = kReduceThreads / 40;   // 25 groups of 40 lanes, group g adds partials g, g+25, ...
    __shared__ double acc[kGroups][2][40];
    __shared__ double tri[40];
    __shared__ double full[72];
    int e = threadIdx.x % 40, g = threadIdx.x / 40;
    if (g < kGroups) {
        // independent loads first (8 in flight per lane), adds in a fixed order afterwards
        double s = 0.0, s2 = 0.0;
        for (int b0 = g; b0 < nb_surf; b0 += kGroups * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { int b = b0 + u * kGroups; v[u] = b < nb_surf ? part_surf[(size_t)b * kPartialDoubles + e] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (int b0 = g; b0 < nb_edge; b0 += kGroups * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { int b = b0 + u * kGroups; v[u] = b < nb_edge ? part_edge[(size_t)b * kPartialDoubles + e] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; u++) s2 += v[u];
        }
        acc[g][0][e] = s; acc[g][1][e] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 40) {
        int k = threadIdx.x;
        double ss = 0.0, se = 0.0;
        for (int gg = 0; gg < kGroups; gg++) { ss += acc[gg][0][k]; se += acc[gg][1][k]; }
        tri[k] = ss + se;
        if (k == 37) { full[65] = ss; full[66] = se; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        int r = threadIdx.x >> 3, c = threadIdx.x & 7;
        int a = r < c ? r : c, b = r < c ? c : r;
        full[threadIdx.x] = tri[a * 8 - a * (a - 1) / 2 + (b - a)];
    }
    if (threadIdx.x == 64) full[64] = tri[36];
    if (threadIdx.x >= 67 && threadIdx.x < 72) full[threadIdx.x] = 0.0;
    __syncthreads();
    if (threadIdx.x < 72) out[threadIdx.x] = full[threadIdx.x];
    if (do_gn) gn_update_block(full, state);
}

__global__ void k_gn_update(const double* __restrict__ gram, SlotState* __restrict__ state) {
    gn_update_block(gram, state);
}

}  // namespace lili
