// Livox Horizon feature extractor on gfx950 — replaces L/src/Preprocessing.cpp:219-383 (L/ = LiLi-OM/) behind
// lili_extract_livox():
//   k_livox_prep     NaN / near filter, line id, IMU deskew (slerp, f64), range / reflectivity gates, time-slot
//                    column, first-writer-wins grid fill as atomicMin(source index) per cell          (L:243-268)
//   k_livox_cut      ordered compaction of every deskewed point with a valid line (/lidar_cloud_cutted, L:253-257)
//   k_livox_grid     gather the winning point of each of the 6 x 4000 cells
//   k_livox_blocks   one wave per 6-column block (664 of them): 36-cell covariance + 3x3 symmetric eigen (f64),
//                    depth-Laplacian edge candidate per line, edge PCA, edge / plane emission with the block-local
//                    tombstones of the reference (L:270-383)
//   k_livox_compact  ordered concatenation (block, then line | column, line)
// 24 k points per scan: this path is latency-, not bandwidth-bound; the grid (1.15 MB in the reference) never
// leaves L2.  Eigenvector signs (stored as normal / direction) are canonicalised — Eigen's are arbitrary.
#include "lili_ctx.h"
#include "lili_device_math.h"

namespace lili {

constexpr int kLvLines = 6, kLvCols = 4000, kLvCells = kLvLines * kLvCols;
constexpr int kLvBlocks = 664;   // i = 5, 11, ... < 3988 (L:270)

struct LivoxDev { double q_imu[4]; double surf_thres, edge_thres; float near_thres; };
struct LivoxState { int n_cut, n_edge, n_surf; };

// Eigen 3.3 slerp of Identity towards b, in two parts (as in lili_extract_rot.hip): what depends on b alone — the angle and its sine, the same for every point of a scan,
// computed once per workgroup (round 6: every point paid an acos and a sin for them) — and what depends on t.
struct LvSlerpConst { double th, sn; int linear; };
__device__ __forceinline__ LvSlerpConst lv_slerp_prepare(dq b) {
    const double one = 1.0 - 2.220446049250313e-16;
    const double absD = fabs(b.w);
    LvSlerpConst c{0.0, 1.0, 1};
    if (!(absD >= one)) { c.th = acos(absD); c.sn = sin(c.th); c.linear = 0; }
    return c;
}
__device__ __forceinline__ dq lv_slerp_identity(double t, dq b, LvSlerpConst c) {
    double s0, s1;
    if (c.linear) { s0 = 1.0 - t; s1 = t; }
    else { s0 = sin((1.0 - t) * c.th) / c.sn; s1 = sin(t * c.th) / c.sn; }
    if (b.w < 0) s1 = -s1;
    return dq{s0 + s1 * b.w, s1 * b.x, s1 * b.y, s1 * b.z};
}

__global__ void k_livox_init(int* __restrict__ owner, LivoxState* st) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kLvCells) owner[i] = 0x7fffffff;
    if (i == 0) { st->n_cut = 0; st->n_edge = 0; st->n_surf = 0; }
}

// (reads the caller's rows as they are — x, y, z at the start, intensity and curvature at their byte offsets — : no layout conversion launches in front)
__global__ __launch_bounds__(256) void k_livox_prep(const unsigned char* __restrict__ raw, int stride, int off_intensity, int off_curvature, int n,
                                                    LivoxDev P, float4* __restrict__ und, float* __restrict__ curv, unsigned char* __restrict__ keep,
                                                    int* __restrict__ owner, int* __restrict__ blk_keep /*[gridDim.x]: kept points of this block (for k_livox_cut)*/) {
    __shared__ LvSlerpConst slerp_s;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    float c = 0.f;
    if (threadIdx.x == blockDim.x - 1) slerp_s = lv_slerp_prepare(dq{P.q_imu[0], P.q_imu[1], P.q_imu[2], P.q_imu[3]});      // (published by the barrier of the count below)
    if (live) {
        const unsigned char* row = raw + (size_t)i * (size_t)stride;
        const float* xyz = reinterpret_cast<const float*>(row);
        p = make_float4(xyz[0], xyz[1], xyz[2], *reinterpret_cast<const float*>(row + off_intensity));
        c = *reinterpret_cast<const float*>(row + off_curvature);
    }
    bool ok = live && isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && !(p.x * p.x + p.y * p.y + p.z * p.z < P.near_thres * P.near_thres);   // L:225-226
    int scan_id = (int)p.w;                                                                                                          // L:252
    ok = ok && scan_id >= 0 && scan_id < kLvLines;   // lines >= 6 would index mat[] out of bounds in the reference
    if (live) keep[i] = ok;
    const int nk = __syncthreads_count(ok ? 1 : 0);
    if (threadIdx.x == 0) blk_keep[blockIdx.x] = nk;
    if (!ok) return;
    // undistortion, L:104-127
    double dt_i = (double)(p.w - (float)scan_id);
    double ratio = dt_i / 0.1;
    if (ratio >= 1.0) ratio = 1.0;
    dq qs = lv_slerp_identity(ratio, dq{P.q_imu[0], P.q_imu[1], P.q_imu[2], P.q_imu[3]}, slerp_s);
    d3 r = qrot(qs, d3{(double)p.x, (double)p.y, (double)p.z});
    float ux = (float)r.x, uy = (float)r.y, uz = (float)r.z;
    und[i] = make_float4(ux, uy, uz, p.w);
    curv[i] = c;
    double dep = (double)(ux * ux + uy * uy + uz * uz);                                            // float expression widened (L:259)
    if (dep > 40000.0 || dep < 4.0 || (double)c < 0.05 || (double)c > 25.45) return;              // L:260
    const double t_interval = 0.1 / (double)(kLvCols - 1);
    int col = (int)round((double)(p.w - (float)scan_id) / t_interval);                            // L:262
    if (col >= kLvCols || col < 0) return;
    atomicMin(&owner[scan_id * kLvCols + col], i);                                                 // first point of the stream wins (L:265-267)
}

// ordered compaction: cutted[rank] = deskewed point, cut_src[rank] = i.  Same grid as k_livox_prep (256 points per workgroup), which
// left the number of kept points per workgroup: every workgroup adds up the counts in front of it and ranks its own 256 points —
// no serial pass over the scan (round 1: ONE workgroup, 24 rounds of two barriers, 30 us).
__device__ void livox_cut_block(int block, int n_blocks, const float4* __restrict__ und, const float* __restrict__ curv, const unsigned char* __restrict__ keep, int n,
                                const int* __restrict__ blk_keep, float4* __restrict__ cut_a, float4* __restrict__ cut_b, int* __restrict__ cut_src, LivoxState* st) {
    __shared__ int ws[5];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int before = 0, total = 0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) { const int cnt = blk_keep[b]; total += cnt; if (b < block) before += cnt; }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); total += __shfl_xor(total, o); }
    if (lane == 0) { ws[wave] = before; }
    __syncthreads();
    if (threadIdx.x == 0) s_base = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    if (block == 0) {      // n_cut: the grand total
        if (lane == 0) ws[wave] = total;
        __syncthreads();
        if (threadIdx.x == 0) st->n_cut = ws[0] + ws[1] + ws[2] + ws[3];
        __syncthreads();
    }
    const int i = block * 256 + threadIdx.x;
    const int f = i < n ? (int)keep[i] : 0;
    int inc = f;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += ws[w];
    if (f) {
        const int pos = s_base + base + inc - 1;
        float4 u = und[i];
        cut_a[pos] = make_float4(u.x, u.y, u.z, 0.f);          // x, y, z, normal_x
        cut_b[pos] = make_float4(0.f, 0.f, u.w, curv[i]);       // normal_y, normal_z, intensity, curvature
        cut_src[pos] = i;
    }
}

// (also re-arms the ownership table for the NEXT scan — every cell is read exactly once, here — and clears n_cut of an empty scan: no init launch per scan)
// ONE launch for the ordered cut (workgroups [0, n_cut_blocks)) and the grid (the workgroups behind them): both only need k_livox_prep's products.
__global__ __launch_bounds__(256) void k_livox_cut_grid(int n_cut_blocks, int* __restrict__ owner, const float4* __restrict__ und, const float* __restrict__ curv,
                                                        const unsigned char* __restrict__ keep, const int* __restrict__ blk_keep, float4* __restrict__ cut_a,
                                                        float4* __restrict__ cut_b, int* __restrict__ cut_src,
                                                        float4* __restrict__ cell_pt, float* __restrict__ cell_curv, int* __restrict__ cell_src, int n, LivoxState* st) {
    if ((int)blockIdx.x < n_cut_blocks) { livox_cut_block((int)blockIdx.x, n_cut_blocks, und, curv, keep, n, blk_keep, cut_a, cut_b, cut_src, st); return; }
    int c = ((int)blockIdx.x - n_cut_blocks) * blockDim.x + threadIdx.x;
    if (c == 0 && n == 0) st->n_cut = 0;
    if (c >= kLvCells) return;
    int o = owner[c];
    owner[c] = 0x7fffffff;
    if (o == 0x7fffffff) { cell_pt[c] = make_float4(0.f, 0.f, 0.f, 0.f); cell_curv[c] = 0.f; cell_src[c] = -1; }
    else { cell_pt[c] = und[o]; cell_curv[c] = curv[o]; cell_src[c] = o; }
}

__device__ __forceinline__ double lv_depth(const float4* __restrict__ cell_pt, int k, int c) {   // getDepth: float sqrt, widened
    float4 p = cell_pt[k * kLvCols + c];
    return (double)sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
}

// value of lane `src` (wave-uniform index) as a double on every lane
__device__ __forceinline__ double bcast_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// per-block outputs: edges [block][6] and surfs [block][36] as cell ids + the shared direction / normal.
// ONE WAVE per 6-column block (round 1 ran one thread per block: 664 threads on a 256-CU part, 83 us): lane o = 6 j + k
// owns cell (line k, column i + j) — the order in which the reference visits the 36 cells (L:271-296: j outer, k inner) — so all
// loads are one round trip.  The f64 sums keep the reference's ORDER: every lane forms its own term, then the terms are added lane
// by lane (v_readlane broadcast), i.e. exactly the sequence of additions the serial loop performs; empty cells contribute +0.0, which
// leaves an accumulator that is never -0.0 unchanged.  The two 3x3 eigen-decompositions of a block run side by side, one per half of the wave.
__global__ __launch_bounds__(64) void k_livox_blocks(const float4* __restrict__ cell_pt, const float* __restrict__ cell_curv, LivoxDev P,
                                                     int* __restrict__ blk_nedge, int* __restrict__ blk_edge_cell, float* __restrict__ blk_edge_dir,
                                                     int* __restrict__ blk_nsurf, int* __restrict__ blk_surf_cell, float* __restrict__ blk_surf_nrm) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = 5 + 6 * b;
    const bool cell = lane < 36;
    const int j = cell ? lane / 6 : 0, k = cell ? lane % 6 : 0;
    const int c = k * kLvCols + i + j;
    const float cv = cell ? cell_curv[c] : 0.f;
    const float4 p = cell ? cell_pt[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    // the nine depths of this lane's Laplacian (L:310-315) are requested now as well: one memory round trip for the whole block
    float4 nb[9];
#pragma unroll
    for (int t = 0; t < 9; t++) nb[t] = cell ? cell_pt[k * kLvCols + i + j + t - 4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool valid = cell && !(cv <= 0.f);
    const unsigned long long vm = __ballot(valid);
    const int num = __popcll(vm);                                           // L:271-283
    if (lane == 0) { blk_nedge[b] = 0; blk_nsurf[b] = 0; }
    if (num < 25) return;                                                   // wave-uniform
    const double px = valid ? (double)p.x : 0.0, py = valid ? (double)p.y : 0.0, pz = valid ? (double)p.z : 0.0;
    double cx = 0, cy = 0, cz = 0;
    for (int o = 0; o < 36; o++) { cx = cx + bcast_f64(px, o); cy = cy + bcast_f64(py, o); cz = cz + bcast_f64(pz, o); }
    const double nd = (double)num;
    const d3 center{cx / nd, cy / nd, cz / nd};
    const d3 z = d3{(double)p.x, (double)p.y, (double)p.z} - center;
    const double t00 = valid ? z.x * z.x : 0.0, t01 = valid ? z.x * z.y : 0.0, t02 = valid ? z.x * z.z : 0.0;
    const double t11 = valid ? z.y * z.y : 0.0, t12 = valid ? z.y * z.z : 0.0, t22 = valid ? z.z * z.z : 0.0;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    for (int o = 0; o < 36; o++) {
        a00 += bcast_f64(t00, o); a01 += bcast_f64(t01, o); a02 += bcast_f64(t02, o);
        a11 += bcast_f64(t11, o); a12 += bcast_f64(t12, o); a22 += bcast_f64(t22, o);
    }
    // edge candidate per line (L:302-331): every lane scores its own cell, the line's winner is the FIRST column with the largest score
    double g1 = 0.0;
    if (valid) {
        double dep[9];
#pragma unroll
        for (int t = 0; t < 9; t++) dep[t] = (double)sqrtf(nb[t].x * nb[t].x + nb[t].y * nb[t].y + nb[t].z * nb[t].z);     // getDepth: float sqrt, widened
        const double d0 = dep[4];
        g1 = dep[0] + dep[1] + dep[2] + dep[3] - 8 * d0 + dep[5] + dep[6] + dep[7] + dep[8];
        g1 = g1 / (8 * d0 + 1e-3);
    }
    int ex[kLvLines], ey[kLvLines], ne = 0;
    for (int kk = 0; kk < kLvLines; kk++) {
        double max_s = 0; int idx = i;
        for (int jj = 0; jj < 6; jj++) {
            const int src = jj * 6 + kk;
            const double g = bcast_f64(g1, src);
            const bool ok = (vm >> src) & 1ull;
            if (ok && g > 0.06 && g > max_s) { max_s = g; idx = i + jj; }
        }
        if (max_s != 0) { ex[ne] = kk; ey[ne] = idx; ne++; }
    }
    // the edge candidates' scatter matrix (with <= 3 candidates the reference's test fails whatever the eigenvalues are, App. A6: no matrix, no decomposition)
    double e00 = 0, e01 = 0, e02 = 0, e11 = 0, e12 = 0, e22 = 0;
    if (ne > 3) {
        d3 ce{0, 0, 0};
        for (int q = 0; q < ne; q++) { float4 e = cell_pt[ex[q] * kLvCols + ey[q]]; ce = ce + d3{(double)e.x, (double)e.y, (double)e.z}; }
        const double ned = (double)ne;
        ce = d3{ce.x / ned, ce.y / ned, ce.z / ned};
        for (int q = 0; q < ne; q++) {
            float4 e = cell_pt[ex[q] * kLvCols + ey[q]];
            d3 zz = d3{(double)e.x, (double)e.y, (double)e.z} - ce;
            e00 += zz.x * zz.x; e01 += zz.x * zz.y; e02 += zz.x * zz.z; e11 += zz.y * zz.y; e12 += zz.y * zz.z; e22 += zz.z * zz.z;
        }
    }
    // BOTH 3x3 eigen-decompositions of the block in ONE pass of the Jacobi code (round 5): lanes 0-31 take the patch's scatter matrix, lanes 32-63 the edge
    // candidates' (a diagonal zero matrix when there is none: the sweep loop leaves at once) — the same instructions on different operands, so the launch's slowest
    // waves (blocks with edge candidates on more than three lines) run one decomposition's worth of dependent divisions and square roots instead of two.
    const bool hi_half = lane >= 32;
    double evx[3]; d3 vmn_x, vmx_x;
    eig3_sym(hi_half ? e00 : a00, hi_half ? e01 : a01, hi_half ? e02 : a02, hi_half ? e11 : a11, hi_half ? e12 : a12, hi_half ? e22 : a22, evx, vmn_x, vmx_x);
    double ev[3], eve[3]; d3 vmin, vmx;
#pragma unroll
    for (int t = 0; t < 3; t++) { ev[t] = bcast_f64(evx[t], 0); eve[t] = bcast_f64(evx[t], 32); }
    vmin = d3{bcast_f64(vmn_x.x, 0), bcast_f64(vmn_x.y, 0), bcast_f64(vmn_x.z, 0)};
    vmx = d3{bcast_f64(vmx_x.x, 32), bcast_f64(vmx_x.y, 32), bcast_f64(vmx_x.z, 32)};
    unsigned tomb_cells = 0u;   // bit (line): the edge cell of that line is removed from this block's plane set (curvature *= -1, L:363)
    if (ne > 3) {
        if (eve[2] > P.edge_thres * eve[1]) {                                                    // L:353
            d3 u = canon_sign(vmx);
            if (lane == 0) { blk_edge_dir[3 * b] = (float)u.x; blk_edge_dir[3 * b + 1] = (float)u.y; blk_edge_dir[3 * b + 2] = (float)u.z; blk_nedge[b] = ne; }
            for (int q = 0; q < ne; q++) {
                // the reference's `curvature <= 0 && intensity <= 0` skip can never fire here: candidates have curvature > 0
                if (lane == q) blk_edge_cell[b * kLvLines + q] = ex[q] * kLvCols + ey[q];
                tomb_cells |= 1u << ex[q];   // at most one candidate per line, so the line id identifies the tombstoned cell
            }
        }
    }
    if (ev[0] < P.surf_thres * ev[1]) {                                                          // L:367
        d3 u = canon_sign(vmin);
        bool is_tomb = false;
        if ((tomb_cells >> k) & 1u) { for (int q = 0; q < ne; q++) if (ex[q] == k && ey[q] == i + j) is_tomb = true; }
        const bool emit = valid && !is_tomb;
        const unsigned long long em = __ballot(emit);
        if (emit) blk_surf_cell[b * 36 + __popcll(em & ((1ull << lane) - 1ull))] = c;          // (column, line) order = lane order
        if (lane == 0) { blk_surf_nrm[3 * b] = (float)u.x; blk_surf_nrm[3 * b + 1] = (float)u.y; blk_surf_nrm[3 * b + 2] = (float)u.z; blk_nsurf[b] = __popcll(em); }
    }
}

__global__ __launch_bounds__(1024) void k_livox_compact(const float4* __restrict__ cell_pt, const float* __restrict__ cell_curv,
                                                        const int* __restrict__ blk_nedge, const int* __restrict__ blk_edge_cell, const float* __restrict__ blk_edge_dir,
                                                        const int* __restrict__ blk_nsurf, const int* __restrict__ blk_surf_cell, const float* __restrict__ blk_surf_nrm,
                                                        float4* __restrict__ edge_a, float4* __restrict__ edge_b, int* __restrict__ edge_cell,
                                                        float4* __restrict__ surf_a, float4* __restrict__ surf_b, int* __restrict__ surf_cell, LivoxState* st) {
    __shared__ int eoff[kLvBlocks + 1], soff[kLvBlocks + 1];
    __shared__ int ws[2][17];
    {   // exclusive prefix of the per-block counts: thread t = block t (664 <= 1024), one block scan for both lists
        const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
        const int e = t < kLvBlocks ? blk_nedge[t] : 0, sf = t < kLvBlocks ? blk_nsurf[t] : 0;
        int ie = e, is = sf;
        for (int o = 1; o < 64; o <<= 1) { int a = __shfl_up(ie, o), c2 = __shfl_up(is, o); if (lane >= o) { ie += a; is += c2; } }
        if (lane == 63) { ws[0][wave] = ie; ws[1][wave] = is; }
        __syncthreads();
        int be = 0, bs = 0, te = 0, ts = 0;
        for (int w = 0; w < 16; w++) { const int a = ws[0][w], c2 = ws[1][w]; if (w < wave) { be += a; bs += c2; } te += a; ts += c2; }
        if (t < kLvBlocks) { eoff[t] = be + ie - e; soff[t] = bs + is - sf; }
        if (t == 0) { eoff[kLvBlocks] = te; soff[kLvBlocks] = ts; if (blockIdx.x == 0) { st->n_edge = te; st->n_surf = ts; } }
    }
    __syncthreads();
    // every workgroup repeats the (tiny) prefix above and then copies the lists of ITS 16 blocks, one wave per block — a single
    // workgroup walking all 664 blocks spent 42 dependent rounds on it (50 us)
    for (int b = (int)blockIdx.x * 16 + (threadIdx.x >> 6); b < kLvBlocks; b += (int)gridDim.x * 16) {
        int lane = threadIdx.x & 63;
        if (lane < blk_nedge[b]) {
            int c = blk_edge_cell[b * kLvLines + lane];
            float4 p = cell_pt[c];
            int o = eoff[b] + lane;
            edge_a[o] = make_float4(p.x, p.y, p.z, blk_edge_dir[3 * b]);
            edge_b[o] = make_float4(blk_edge_dir[3 * b + 1], blk_edge_dir[3 * b + 2], p.w, cell_curv[c]);
            edge_cell[o] = c;
        }
        if (lane < blk_nsurf[b]) {
            int c = blk_surf_cell[b * 36 + lane];
            float4 p = cell_pt[c];
            int o = soff[b] + lane;
            surf_a[o] = make_float4(p.x, p.y, p.z, blk_surf_nrm[3 * b]);
            surf_b[o] = make_float4(blk_surf_nrm[3 * b + 1], blk_surf_nrm[3 * b + 2], p.w, cell_curv[c]);
            surf_cell[o] = c;
        }
    }
}

// (x,y,z,nx | ny,nz,intensity,curvature) pairs -> packed 8-float records or pcl::PointXYZINormal (48 B)
// (the list length is read on the device — n_dev — so that the three lists are packed BEFORE the host has seen the counts; cap = the caller's capacity)
__global__ void k_livox_pack(const float4* __restrict__ a, const float4* __restrict__ b, const int* __restrict__ n_dev, int cap, int pcl_layout, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*n_dev, cap)) return;
    float4 u = a[i], v = b[i];
    // 16-byte stores (every destination — the library's pack buffers, a caller's page-locked buffer written across PCIe — is 16-byte aligned)
    if (pcl_layout) {
        float4* o = reinterpret_cast<float4*>(out + (size_t)i * 12);
        o[0] = make_float4(u.x, u.y, u.z, 1.f); o[1] = make_float4(u.w, v.x, v.y, 0.f); o[2] = make_float4(v.z, v.w, 0.f, 0.f);
    } else {
        float4* o = reinterpret_cast<float4*>(out + (size_t)i * 8);
        o[0] = u; o[1] = v;
    }
}

__global__ void k_livox_xyzc(const float4* __restrict__ a, const float4* __restrict__ b, int n, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 u = a[i]; out[i] = make_float4(u.x, u.y, u.z, b[i].w); }
}

}  // namespace lili

namespace lili_detail {
struct LivoxBuffers {
    DevBuf und, curv, keep, owner, state, cut_a, cut_b, cut_src, cell_pt, cell_curv, cell_src, blk_keep;
    DevBuf blk_nedge, blk_edge_cell, blk_edge_dir, blk_nsurf, blk_surf_cell, blk_surf_nrm;
    DevBuf edge_a, edge_b, edge_cell, surf_a, surf_b, surf_cell, pack, pack_e, pack_s, xyzc_edge, xyzc_surf;
    lili::LivoxState host{};
    int* h_counts = nullptr;          // page-locked {n_cut, n_edge, n_surf}, written by k_livox_pack3; h_counts_dev: the same memory as the device sees it
    int* h_counts_dev = nullptr;
    bool have = false;
    unsigned long long pending_gen = 0;      // ctx->readback_gen when the counts' read-back joined the pending list
    bool pending = false;    // lili_extract_livox_enqueue: the counts' read-back is on the stream, lili_extract_livox_complete has not taken it yet
    bool armed = false;      // the ownership table holds "no owner" everywhere (k_livox_init once, k_livox_grid after every scan)
    void release() {
        for (DevBuf* b : {&und, &curv, &keep, &owner, &state, &cut_a, &cut_b, &cut_src, &cell_pt, &cell_curv, &cell_src, &blk_nedge, &blk_edge_cell,
                          &blk_edge_dir, &blk_nsurf, &blk_surf_cell, &blk_surf_nrm, &edge_a, &edge_b, &edge_cell, &surf_a, &surf_b, &surf_cell, &pack, &pack_e, &pack_s, &blk_keep, &xyzc_edge, &xyzc_surf}) b->release();
        if (h_counts) { (void)hipHostFree(h_counts); h_counts = nullptr; h_counts_dev = nullptr; }
    }
};
}  // namespace lili_detail

static lili_detail::LivoxBuffers* livox_of(lili_ctx* ctx) {
    if (!ctx->ext_livox) { ctx->ext_livox = new lili_detail::LivoxBuffers(); ctx->ext_livox_free = [](void* p) { auto* r = static_cast<lili_detail::LivoxBuffers*>(p); r->release(); delete r; }; }
    return static_cast<lili_detail::LivoxBuffers*>(ctx->ext_livox);
}

// the edge and the surf list in ONE launch (workgroups [0, nb0): the first list)
__device__ __forceinline__ void livox_pack_one(int i, const float4* __restrict__ a, const float4* __restrict__ b, int n, int pcl_layout, float* __restrict__ out) {
    if (i >= n) return;
    float4 u = a[i], v = b[i];
    // 16-byte stores (every destination — the library's pack buffers, a caller's page-locked buffer written across PCIe — is 16-byte aligned)
    if (pcl_layout) {
        float4* o = reinterpret_cast<float4*>(out + (size_t)i * 12);
        o[0] = make_float4(u.x, u.y, u.z, 1.f); o[1] = make_float4(u.w, v.x, v.y, 0.f); o[2] = make_float4(v.z, v.w, 0.f, 0.f);
    } else {
        float4* o = reinterpret_cast<float4*>(out + (size_t)i * 8);
        o[0] = u; o[1] = v;
    }
}
__global__ void k_livox_pack2(int nb0, const float4* __restrict__ a0, const float4* __restrict__ b0, const int* __restrict__ n0_dev, int cap0, int layout0, float* __restrict__ out0,
                              const float4* __restrict__ a1, const float4* __restrict__ b1, const int* __restrict__ n1_dev, int cap1, int layout1, float* __restrict__ out1) {
    if ((int)blockIdx.x < nb0) livox_pack_one(blockIdx.x * blockDim.x + threadIdx.x, a0, b0, min(*n0_dev, cap0), layout0, out0);
    else livox_pack_one(((int)blockIdx.x - nb0) * blockDim.x + threadIdx.x, a1, b1, min(*n1_dev, cap1), layout1, out1);
}

// all three lists in ONE launch (workgroups [0, nbc): lidar_cloud_cutted, [nbc, nbc + nb0): the edge list, then the surf list) — round 4: straight into the caller's
// page-locked buffers
__global__ void k_livox_pack3(int nbc, const float4* __restrict__ ac, const float4* __restrict__ bc, const int* __restrict__ nc_dev, int capc, int layoutc, float* __restrict__ outc,
                              int nb0, const float4* __restrict__ a0, const float4* __restrict__ b0, const int* __restrict__ n0_dev, int cap0, int layout0, float* __restrict__ out0,
                              const float4* __restrict__ a1, const float4* __restrict__ b1, const int* __restrict__ n1_dev, int cap1, int layout1, float* __restrict__ out1,
                              int* __restrict__ counts_host /*page-locked: {n_cut, n_edge, n_surf} for the host, no copy launch before the synchronisation*/) {
    const int b = (int)blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && counts_host) { counts_host[0] = *nc_dev; counts_host[1] = *n0_dev; counts_host[2] = *n1_dev; }
    if (b < nbc) livox_pack_one(b * blockDim.x + threadIdx.x, ac, bc, min(*nc_dev, capc), layoutc, outc);
    else if (b < nbc + nb0) livox_pack_one((b - nbc) * blockDim.x + threadIdx.x, a0, b0, min(*n0_dev, cap0), layout0, out0);
    else livox_pack_one((b - nbc - nb0) * blockDim.x + threadIdx.x, a1, b1, min(*n1_dev, cap1), layout1, out1);
}

// Packs one list into the caller's record layout (enqueued before the counts are read back; `bound` = an upper bound of the list length)...
static int livox_pack(lili_ctx* ctx, DevBuf& pack, const lili_feature_out* o, const float4* a, const float4* b, const int* d_count, size_t bound) {
    if (!o || !o->data) return LILI_OK;
    const size_t k = std::min(bound, o->capacity);
    if (k == 0) return LILI_OK;
    const size_t stride = o->stride ? o->stride : 32;
    ARGCHK(stride == 32 || stride == 48, "feature_out: Livox records are 32 B (packed x,y,z,nx,ny,nz,intensity,curvature) or 48 B (pcl::PointXYZINormal)");
    HIPCHK(pack.ensure(k * stride));
    hipLaunchKernelGGL(k_livox_pack, dim3(nblocks((int64_t)k, 256)), dim3(256), 0, ctx->stream, a, b, d_count, (int)k, stride == 48 ? 1 : 0, pack.as<float>());
    HIPCHK(hipGetLastError());
    return LILI_OK;
}
// ... and copies it out once the count is known (async).
static int livox_copy_out(lili_ctx* ctx, DevBuf& pack, const lili_feature_out* o, size_t count) {
    if (!o || !o->data || count == 0) return LILI_OK;
    const size_t k = std::min(count, o->capacity);
    if (k == 0) return LILI_OK;
    const size_t stride = o->stride ? o->stride : 32;
    HIPCHK(hipMemcpyAsync(o->data, pack.p, k * stride, o->mem == LILI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return LILI_OK;
}

static int extract_livox_impl(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params,
                              lili_feature_out* cutted, lili_feature_out* edge, lili_feature_out* surf, bool defer);

// Internal (lili_ctx.h -> lili_pipeline.hip): the extraction ENQUEUED only — kernels and the read-back of its counts are on the stream, nothing is waited for; the features
// stay in the extractor's device lists.  lili_extract_livox_complete takes the counts once a synchronisation of the context's stream has delivered them (it
// synchronises itself if none has) — lili_frontend_frame puts the previous frame's local-map commit, which synchronises anyway, between the two.
int lili_extract_livox_enqueue(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params) {
    return extract_livox_impl(ctx, scan, curvature_offset, q_imu, params, nullptr, nullptr, nullptr, true);
}
int lili_extract_livox_complete(lili_ctx* ctx) {
    auto* B = livox_of(ctx);
    if (!B->pending) return ctx->fail(LILI_E_STATE, "extract_livox_complete: nothing enqueued");
    B->pending = false;
    if (ctx->readback_gen == B->pending_gen) { const int rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }      // (else a synchronisation of the caller's has delivered the counts)
    B->have = true;
    return LILI_OK;
}

extern "C" {

int lili_extract_livox(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params,
                       lili_feature_out* cutted, lili_feature_out* edge, lili_feature_out* surf) {
    return extract_livox_impl(ctx, scan, curvature_offset, q_imu, params, cutted, edge, surf, false);
}

}  // extern "C"

static int extract_livox_impl(lili_ctx* ctx, const lili_cloud* scan, int curvature_offset, const double q_imu[4], const lili_livox_params* params,
                              lili_feature_out* cutted, lili_feature_out* edge, lili_feature_out* surf, bool defer) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(scan && q_imu && params, "extract_livox: null argument");
    ARGCHK(scan->aux_offset >= 0 && curvature_offset >= 0 && (size_t)curvature_offset + 4 <= scan->stride, "extract_livox: intensity (aux_offset) and curvature offsets are required");
    HIPCHK(hipSetDevice(ctx->device));
    auto* B = livox_of(ctx);
    B->have = false; B->pending = false;
    ARGCHK(scan->n == 0 || scan->data, "cloud: null data");
    ARGCHK(scan->stride >= 12 && scan->stride % 4 == 0 && (size_t)scan->aux_offset + 4 <= scan->stride, "cloud: stride must be a multiple of 4 and >= 12, offsets inside the point");
    ARGCHK(scan->n < (size_t)1 << 31, "cloud: too many points");
    ARGCHK(scan->mem == LILI_MEM_HOST || scan->mem == LILI_MEM_DEVICE, "cloud: bad mem");
    int rc = LILI_OK;
    const int n = (int)scan->n;
    const unsigned char* raw = static_cast<const unsigned char*>(scan->data);
    if (scan->mem == LILI_MEM_HOST && n > 0) {
        // A PAGE-LOCKED scan (a driver's DMA buffer, lili_host_alloc) is read by k_livox_prep where it lies, across PCIe (round 4: the copy engine's transfer + the
        // ~11 us before a kernel sees its completion were 28 us of the call; the kernel reads the same 1.15 MB in its own time).  Pageable memory: ONE transfer of the
        // rows as they are; k_livox_prep picks the fields.
        void* d = lili_pinned_dev_ptr(scan->data, 4);
        if (d) raw = static_cast<const unsigned char*>(d);
        else {
            HIPCHK(ctx->staging.ensure(scan->n * scan->stride));
            HIPCHK(hipMemcpyAsync(ctx->staging.p, scan->data, scan->n * scan->stride, hipMemcpyHostToDevice, ctx->stream));
            raw = ctx->staging.as<unsigned char>();
        }
    }
    HIPCHK(B->state.ensure(sizeof(LivoxState))); HIPCHK(B->owner.ensure(kLvCells * 4));
    HIPCHK(B->cell_pt.ensure(kLvCells * 16)); HIPCHK(B->cell_curv.ensure(kLvCells * 4)); HIPCHK(B->cell_src.ensure(kLvCells * 4));
    HIPCHK(B->blk_nedge.ensure(kLvBlocks * 4)); HIPCHK(B->blk_edge_cell.ensure(kLvBlocks * kLvLines * 4)); HIPCHK(B->blk_edge_dir.ensure(kLvBlocks * 12));
    HIPCHK(B->blk_nsurf.ensure(kLvBlocks * 4)); HIPCHK(B->blk_surf_cell.ensure(kLvBlocks * 36 * 4)); HIPCHK(B->blk_surf_nrm.ensure(kLvBlocks * 12));
    HIPCHK(B->edge_a.ensure(kLvCells * 16)); HIPCHK(B->edge_b.ensure(kLvCells * 16)); HIPCHK(B->edge_cell.ensure(kLvCells * 4));
    HIPCHK(B->surf_a.ensure(kLvCells * 16)); HIPCHK(B->surf_b.ensure(kLvCells * 16)); HIPCHK(B->surf_cell.ensure(kLvCells * 4));
    const size_t cap = (size_t)std::max(n, 1);
    HIPCHK(B->und.ensure(cap * 16)); HIPCHK(B->curv.ensure(cap * 4)); HIPCHK(B->keep.ensure(cap));
    HIPCHK(B->cut_a.ensure(cap * 16)); HIPCHK(B->cut_b.ensure(cap * 16)); HIPCHK(B->cut_src.ensure(cap * 4));
    LivoxState* st = B->state.as<LivoxState>();
    // A deferred extraction of a frame (lili_pipeline.hip) on a stream of its own: its kernels then run NEXT TO the ring merge the frame enqueues on the context's stream
    // right behind this call; the context's stream waits for them at its next read-back (lili_readback_finish), before anything reads the lists or the counts.
    if (ctx->extract_join_pending) {      // (a frame that ended before its read-back — an error path: the previous extraction may still run on the side stream, and this one reuses its buffers)
        ctx->extract_join_pending = false;
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->join_ev[lili_ctx::kExtractSide], 0));
    }
    hipStream_t xs = ctx->stream;
    const bool on_side = defer && ctx->extract_side_next && n > 0 && raw != ctx->staging.as<unsigned char>() && !cutted && !edge && !surf;
    ctx->extract_side_next = false;
    if (on_side) {
        constexpr int kS = lili_ctx::kExtractSide;
        if (!ctx->side[kS]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[kS], hipStreamNonBlocking));
        if (!ctx->join_ev[kS]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[kS], hipEventDisableTiming));
        if (!ctx->extract_fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->extract_fork_ev, hipEventDisableTiming));
        xs = ctx->side[kS];
        HIPCHK(hipEventRecord(ctx->extract_fork_ev, ctx->stream));      // (behind whatever the previous frame left on the context's stream: its ring push)
        HIPCHK(hipStreamWaitEvent(xs, ctx->extract_fork_ev, 0));
    }
    LivoxDev P{};
    for (int i = 0; i < 4; i++) P.q_imu[i] = q_imu[i];
    P.surf_thres = params->surf_thres; P.edge_thres = params->edge_thres; P.near_thres = params->near_range;
    if (!B->armed) { hipLaunchKernelGGL(k_livox_init, dim3(nblocks(kLvCells, 256)), dim3(256), 0, xs, B->owner.as<int>(), st); }   // first scan (or after a failed call) only: k_livox_grid re-arms the table
    B->armed = false;
    if (n > 0) {
        HIPCHK(B->blk_keep.ensure((size_t)nblocks(n, 256) * 4));
        hipLaunchKernelGGL(k_livox_prep, dim3(nblocks(n, 256)), dim3(256), 0, xs, raw, (int)scan->stride, (int)scan->aux_offset, curvature_offset, n, P, B->und.as<float4>(),
                           B->curv.as<float>(), B->keep.as<unsigned char>(), B->owner.as<int>(), B->blk_keep.as<int>());
    }
    const int n_cut_blocks = n > 0 ? nblocks(n, 256) : 0;
    hipLaunchKernelGGL(k_livox_cut_grid, dim3(n_cut_blocks + nblocks(kLvCells, 256)), dim3(256), 0, xs, n_cut_blocks, B->owner.as<int>(), B->und.as<float4>(), B->curv.as<float>(),
                       B->keep.as<unsigned char>(), B->blk_keep.as<int>(), B->cut_a.as<float4>(), B->cut_b.as<float4>(), B->cut_src.as<int>(),
                       B->cell_pt.as<float4>(), B->cell_curv.as<float>(), B->cell_src.as<int>(), n, st);
    B->armed = true;
    // lidar_cloud_cutted is final here: it is packed now and travels to the host on a side stream under the grid / block / compaction kernels (768 KB for
    // a 24 k-point scan).  All min(n, capacity) records travel — the count is known only at the end; records behind `count` are unspecified.
    // Round 4: the host's stream / event calls for that copy (~15 us of API time) are issued AFTER the block and compaction kernels have been launched — the event that
    // releases the copy is recorded here, the copy itself is enqueued below while the GPU already works on the blocks.
    // All three outputs in page-locked host memory (what a node that publishes them would use): ONE packing launch writes them across PCIe after the compaction — no
    // staging copies, no side stream, no count round trip (k_livox_pack3).
    const auto out_ok = [](const lili_feature_out* o) { return o && o->data && o->capacity > 0 && o->mem == LILI_MEM_HOST && (o->stride == 0 || o->stride == 32 || o->stride == 48); };
    if (!B->h_counts) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&B->h_counts), 4 * sizeof(int), hipHostMallocDefault));
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, B->h_counts, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
        B->h_counts_dev = static_cast<int*>(d);
    }
    void *dc = nullptr, *de3 = nullptr, *ds3 = nullptr;
    bool all3 = n > 0 && out_ok(cutted) && out_ok(edge) && out_ok(surf);
    if (all3) { dc = lili_pinned_dev_ptr(cutted->data, 16); de3 = lili_pinned_dev_ptr(edge->data, 16); ds3 = lili_pinned_dev_ptr(surf->data, 16); all3 = dc && de3 && ds3; }
    bool cut_early = false;
    // Whatever way this call ends, no DMA into the caller's `cutted` buffer may outlive it (lili_extract_rot has the same guard): every return between
    // the side-stream copy and its join — a HIP error, a stride the packing rejects, a failed read-back — drains the side stream first.
    struct DrainSide { hipStream_t s = nullptr; ~DrainSide() { if (s) (void)hipStreamSynchronize(s); } } drain_side;
    if (!all3 && n > 0 && cutted && cutted->data && cutted->mem == LILI_MEM_HOST && cutted->capacity > 0) {
        rc = livox_pack(ctx, B->pack, cutted, B->cut_a.as<float4>(), B->cut_b.as<float4>(), &st->n_cut, (size_t)n);
        if (rc != LILI_OK) return rc;
        if (!ctx->fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
        if (!ctx->side[1]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[1], hipStreamNonBlocking));
        if (!ctx->join_ev[1]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[1], hipEventDisableTiming));
        HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));
        cut_early = true;
    }
    hipLaunchKernelGGL(k_livox_blocks, dim3(kLvBlocks), dim3(64), 0, xs, B->cell_pt.as<float4>(), B->cell_curv.as<float>(), P,
                       B->blk_nedge.as<int>(), B->blk_edge_cell.as<int>(), B->blk_edge_dir.as<float>(), B->blk_nsurf.as<int>(), B->blk_surf_cell.as<int>(),
                       B->blk_surf_nrm.as<float>());
    if (cut_early) {          // the side stream's copy of lidar_cloud_cutted, enqueued while the GPU works on the blocks (which use no PCIe); joined before the call's ONE synchronisation
        hipError_t e = hipStreamWaitEvent(ctx->side[1], ctx->fork_ev, 0);
        drain_side.s = ctx->side[1];
        if (e == hipSuccess) e = hipMemcpyAsync(cutted->data, B->pack.p, std::min((size_t)n, cutted->capacity) * (cutted->stride ? cutted->stride : 32), hipMemcpyDeviceToHost, ctx->side[1]);
        if (e == hipSuccess) e = hipEventRecord(ctx->join_ev[1], ctx->side[1]);
        if (e != hipSuccess) return ctx->fail(LILI_E_HIP, std::string("extract_livox: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_livox_compact, dim3((kLvBlocks + 15) / 16), dim3(1024), 0, xs, B->cell_pt.as<float4>(), B->cell_curv.as<float>(), B->blk_nedge.as<int>(),
                       B->blk_edge_cell.as<int>(), B->blk_edge_dir.as<float>(), B->blk_nsurf.as<int>(), B->blk_surf_cell.as<int>(), B->blk_surf_nrm.as<float>(),
                       B->edge_a.as<float4>(), B->edge_b.as<float4>(), B->edge_cell.as<int>(), B->surf_a.as<float4>(), B->surf_b.as<float4>(), B->surf_cell.as<int>(), st);
    HIPCHK(hipGetLastError());
    if (on_side) { HIPCHK(hipEventRecord(ctx->join_ev[lili_ctx::kExtractSide], xs)); ctx->extract_join_pending = true; }
    // the three lists are packed into the caller's layout (the kernels read the counts where they lie); then the counts travel
    bool direct = false;      // the packing kernel wrote the caller's (page-locked) edge / surf buffers itself
    if (all3) {
        const size_t kc = std::min((size_t)n, cutted->capacity), k0 = std::min((size_t)kLvCells, edge->capacity), k1 = std::min((size_t)kLvCells, surf->capacity);
        const int nbc = nblocks((int64_t)kc, 256), nb0 = nblocks((int64_t)k0, 256), nb1 = nblocks((int64_t)k1, 256);
        hipLaunchKernelGGL(k_livox_pack3, dim3(nbc + nb0 + nb1), dim3(256), 0, ctx->stream,
                           nbc, B->cut_a.as<float4>(), B->cut_b.as<float4>(), &st->n_cut, (int)kc, cutted->stride == 48 ? 1 : 0, static_cast<float*>(dc),
                           nb0, B->edge_a.as<float4>(), B->edge_b.as<float4>(), &st->n_edge, (int)k0, edge->stride == 48 ? 1 : 0, static_cast<float*>(de3),
                           B->surf_a.as<float4>(), B->surf_b.as<float4>(), &st->n_surf, (int)k1, surf->stride == 48 ? 1 : 0, static_cast<float*>(ds3), B->h_counts_dev);
        if (hipGetLastError() != hipSuccess) rc = ctx->fail(LILI_E_HIP, "extract_livox: pack launch failed");
        direct = true;
    }
    if (!all3 && !cut_early) rc = livox_pack(ctx, B->pack, cutted, B->cut_a.as<float4>(), B->cut_b.as<float4>(), &st->n_cut, (size_t)n);
    const bool both = !all3 && edge && edge->data && edge->capacity && surf && surf->data && surf->capacity;
    if (all3) {}
    else if (rc == LILI_OK && both) {
        const size_t k0 = std::min((size_t)kLvCells, edge->capacity), k1 = std::min((size_t)kLvCells, surf->capacity);
        const size_t s0 = edge->stride ? edge->stride : 32, s1 = surf->stride ? surf->stride : 32;
        ARGCHK((s0 == 32 || s0 == 48) && (s1 == 32 || s1 == 48), "feature_out: Livox records are 32 B (packed x,y,z,nx,ny,nz,intensity,curvature) or 48 B (pcl::PointXYZINormal)");
        HIPCHK(B->pack_e.ensure(k0 * s0)); HIPCHK(B->pack_s.ensure(k1 * s1));
        const int nb0 = nblocks((int64_t)k0, 256), nb1 = nblocks((int64_t)k1, 256);
        // Round 4: page-locked feature buffers are written by the packing kernel ITSELF, across PCIe, exactly `count` records each: no staging copy, no count round
        // trip before the transfers, one synchronisation per call (the count round trip + a second one were ~35 us of 120).
        float* out_e = B->pack_e.as<float>(); float* out_s = B->pack_s.as<float>();
        if (edge->mem == LILI_MEM_HOST && surf->mem == LILI_MEM_HOST) {
            void* de = lili_pinned_dev_ptr(edge->data, 16); void* ds = lili_pinned_dev_ptr(surf->data, 16);
            if (de && ds) { out_e = static_cast<float*>(de); out_s = static_cast<float*>(ds); direct = true; }
        }
        hipLaunchKernelGGL(k_livox_pack2, dim3(nb0 + nb1), dim3(256), 0, ctx->stream, nb0, B->edge_a.as<float4>(), B->edge_b.as<float4>(), &st->n_edge, (int)k0, s0 == 48 ? 1 : 0,
                           out_e, B->surf_a.as<float4>(), B->surf_b.as<float4>(), &st->n_surf, (int)k1, s1 == 48 ? 1 : 0, out_s);
        if (hipGetLastError() != hipSuccess) rc = ctx->fail(LILI_E_HIP, "extract_livox: pack launch failed");
    } else {
        if (rc == LILI_OK) rc = livox_pack(ctx, B->pack_e, edge, B->edge_a.as<float4>(), B->edge_b.as<float4>(), &st->n_edge, (size_t)kLvCells);
        if (rc == LILI_OK) rc = livox_pack(ctx, B->pack_s, surf, B->surf_a.as<float4>(), B->surf_b.as<float4>(), &st->n_surf, (size_t)kLvCells);
    }
    if (rc != LILI_OK) return rc;
    const bool counts_direct = all3 && B->h_counts_dev != nullptr;      // the counts arrive with the packing kernel
    if (!counts_direct) { rc = lili_readback_add(ctx, &B->host, st, sizeof(LivoxState)); if (rc) return rc; }
    if (defer && !counts_direct && !cut_early) { B->pending = true; B->pending_gen = ctx->readback_gen; return LILI_OK; }      // (no caller buffers in this mode: nothing else to do once the counts are there)
    if (cut_early) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->join_ev[1], 0));
    size_t sent_e = 0, sent_s = 0;      // records already in the caller's buffers when the counts arrive
    auto send = [&](DevBuf& pack, const lili_feature_out* o, size_t first, size_t last) -> int {      // records [first, last) of a packed list
        if (!o || !o->data || last <= first) return LILI_OK;
        const size_t stride = o->stride ? o->stride : 32;
        HIPCHK(hipMemcpyAsync(static_cast<char*>(o->data) + first * stride, pack.as<char>() + first * stride, (last - first) * stride,
                              o->mem == LILI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        return LILI_OK;
    };
    if (direct) { sent_e = std::min((size_t)kLvCells, edge->capacity); sent_s = std::min((size_t)kLvCells, surf->capacity); }
    if (counts_direct) { HIPCHK(hipStreamSynchronize(ctx->stream)); B->host.n_cut = B->h_counts[0]; B->host.n_edge = B->h_counts[1]; B->host.n_surf = B->h_counts[2]; }
    else { const int rb = lili_readback_finish(ctx); if (rc) return rc; if (rb) return rb; }      // (the pending read is always finished)  — the call's synchronisation
    drain_side.s = nullptr;      // joined: the synchronisation above covered the side stream's copy (hipStreamWaitEvent on its join event)
    B->have = true;
    bool more = false;
    if (cutted) {
        cutted->count = (size_t)B->host.n_cut;
        if (!all3 && !cut_early && cutted->data && cutted->count) { rc = livox_copy_out(ctx, B->pack, cutted, cutted->count); if (rc) return rc; more = true; }
    }
    if (edge) {
        edge->count = (size_t)B->host.n_edge;
        const size_t want = std::min(edge->count, edge->capacity);
        if (want > sent_e) { rc = send(B->pack_e, edge, sent_e, want); if (rc) return rc; more = true; }
    }
    if (surf) {
        surf->count = (size_t)B->host.n_surf;
        const size_t want = std::min(surf->count, surf->capacity);
        if (want > sent_s) { rc = send(B->pack_s, surf, sent_s, want); if (rc) return rc; more = true; }
    }
    if (more) HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}

extern "C" {

// Intermediate products of the last lili_extract_livox (parity tests): counts = {n_cut, n_edge, n_surf};
// cut_src[n_cut], cell_src[24000] (-1 = empty), edge_cell[n_edge], surf_cell[n_surf] (cell = line * 4000 + column).
int lili_extract_livox_debug(lili_ctx* ctx, int32_t counts[3], int32_t* cut_src, int32_t* cell_src, int32_t* edge_cell, int32_t* surf_cell) {
    if (!ctx) return LILI_E_ARG;
    auto* B = livox_of(ctx);
    if (!B->have) return ctx->fail(LILI_E_STATE, "extract_livox_debug: run lili_extract_livox first");
    HIPCHK(hipSetDevice(ctx->device));
    if (counts) { counts[0] = B->host.n_cut; counts[1] = B->host.n_edge; counts[2] = B->host.n_surf; }
    auto dl = [&](void* dst, const DevBuf& src, size_t bytes) -> hipError_t { return (dst && bytes) ? hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess; };
    HIPCHK(dl(cut_src, B->cut_src, (size_t)B->host.n_cut * 4)); HIPCHK(dl(cell_src, B->cell_src, (size_t)kLvCells * 4));
    HIPCHK(dl(edge_cell, B->edge_cell, (size_t)B->host.n_edge * 4)); HIPCHK(dl(surf_cell, B->surf_cell, (size_t)B->host.n_surf * 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}


// Device views of the last lili_extract_livox results as (x, y, z, curvature) float4 arrays — the fields the
// Livox matcher consumes (L/src/BackendFusion.cpp:1608-1622); valid until the next extract on this context.
int lili_extract_livox_device(lili_ctx* ctx, lili_cloud* edge, lili_cloud* surf) { return lili_extract_livox_device_ex(ctx, edge, surf, true); }

}  // extern "C"
// (internal, lili_pipeline.hip) convert_edge = false: the edge cloud comes back with its count and NO data — a front-end frame matches surf features only, the edge
// list's conversion was a launch per frame for nothing
int lili_extract_livox_device_ex(lili_ctx* ctx, lili_cloud* edge, lili_cloud* surf, bool convert_edge) {
    if (!ctx) return LILI_E_ARG;
    auto* B = livox_of(ctx);
    if (!B->have) return ctx->fail(LILI_E_STATE, "extract_livox_device: run lili_extract_livox first");
    HIPCHK(hipSetDevice(ctx->device));
    const int ne = B->host.n_edge, ns = B->host.n_surf;
    HIPCHK(B->xyzc_edge.ensure((size_t)std::max(ne, 1) * 16)); HIPCHK(B->xyzc_surf.ensure((size_t)std::max(ns, 1) * 16));
    if (ne && convert_edge) hipLaunchKernelGGL(k_livox_xyzc, dim3(nblocks(ne, 256)), dim3(256), 0, ctx->stream, B->edge_a.as<float4>(), B->edge_b.as<float4>(), ne, B->xyzc_edge.as<float4>());
    if (ns) hipLaunchKernelGGL(k_livox_xyzc, dim3(nblocks(ns, 256)), dim3(256), 0, ctx->stream, B->surf_a.as<float4>(), B->surf_b.as<float4>(), ns, B->xyzc_surf.as<float4>());
    HIPCHK(hipGetLastError());
    if (edge) *edge = lili_cloud{convert_edge ? B->xyzc_edge.p : nullptr, (size_t)ne, 16, 12, LILI_MEM_DEVICE};
    if (surf) *surf = lili_cloud{B->xyzc_surf.p, (size_t)ns, 16, 12, LILI_MEM_DEVICE};
    return LILI_OK;
}
