// LOAM-style feature extractor of LiLi-OM-ROT on gfx950 — replaces R/src/Preprocessing.cpp:277-509
// (R/ = LiLi-OM-ROT/) behind lili_extract_rot():
//   k_rot_valid      NaN / near-range filter, first & last surviving point            (R:280-294)
//   k_rot_classify   elevation -> ring id, azimuth, `halfPassed` latch index, per-block ring histogram   (R:308-365)
//   k_rot_ring_scan  ring offsets (stable per-ring compaction = laserCloudScans[] + concatenation)        (R:371-382)
//   k_rot_scatter    relTime / intensity, IMU deskew (slerp, f64), scatter into the ring-concatenated cloud (R:367-372,153-177)
//   k_rot_curvature  11-tap curvature over the concatenated cloud, LDS-staged tile + 5-point halo         (R:385-394)
//   k_rot_select     one workgroup per ring: 6 segments rank-sorted by (curvature, index), greedy sharp /
//                    less-sharp / flat picks with +-5 neighbour suppression, less-flat list, per-ring
//                    VoxelGrid(ds_v) (bitonic sort of (voxel, index) keys in LDS, in-order centroids)     (R:401-508)
//   k_rot_compact    ordered concatenation of the per-ring lists
// Decisions are integer / f32 exact; the only transcendental inputs to a decision (atan for the ring id,
// atan2 for relTime) follow glibc's float routines statement for statement (fd_atanf / fd_atan2f below) — the
// reference build's bits; option "rot_atan" = 1 selects the f64 functions rounded to f32 instead.
// Sort ties: (curvature, index) / (voxel, index) — std::sort's order on ties is unspecified.
#include "lili_ctx.h"
#include "lili_device_math.h"
#include <cstdio>

namespace lili {

constexpr int kRotBlock = 1024;
constexpr int kMaxRings = 64;
constexpr int kSegEdgeCap = 10;   // <= 10 less-sharp picks per segment (R:425)
constexpr int kRingEdgeCap = 6 * kSegEdgeCap;
constexpr int kRingFlatCap = 6 * 4;
constexpr int kRingSharpCap = 6 * 2;

struct RotDev {
    int n_scans, ds_rate;
    int atan_mode;      // 2 = glibc fdlibm float atan / atan2 (default), 1 = f64 function rounded to f32
    float ds_v, near_thres;
    double q_imu[4], q_lb[4];
};

struct RotState {
    int first_valid, last_valid, half_idx, n_full;
    int ring_count[kMaxRings], ring_base[kMaxRings], ring_start[kMaxRings], ring_end[kMaxRings];
    int ring_nedge[kMaxRings], ring_nsharp[kMaxRings], ring_nflat[kMaxRings], ring_nlf[kMaxRings], ring_nsurf[kMaxRings];
    int n_edge, n_sharp, n_flat, n_lessflat, n_surf;
    int fallback_rings;   // rings that did not fit the LDS budget and took the global-memory path
    int redo_segments;    // segments whose concurrent greedy run had to be repeated with the previous segment's marks (diagnostics)
    long long tphase[8];  // profiling: per-phase clock ticks of ring 0's workgroup (wall_clock64)
};

// f32 atan / atan2.  Mode 2 (default): glibc's fdlibm float routines, statement for statement (sysdeps/ieee754/flt-32/s_atanf.c,
// e_atan2f.c — every glibc up to 2.40, i.e. what a build of the reference calls on the ROS releases its README names): float-only
// arithmetic, no FMA contraction, hence bit-identical to the reference build; the same statements are kept by the test checker (lo_math.h), where
// they are pinned against the image's libm on all 2^32 arguments (tools/check_fdlibm_atan.cpp).  Mode 1 ("rot_atan" = 1): the f64
// function rounded to f32 — libm-independent (glibc >= 2.41 rounds atanf correctly, like this).
__device__ float fd_atanf(float x) {
    const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                               6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const float one = 1.0f, huge = 1.0e30f;
    float w, s1, s2, z;
    int32_t ix, hx, id;
    hx = __float_as_int(x);
    ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {          /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        else return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {           /* |x| < 0.4375 */
        if (ix < 0x31000000) {       /* |x| < 2^-29 */
            if (huge + x > one) return x;
        }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {       /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}
__device__ float fd_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    int32_t k, m, hx, hy, ix, iy;
    hx = __float_as_int(x); ix = hx & 0x7fffffff;
    hy = __float_as_int(y); iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return fd_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            case 3: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                case 3: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                case 3: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fd_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}
__device__ __forceinline__ float atan_r(float v, int mode) { return mode == 2 ? fd_atanf(v) : (float)atan((double)v); }
__device__ __forceinline__ float atan2_r(float y, float x, int mode) { return mode == 2 ? fd_atan2f(y, x) : (float)atan2((double)y, (double)x); }

__device__ __forceinline__ dq qslerp_identity(double t, dq b) {   // Eigen 3.3 slerp of Identity towards b
    const double one = 1.0 - 2.220446049250313e-16;
    double d = b.w;
    double absD = fabs(d);
    double s0, s1;
    if (absD >= one) { s0 = 1.0 - t; s1 = t; }
    else {
        double theta = acos(absD);
        double sinTheta = sin(theta);
        s0 = sin((1.0 - t) * theta) / sinTheta;
        s1 = sin(t * theta) / sinTheta;
    }
    if (d < 0) s1 = -s1;
    return dq{s0 + s1 * b.w, s1 * b.x, s1 * b.y, s1 * b.z};
}

__global__ void k_rot_init(RotState* st) {
    int t = threadIdx.x;
    if (t == 0) { st->first_valid = 0x7fffffff; st->last_valid = -1; st->half_idx = 0x7fffffff; st->n_full = 0;
                  st->n_edge = st->n_sharp = st->n_flat = st->n_lessflat = st->n_surf = 0; st->fallback_rings = 0; st->redo_segments = 0; }
    if (t < kMaxRings) { st->ring_count[t] = 0; st->ring_base[t] = 0; st->ring_start[t] = 0; st->ring_end[t] = 0;
                         st->ring_nedge[t] = st->ring_nsharp[t] = st->ring_nflat[t] = st->ring_nlf[t] = st->ring_nsurf[t] = 0; }
}

__global__ __launch_bounds__(256) void k_rot_valid(const float4* __restrict__ in, int n, float thres, unsigned char* __restrict__ valid, RotState* st) {
    int first = 0x7fffffff, last = -1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = in[i];
        bool ok = isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && !(p.x * p.x + p.y * p.y + p.z * p.z < thres * thres);   // R:131-134
        valid[i] = ok;
        if (ok) { first = min(first, i); last = max(last, i); }
    }
    for (int o = 32; o > 0; o >>= 1) { first = min(first, __shfl_xor(first, o)); last = max(last, __shfl_xor(last, o)); }
    __shared__ int sf[4], sl[4];
    if ((threadIdx.x & 63) == 0) { sf[threadIdx.x >> 6] = first; sl[threadIdx.x >> 6] = last; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int f = min(min(sf[0], sf[1]), min(sf[2], sf[3])), l = max(max(sl[0], sl[1]), max(sl[2], sl[3]));
        if (f != 0x7fffffff) atomicMin(&st->first_valid, f);
        if (l >= 0) atomicMax(&st->last_valid, l);
    }
}

__device__ __forceinline__ void start_end_ori(const float4* __restrict__ in, const RotState* st, int am, float& startOri, float& endOri) {
    float4 a = in[st->first_valid], b = in[st->last_valid];
    startOri = -atan2_r(a.y, a.x, am);                                 // R:285
    endOri = (float)((double)(-atan2_r(b.y, b.x, am)) + 2 * M_PI);       // R:286-288
    if ((double)(endOri - startOri) > 3 * M_PI) endOri = (float)((double)endOri - 2 * M_PI);
    else if ((double)(endOri - startOri) < M_PI) endOri = (float)((double)endOri + 2 * M_PI);
}

// ring id (R:315-343); returns -1 when the point is dropped
__device__ __forceinline__ int ring_of(float4 p, int n_scans, int am) {
    float at = atan_r(p.z / sqrtf(p.x * p.x + p.y * p.y), am);
    float angle = (float)((double)(at * 180.0f) / M_PI);    // float product, double division, narrowed (R:315)
    int scanID;
    if (n_scans == 16) {
        scanID = (int)((double)((angle + 15.0f) / 2.0f) + 0.5);
        if (scanID > 15 || scanID < 0) return -1;
    } else if (n_scans == 32) {
        scanID = (int)(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);
        if (scanID > 31 || scanID < 0) return -1;
    } else {
        if ((double)angle >= -8.83) scanID = (int)((double)(2.0f - angle) * 3.0 + 0.5);
        else scanID = 32 + (int)((-8.83 - (double)angle) * 2.0 + 0.5);
        if ((double)angle > 2.0 || (double)angle < -24.33 || scanID > 50 || scanID < 0) return -1;
    }
    return scanID;
}

__global__ __launch_bounds__(kRotBlock) void k_rot_classify(const float4* __restrict__ in, int n, const unsigned char* __restrict__ valid,
                                                            RotDev P, RotState* st, signed char* __restrict__ scan_id, float* __restrict__ ori_raw,
                                                            int* __restrict__ block_hist /*[nb][64]*/) {
    __shared__ int hist[kMaxRings];
    __shared__ int half_min;
    if (threadIdx.x < kMaxRings) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) half_min = 0x7fffffff;
    __syncthreads();
    if (st->last_valid < 0) { if (threadIdx.x < kMaxRings) block_hist[blockIdx.x * kMaxRings + threadIdx.x] = 0; return; }
    float startOri, endOri;
    start_end_ori(in, st, P.atan_mode, startOri, endOri);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int id = -1;
    if (i < n && valid[i]) {
        float4 p = in[i];
        id = ring_of(p, P.n_scans, P.atan_mode);
        if (id >= 0) {
            float ori = -atan2_r(p.y, p.x, P.atan_mode);                       // R:349
            ori_raw[i] = ori;
            // would this point set halfPassed if it were reached with halfPassed == false?  (R:351-357)
            float o1 = ori;
            if ((double)o1 < (double)startOri - M_PI / 2) o1 = (float)((double)o1 + 2 * M_PI);
            else if ((double)o1 > (double)startOri + M_PI * 3 / 2) o1 = (float)((double)o1 - 2 * M_PI);
            if ((double)(o1 - startOri) > M_PI) atomicMin(&half_min, i);
            atomicAdd(&hist[id], 1);
        }
    }
    if (i < n) scan_id[i] = (signed char)id;
    __syncthreads();
    if (threadIdx.x < kMaxRings) block_hist[blockIdx.x * kMaxRings + threadIdx.x] = hist[threadIdx.x];
    if (threadIdx.x == 0 && half_min != 0x7fffffff) atomicMin(&st->half_idx, half_min);
}

// exclusive prefix of the per-block ring histograms (per ring, over blocks): 1024 threads = 64 rings x 16 parts
__global__ __launch_bounds__(kRotBlock) void k_rot_ring_scan(int* __restrict__ block_hist, int nb, RotDev P, RotState* st) {
    __shared__ int part_sum[16][kMaxRings];
    __shared__ int cnt[kMaxRings];
    const int r = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int per = (nb + 15) / 16;
    const int b0 = part * per, b1 = min(nb, b0 + per);
    int run = 0;
    for (int b = b0; b < b1; b++) run += block_hist[b * kMaxRings + r];
    part_sum[part][r] = run;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < part; q++) base += part_sum[q][r];
    if (part == 15) cnt[r] = base + run;
    run = base;
    for (int b = b0; b < b1; b++) { int c = block_hist[b * kMaxRings + r]; block_hist[b * kMaxRings + r] = run; run += c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int bs = 0;
        for (int k = 0; k < kMaxRings; k++) {
            st->ring_count[k] = cnt[k]; st->ring_base[k] = bs;
            if (k < P.n_scans) { st->ring_start[k] = bs + 5; st->ring_end[k] = bs + cnt[k] - 6; }   // R:379-381
            bs += cnt[k];
        }
        st->n_full = bs;
    }
}

__global__ __launch_bounds__(kRotBlock) void k_rot_scatter(const float4* __restrict__ in, int n, const signed char* __restrict__ scan_id,
                                                           const float* __restrict__ ori_raw, RotDev P, const RotState* __restrict__ st,
                                                           const int* __restrict__ block_base, float4* __restrict__ full, int* __restrict__ full_src) {
    __shared__ int wave_hist[kRotBlock / 64][kMaxRings];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < (kRotBlock / 64) * kMaxRings; k += blockDim.x) (&wave_hist[0][0])[k] = 0;
    __syncthreads();
    if (st->last_valid < 0) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int id = i < n ? (int)scan_id[i] : -1;
    // stable rank of the point among the points of its ring inside this wave
    int rank = 0;
    unsigned long long todo = __ballot(id >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int r0 = __shfl(id, leader);
        unsigned long long m = __ballot(id == r0);
        if (id == r0) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == leader) wave_hist[wave][r0] = __popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (threadIdx.x < kMaxRings) {   // exclusive prefix over the waves of the block, per ring
        int run = 0;
        for (int w = 0; w < kRotBlock / 64; w++) { int c = wave_hist[w][threadIdx.x]; wave_hist[w][threadIdx.x] = run; run += c; }
    }
    __syncthreads();
    if (id < 0) return;
    int pos = st->ring_base[id] + block_base[blockIdx.x * kMaxRings + id] + wave_hist[wave][id] + rank;
    float startOri, endOri;
    start_end_ori(in, st, P.atan_mode, startOri, endOri);
    float ori = ori_raw[i];
    if (i <= st->half_idx) {   // halfPassed was still false when the reference reached this point (R:350-358)
        if ((double)ori < (double)startOri - M_PI / 2) ori = (float)((double)ori + 2 * M_PI);
        else if ((double)ori > (double)startOri + M_PI * 3 / 2) ori = (float)((double)ori - 2 * M_PI);
    } else {                    // R:359-365
        ori = (float)((double)ori + 2 * M_PI);
        if ((double)ori < (double)endOri - M_PI * 3 / 2) ori = (float)((double)ori + 2 * M_PI);
        else if ((double)ori > (double)endOri + M_PI / 2) ori = (float)((double)ori - 2 * M_PI);
    }
    float relTime = (ori - startOri) / (endOri - startOri);          // R:367
    float intensity = (float)((double)id + 0.1 * (double)relTime);  // R:368
    // undistortion, R:153-177
    float4 p = in[i];
    int line = (int)intensity;
    double dt_i = (double)(intensity - (float)line);
    double ratio = dt_i / 0.1;
    if (ratio >= 1.0) ratio = 1.0;
    dq qimu{P.q_imu[0], P.q_imu[1], P.q_imu[2], P.q_imu[3]}, qlb{P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]};
    dq qs = qslerp_identity(ratio, qimu);
    qs = qmul(qmul(qlb, qs), qinv(qlb));
    d3 r = qrot(qs, d3{(double)p.x, (double)p.y, (double)p.z});
    full[pos] = make_float4((float)r.x, (float)r.y, (float)r.z, intensity);
    full_src[pos] = i;
}

// 11-tap curvature (R:385-394): strictly left-to-right f32 sums, tile of 256 + 5-point halo in LDS
__global__ __launch_bounds__(256) void k_rot_curvature(const float4* __restrict__ full, const RotState* __restrict__ st, float* __restrict__ curv) {
    __shared__ float sx[256 + 10], sy[256 + 10], sz[256 + 10];
    const int n = st->n_full;
    const int base = blockIdx.x * 256;
    if (base >= n) return;
    for (int k = threadIdx.x; k < 256 + 10; k += 256) {
        int g = base - 5 + k;
        float4 p = (g >= 0 && g < n) ? full[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        sx[k] = p.x; sy[k] = p.y; sz[k] = p.z;
    }
    __syncthreads();
    int i = base + threadIdx.x;
    if (i >= n) return;
    float c = 0.f;
    if (i >= 5 && i < n - 5) {
        const int k = threadIdx.x + 5;
        float dX = sx[k - 5] + sx[k - 4] + sx[k - 3] + sx[k - 2] + sx[k - 1] - 10 * sx[k] + sx[k + 1] + sx[k + 2] + sx[k + 3] + sx[k + 4] + sx[k + 5];
        float dY = sy[k - 5] + sy[k - 4] + sy[k - 3] + sy[k - 2] + sy[k - 1] - 10 * sy[k] + sy[k + 1] + sy[k + 2] + sy[k + 3] + sy[k + 4] + sy[k + 5];
        float dZ = sz[k - 5] + sz[k - 4] + sz[k - 3] + sz[k - 2] + sz[k - 1] - 10 * sz[k] + sz[k + 1] + sz[k + 2] + sz[k + 3] + sz[k + 4] + sz[k + 5];
        c = dX * dX + dY * dY + dZ * dZ;
    }
    curv[i] = c;
}

// ------------------------------------------------------------------------------------------------
// per-ring selection
// ------------------------------------------------------------------------------------------------
constexpr int kRingLdsCap = 4096;   // ring points held in LDS (SYN: 3125, HDL-64E: ~2100, VLP-16: ~1800)

__device__ __forceinline__ float gap2(const float4* __restrict__ P, int a, int b) {   // R:435-438
    float dX = P[a].x - P[b].x, dY = P[a].y - P[b].y, dZ = P[a].z - P[b].z;
    return dX * dX + dY * dY + dZ * dZ;
}
__device__ __forceinline__ float range2(const float4* __restrict__ P, int k) { return P[k].x * P[k].x + P[k].y * P[k].y + P[k].z * P[k].z; }

constexpr int kSegEdge = 10, kSegFlat = 4;
struct RingLds {
    float4 pts[kRingLdsCap + 16];          // ring points incl. the +-5 margins used by the suppression loops
    unsigned vidx[kRingLdsCap];            // voxel index of the q-th less-flat point
    unsigned short ord_a[kRingLdsCap];     // radix-sort ping-pong: positions q in the less-flat list, ordered by (voxel, q)
    unsigned short ord_b[kRingLdsCap];
    int rcnt[32 * 4 * (kRotBlock / 64)];   // radix pass: counts / offsets [digit][slot][wave]
    float curv[kRingLdsCap + 16];
    int sort_ind[kRingLdsCap + 16];
    signed char mark[kRingLdsCap + 16 + 64];   // cloudNeighborPicked, one private stretch per segment (see greedy_segment)
    signed char label[kRingLdsCap + 16];
    int seg_edge[6][kSegEdge], seg_flat[6][kSegFlat], seg_ne[6], seg_nf[6];
    int scan[kRotBlock / 64 + 1];
    float red[6][kRotBlock / 64];
    int misc[8];
};

// The greedy picks of ONE segment (R:413-492), run by one lane.  `M` = this segment's private view of cloudNeighborPicked: M[k] for
// ring-local k in [sp - 5, ep + 5] (cleared by the caller; `spill` = bit l set <=> element sp + l was already marked by the previous
// segment's picks).  Writes the labels of its own picks, its pick lists in push order and its marks.
__device__ void greedy_segment(RingLds& L, signed char* M, int sp, int ep, unsigned spill, int j) {
    const float4* Pp = L.pts;
    for (int l = 0; l < 5; l++) if ((spill >> l) & 1u) M[sp + l] = 1;
    int ne = 0, nf = 0;
    int largest = 0;
    for (int k = ep; k >= sp; k--) {                                    // R:413-453
        int ind = L.sort_ind[k];
        if (!((double)L.curv[ind] > 2.0)) break;                        // sorted: nothing further can qualify
        if (M[ind] == 0) {
            largest++;
            if (largest <= 2) { L.label[ind] = 2; L.seg_edge[j][ne++] = ind; }
            else if (largest <= 10) { L.label[ind] = 1; L.seg_edge[j][ne++] = ind; }
            else break;
            M[ind] = 1;
            for (int l = 1; l <= 5; l++) { if ((double)gap2(Pp, ind + l, ind + l - 1) > 0.05) break; M[ind + l] = 1; }
            for (int l = -1; l >= -5; l--) { if ((double)gap2(Pp, ind + l, ind + l + 1) > 0.05) break; M[ind + l] = 1; }
        }
    }
    int smallest = 0;
    for (int k = sp; k <= ep; k++) {                                    // R:456-492
        int ind = L.sort_ind[k];
        if (!((double)L.curv[ind] < 0.1)) break;                        // sorted ascending
        if ((double)range2(Pp, ind) < 0.25) continue;
        if (M[ind] == 0) {
            L.label[ind] = -1; L.seg_flat[j][nf++] = ind;
            smallest++;
            if (smallest >= 4) break;                                   // before the suppression (R:468-470)
            M[ind] = 1;
            for (int l = 1; l <= 5; l++) { if ((double)gap2(Pp, ind + l, ind + l - 1) > 0.05) break; M[ind + l] = 1; }
            for (int l = -1; l >= -5; l--) { if ((double)gap2(Pp, ind + l, ind + l + 1) > 0.05) break; M[ind + l] = 1; }
        }
    }
    L.seg_ne[j] = ne; L.seg_nf[j] = nf;
}

#define LILI_ROT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// The same picks by a whole WAVE (all 64 lanes call it with the same arguments): the serial loops above spend their time in dependent
// LDS round trips on one lane.  Here 64 candidates of the sorted order are examined at a time (the first eligible one = first set bit of a
// ballot), and the +-5 suppression of a pick is ten gap tests on ten lanes followed by two "first break" bit scans.  Picks, labels,
// lists and marks are exactly those of greedy_segment: a pick only ever ADDS marks, the candidates are visited in the same order, and
// after every pick the eligibility of the remaining lanes is re-read.
__device__ void greedy_segment_wave(RingLds& L, signed char* M, int sp, int ep, int j) {
    const int lane = threadIdx.x & 63;
    const float4* Pp = L.pts;
    // marks the +-5 neighbourhood of `ind` up to the first gap > 0.05 on either side (R:441-452)
    auto suppress = [&](int ind) {
        bool brk = false;
        if (lane < 5) brk = (double)gap2(Pp, ind + lane + 1, ind + lane) > 0.05;                  // l = lane + 1: gap(ind + l, ind + l - 1)
        else if (lane >= 8 && lane < 13) brk = (double)gap2(Pp, ind - (lane - 8) - 1, ind - (lane - 8)) > 0.05;   // l = -(lane - 8) - 1: gap(ind + l, ind + l + 1)
        const unsigned long long bal = __ballot(brk);
        const unsigned fw = (unsigned)(bal & 31ull), bw = (unsigned)((bal >> 8) & 31ull);
        const int nf = fw ? __ffs((int)fw) - 1 : 5, nb = bw ? __ffs((int)bw) - 1 : 5;               // marks before the first break
        if (lane == 0) M[ind] = 1;
        if (lane >= 1 && lane <= nf) M[ind + lane] = 1;
        if (lane >= 16 && lane - 16 < nb) M[ind - (lane - 16) - 1] = 1;
        LILI_ROT_WAVE_SYNC();
    };
    int ne = 0, nf_ = 0, largest = 0;
    bool done = false;
    for (int k0 = ep; k0 >= sp && !done; k0 -= 64) {                        // R:413-453, 64 candidates at a time
        const int k = k0 - lane;
        const bool in = k >= sp;
        const int ind = in ? L.sort_ind[k] : 0;
        const bool big = in && (double)L.curv[ind] > 2.0;
        const unsigned long long stop = __ballot(!big);                     // first lane that ends the loop (curvature too small, or the segment's end)
        const unsigned long long live = stop ? ((stop & (0ull - stop)) - 1ull) : ~0ull;       // lanes before it
        unsigned long long todo = live;
        while (todo) {
            const unsigned long long el = __ballot(big && M[ind] == 0) & todo;
            if (!el) break;
            const int l = __ffsll((long long)el) - 1;
            const int pick = __shfl(ind, l);
            largest++;
            if (largest > 10) { done = true; break; }
            if (lane == 0) { L.label[pick] = largest <= 2 ? 2 : 1; L.seg_edge[j][ne] = pick; }
            ne++;
            suppress(pick);
            todo &= ~((2ull << l) - 1ull);                                  // lanes behind the pick
        }
        if (stop) done = true;
    }
    int smallest = 0;
    done = false;
    for (int k0 = sp; k0 <= ep && !done; k0 += 64) {                        // R:456-492
        const int k = k0 + lane;
        const bool in = k <= ep;
        const int ind = in ? L.sort_ind[k] : 0;
        const bool small = in && (double)L.curv[ind] < 0.1;
        const bool far = in && !((double)range2(Pp, ind) < 0.25);
        const unsigned long long stop = __ballot(!small);
        const unsigned long long live = stop ? ((stop & (0ull - stop)) - 1ull) : ~0ull;
        unsigned long long todo = live;
        while (todo) {
            const unsigned long long el = __ballot(small && far && M[ind] == 0) & todo;
            if (!el) break;
            const int l = __ffsll((long long)el) - 1;
            const int pick = __shfl(ind, l);
            if (lane == 0) { L.label[pick] = -1; L.seg_flat[j][nf_] = pick; }
            nf_++;
            smallest++;
            if (smallest >= 4) { done = true; break; }                      // before the suppression (R:468-470)
            suppress(pick);
            todo &= ~((2ull << l) - 1ull);
        }
        if (stop) done = true;
    }
    if (lane == 0) { L.seg_ne[j] = ne; L.seg_nf[j] = nf_; }
}
__device__ __forceinline__ int block_excl_scan_1024(int v, int* lds, int& total) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kRotBlock / 64; w++) { int s = lds[w]; if (w < wave) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// Rank sort of one segment of one ring by (curvature, index) — std::sort of R:409-410 with ties broken by index.
// One workgroup per (ring, segment, chunk of 256 elements): O(L^2/6) compares per ring spread over the whole chip
// instead of the single CU that owns the ring in k_rot_select.  sort_ind holds GLOBAL indices (into `full`).
__global__ __launch_bounds__(256) void k_rot_rank(const float* __restrict__ curv, RotDev P, const RotState* __restrict__ st, int* __restrict__ sort_ind) {
    __shared__ float seg[kRingLdsCap];
    const int ring = blockIdx.x, j = blockIdx.y;
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    if (ring >= P.n_scans || re - rs < 6 || ring % P.ds_rate != 0) return;
    const int sp = rs + (re - rs) * j / 6, ep = rs + (re - rs) * (j + 1) / 6 - 1;
    const int len = ep - sp + 1;
    if (len <= 0) return;
    if ((int)blockIdx.z * 64 >= len) return;
    const bool in_lds = len <= kRingLdsCap;
    if (in_lds) for (int m = threadIdx.x; m < len; m += 256) seg[m] = curv[sp + m];
    __syncthreads();
    if (!in_lds) {      // a segment of a ring with more than ~24 k points: the same count straight from global memory (rare; rings this long are not the product's fast case)
        const int sub = threadIdx.x & 3;
        for (int e = blockIdx.z * 64 + (threadIdx.x >> 2); e < len; e += gridDim.z * 64) {
            const float ck = curv[sp + e];
            int rank = 0;
            for (int m = sub; m < len; m += 4) { const float cm = curv[sp + m]; rank += (cm < ck || (cm == ck && m < e)) ? 1 : 0; }
            rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2);
            if (sub == 0) sort_ind[sp + rank] = sp + e;
        }
        return;
    }
    // four lanes per element, each counting a quarter of the segment (interleaved by 4): the loop is a quarter as long, the partial
    // ranks meet in two shuffles
    const int sub = threadIdx.x & 3;
    for (int e = blockIdx.z * 64 + (threadIdx.x >> 2); e < len; e += gridDim.z * 64) {
        const float ck = seg[e];
        int rank = 0;
        for (int m = sub; m < len; m += 4) { const float cm = seg[m]; rank += (cm < ck || (cm == ck && m < e)) ? 1 : 0; }
        rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2);
        if (sub == 0) sort_ind[sp + rank] = sp + e;
    }
}

__global__ __launch_bounds__(kRotBlock) void k_rot_select(const float4* __restrict__ full, const float* __restrict__ curv_g, const int* __restrict__ sort_ind_g, RotDev P, RotState* st,
                                                          int* __restrict__ label_g, int* __restrict__ ring_edge /*[64][60]*/,
                                                          int* __restrict__ ring_sharp /*[64][12]*/, int* __restrict__ ring_flat /*[64][24]*/,
                                                          int* __restrict__ lessflat_tmp /*[n]*/, float4* __restrict__ surf_tmp /*[n]*/,
                                                          int* __restrict__ surf_cnt_tmp /*[n]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RingLds& L = *reinterpret_cast<RingLds*>(smem);
    const int ring = blockIdx.x;
    const int tid = threadIdx.x;
    const int rbase = st->ring_base[ring], rcount = st->ring_count[ring];
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    // labels of every point of the ring default to 0 (R:393)
    for (int k = tid; k < rcount; k += kRotBlock) label_g[rbase + k] = 0;
    if (ring >= P.n_scans || re - rs < 6 || ring % P.ds_rate != 0) return;      // R:402
    if (rcount > kRingLdsCap) {     // does not fit the LDS working set: k_rot_select_big takes this ring (global-memory arrays)
        if (tid == 0) atomicAdd(&st->fallback_rings, 1);
        return;
    }
    // ---- stage the ring: local index l <-> global index rbase + l
    if (ring == 0 && tid == 0) st->tphase[0] = wall_clock64();
    for (int k = tid; k < rcount; k += kRotBlock) {
        L.pts[k] = full[rbase + k]; L.curv[k] = curv_g[rbase + k];
        L.label[k] = 0;
    }
    __syncthreads();
    const int s0 = rs - rbase, e0 = re - rbase;   // local scanStartInd / scanEndInd
    // ---- sorted order of every segment (k_rot_rank): global -> local indices
    if (ring == 0 && tid == 0) st->tphase[1] = wall_clock64();
    for (int k = s0 + tid; k <= e0 - 1; k += kRotBlock) L.sort_ind[k] = sort_ind_g[rbase + k] - rbase;
    __syncthreads();
    // ---- greedy picks.  The reference runs the six segments one after the other and the +-5 neighbour suppression of a pick may
    // reach across a segment border (A4 iv) — but only FORWARD matters (marks that land in an earlier segment are never read again),
    // and only through the first five elements of the next segment.  So the six segments run concurrently, one wave each, every one
    // on a private stretch of the mark array (segment j: mark[k + 10 j], k in [sp - 5, ep + 5] — the stretches do not overlap); then
    // segment j is checked against what segment j - 1 finally marked in its first five elements: if none of j's picks is among
    // them the sequential run would have picked exactly the same (a marked element is only ever skipped), otherwise j is redone
    // with those marks in place (rare: a top-10 curvature within five points of both sides of a border).
    if (ring == 0 && tid == 0) st->tphase[2] = wall_clock64();
    const bool par_seg = (e0 - s0) >= 64;      // every segment longer than the reach of a suppression (5): spills stop in the next segment
    if (!par_seg) {                             // a nearly empty ring: the reference's order, one shared mark stretch
        for (int k = s0 - 5 + tid; k <= e0 + 5; k += kRotBlock) L.mark[k] = 0;
        __syncthreads();
        if (tid == 0) for (int j = 0; j < 6; j++) greedy_segment(L, L.mark, s0 + (e0 - s0) * j / 6, s0 + (e0 - s0) * (j + 1) / 6 - 1, 0u, j);
    } else {
        const int wave = tid >> 6, lane = tid & 63;
        if (wave < 6) {
            const int j = wave;
            const int sp = s0 + (e0 - s0) * j / 6, ep = s0 + (e0 - s0) * (j + 1) / 6 - 1;
            signed char* M = L.mark + 10 * j;
            for (int k = sp - 5 + lane; k <= ep + 5; k += 64) M[k] = 0;
            LILI_ROT_WAVE_SYNC();
            greedy_segment_wave(L, M, sp, ep, j);
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int j = 1; j < 6 && par_seg; j++) {
            const int sp = s0 + (e0 - s0) * j / 6, ep = s0 + (e0 - s0) * (j + 1) / 6 - 1;
            const signed char* Mp = L.mark + 10 * (j - 1);
            unsigned spill = 0;
            for (int l = 0; l < 5 && sp + l <= ep + 5; l++) if (Mp[sp + l]) spill |= 1u << l;
            if (!spill) continue;
            bool hit = false;
            for (int q = 0; q < L.seg_ne[j]; q++) { const int d = L.seg_edge[j][q] - sp; if (d >= 0 && d < 5 && ((spill >> d) & 1u)) hit = true; }
            for (int q = 0; q < L.seg_nf[j]; q++) { const int d = L.seg_flat[j][q] - sp; if (d >= 0 && d < 5 && ((spill >> d) & 1u)) hit = true; }
            if (!hit) continue;
            for (int q = 0; q < L.seg_ne[j]; q++) L.label[L.seg_edge[j][q]] = 0;
            for (int q = 0; q < L.seg_nf[j]; q++) L.label[L.seg_flat[j][q]] = 0;
            signed char* M = L.mark + 10 * j;
            for (int k = sp - 5; k <= ep + 5; k++) M[k] = 0;
            greedy_segment(L, M, sp, ep, spill, j);
            atomicAdd(&st->redo_segments, 1);
        }
        int ne = 0, nsh = 0, nfl = 0;
        for (int j = 0; j < 6; j++) {
            for (int q = 0; q < L.seg_ne[j]; q++) {
                const int g = rbase + L.seg_edge[j][q];
                if (q < 2) ring_sharp[ring * kRingSharpCap + nsh++] = g;
                ring_edge[ring * kRingEdgeCap + ne++] = g;
            }
            for (int q = 0; q < L.seg_nf[j]; q++) ring_flat[ring * kRingFlatCap + nfl++] = rbase + L.seg_flat[j][q];
        }
        st->ring_nedge[ring] = ne; st->ring_nsharp[ring] = nsh; st->ring_nflat[ring] = nfl;
    }
    __syncthreads();
    for (int k = tid; k < rcount; k += kRotBlock) label_g[rbase + k] = L.label[k];
    // ---- less-flat list in index order (R:494-499), compacted with a block scan
    if (ring == 0 && tid == 0) st->tphase[3] = wall_clock64();
    int n_lf = 0;
    for (int k0 = s0; k0 <= e0 - 1; k0 += kRotBlock) {
        int k = k0 + tid;
        bool keep = k <= e0 - 1 && !((double)range2(L.pts, k) < 0.25) && L.label[k] <= 0;
        int tot; int off = block_excl_scan_1024(keep ? 1 : 0, L.scan, tot);
        if (keep) { lessflat_tmp[rbase + n_lf + off] = rbase + k; L.sort_ind[n_lf + off] = k; }   // sort_ind is free again: local less-flat list
        n_lf += tot;
    }
    __syncthreads();
    if (tid == 0) st->ring_nlf[ring] = n_lf;
    // ---- pcl::VoxelGrid(ds_v) on the ring's less-flat points (R:502-508; PCL >= 1.8 semantics)
    if (ring == 0 && tid == 0) st->tphase[4] = wall_clock64();
    if (n_lf == 0) { if (tid == 0) st->ring_nsurf[ring] = 0; return; }
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = tid; q < n_lf; q += kRotBlock) {
        float4 p = L.pts[L.sort_ind[q]];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { L.red[c][tid >> 6] = mn[c]; L.red[3 + c][tid >> 6] = mx[c]; }
    }
    __syncthreads();
    const float inv = 1.0f / P.ds_v;
    int min_b[3], div_b[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = L.red[c][0], b = L.red[3 + c][0];
        for (int w = 1; w < kRotBlock / 64; w++) { a = fminf(a, L.red[c][w]); b = fmaxf(b, L.red[3 + c][w]); }
        min_b[c] = (int)floorf(a * inv);
        div_b[c] = (int)floorf(b * inv) - min_b[c] + 1;
    }
    // order of the less-flat points by (voxel index, position in the list): stable LSD radix sort of the POSITIONS (16-bit) on 5-bit
    // digits of the voxel index, one workgroup, in LDS — 4 block barriers per pass and ceil(bits / 5) passes, against the 78 barriers of
    // the bitonic network on 64-bit (voxel, position) keys it replaces (55 -> ~12 us per ring).  Ranks inside a wave come from
    // five ballots (the lanes that hold the same digit), across waves / slots from one scan of the [digit][slot][wave] count table.
    for (int q = tid; q < n_lf; q += kRotBlock) {
        float4 p = L.pts[L.sort_ind[q]];
        int i0 = (int)(floorf(p.x * inv) - (float)min_b[0]);
        int i1 = (int)(floorf(p.y * inv) - (float)min_b[1]);
        int i2 = (int)(floorf(p.z * inv) - (float)min_b[2]);
        L.vidx[q] = (unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
        L.ord_a[q] = (unsigned short)q;
    }
    const unsigned n_vox = (unsigned)div_b[0] * (unsigned)div_b[1] * (unsigned)div_b[2];    // PCL itself rejects grids beyond 2^31 cells
    int bits = 1; while (bits < 32 && (n_vox - 1u) >> bits) bits++;
    unsigned short* src = L.ord_a;
    unsigned short* dst = L.ord_b;
    __syncthreads();
    {
        const int wave = tid >> 6, lane = tid & 63;
        for (int shift = 0; shift < bits; shift += 5) {
            L.rcnt[tid] = 0; L.rcnt[tid + kRotBlock] = 0;
            __syncthreads();
            int dig[4], rk[4], qq[4];
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const int i = sl * kRotBlock + tid;
                const bool act = i < n_lf;
                qq[sl] = act ? (int)src[i] : 0;
                const int d = act ? (int)((L.vidx[qq[sl]] >> shift) & 31u) : 0;
                unsigned long long m = __ballot(act);
#pragma unroll
                for (int bb = 0; bb < 5; bb++) { const unsigned long long bal = __ballot(act && ((d >> bb) & 1)); m &= ((d >> bb) & 1) ? bal : ~bal; }
                rk[sl] = __popcll(m & ((1ull << lane) - 1ull));
                dig[sl] = d;
                if (act && rk[sl] == 0) L.rcnt[(d * 4 + sl) * (kRotBlock / 64) + wave] = __popcll(m);
            }
            __syncthreads();
            {
                const int a = L.rcnt[2 * tid], b2 = L.rcnt[2 * tid + 1];
                int tot; const int ex = block_excl_scan_1024(a + b2, L.scan, tot);
                L.rcnt[2 * tid] = ex; L.rcnt[2 * tid + 1] = ex + a;
            }
            __syncthreads();
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const int i = sl * kRotBlock + tid;
                if (i < n_lf) dst[L.rcnt[(dig[sl] * 4 + sl) * (kRotBlock / 64) + wave] + rk[sl]] = (unsigned short)qq[sl];
            }
            __syncthreads();
            unsigned short* t2 = src; src = dst; dst = t2;
        }
    }
    // run heads -> output slots; each head accumulates its voxel in list order (f32, like CentroidPoint)
    if (ring == 0 && tid == 0) st->tphase[5] = wall_clock64();
    int n_out = 0;
    for (int q0 = 0; q0 < n_lf; q0 += kRotBlock) {
        int q = q0 + tid;
        bool head = q < n_lf && (q == 0 || L.vidx[src[q]] != L.vidx[src[q - 1]]);
        int tot; int off = block_excl_scan_1024(head ? 1 : 0, L.scan, tot);
        if (head) {
            unsigned vox = L.vidx[src[q]];
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f; int cnt = 0;
            for (int m = q; m < n_lf && L.vidx[src[m]] == vox; m++) {
                float4 p = L.pts[L.sort_ind[src[m]]];
                sx += p.x; sy += p.y; sz += p.z; si += p.w; cnt++;
            }
            float fn = (float)cnt;
            surf_tmp[rbase + n_out + off] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
            surf_cnt_tmp[rbase + n_out + off] = cnt;
        }
        n_out += tot;
    }
    if (tid == 0) st->ring_nsurf[ring] = n_out;
    if (ring == 0 && tid == 0) st->tphase[6] = wall_clock64();
}

// ---- rings that do not fit the LDS working set (more than kRingLdsCap = 4096 points on one ring: a 16-ring sensor at 0.1 deg, merged
// sweeps; the reference takes any ring up to its 400 000-point arrays, R/src/Preprocessing.cpp:9-12).  Same statements as k_rot_select with
// the ring's arrays in GLOBAL memory (`full`, `curv_g`, `sort_ind_g`, `label_g` in place; marks / voxel ids / sort buffers / digit table in
// a per-scan scratch area): the greedy picks run in the reference's serial order on one lane, the voxel ordering is the same stable LSD
// radix sort with the slot loop no longer unrolled.  Correct and bit-identical to the LDS path; not tuned — a ring this long costs
// ~1 ms.  One workgroup per ring; rings that fit LDS return immediately (k_rot_select handled them).
struct RotBigScratch { signed char* mark; unsigned* vidx; unsigned* ord_a; unsigned* ord_b; int* rcnt; int rcnt_off[kMaxRings]; };   // rcnt_off: start of the ring's [digit table | sort buffer] stretch
__global__ __launch_bounds__(kRotBlock) void k_rot_select_big(const float4* __restrict__ full, const float* __restrict__ curv_g, const int* __restrict__ sort_ind_g, RotDev P, RotState* st,
                                                              int* __restrict__ label_g, int* __restrict__ ring_edge, int* __restrict__ ring_sharp, int* __restrict__ ring_flat,
                                                              int* __restrict__ lessflat_tmp, float4* __restrict__ surf_tmp, int* __restrict__ surf_cnt_tmp, RotBigScratch B) {
    __shared__ int scan[kRotBlock / 64 + 1];
    __shared__ float red[6][kRotBlock / 64];
    const int ring = blockIdx.x, tid = threadIdx.x;
    const int rbase = st->ring_base[ring], rcount = st->ring_count[ring];
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    if (rcount <= kRingLdsCap) return;
    if (ring >= P.n_scans || re - rs < 6 || ring % P.ds_rate != 0) return;      // labels were zeroed by k_rot_select
    signed char* M = B.mark + rbase;                 // index = GLOBAL index - rbase
    for (int k = tid; k < rcount; k += kRotBlock) M[k] = 0;
    __syncthreads();
    if (tid == 0) {                                   // R:401-492, the reference's order
        int ne = 0, nsh = 0, nfl = 0;
        auto gap2g = [&](int a, int b) { const float4 pa = full[a], pb = full[b]; const float dX = pa.x - pb.x, dY = pa.y - pb.y, dZ = pa.z - pb.z; return dX * dX + dY * dY + dZ * dZ; };
        auto range2g = [&](int k) { const float4 p = full[k]; return p.x * p.x + p.y * p.y + p.z * p.z; };
        for (int j = 0; j < 6; j++) {
            const int sp = rs + (re - rs) * j / 6, ep = rs + (re - rs) * (j + 1) / 6 - 1;
            int largest = 0;
            for (int k = ep; k >= sp; k--) {
                const int ind = sort_ind_g[k];
                if (!((double)curv_g[ind] > 2.0)) break;
                if (M[ind - rbase] == 0) {
                    largest++;
                    if (largest <= 2) { label_g[ind] = 2; if (nsh < kRingSharpCap) ring_sharp[ring * kRingSharpCap + nsh++] = ind; if (ne < kRingEdgeCap) ring_edge[ring * kRingEdgeCap + ne++] = ind; }
                    else if (largest <= 10) { label_g[ind] = 1; if (ne < kRingEdgeCap) ring_edge[ring * kRingEdgeCap + ne++] = ind; }
                    else break;
                    M[ind - rbase] = 1;
                    for (int l = 1; l <= 5; l++) { if ((double)gap2g(ind + l, ind + l - 1) > 0.05) break; M[ind + l - rbase] = 1; }
                    for (int l = -1; l >= -5; l--) { if ((double)gap2g(ind + l, ind + l + 1) > 0.05) break; M[ind + l - rbase] = 1; }
                }
            }
            int smallest = 0;
            for (int k = sp; k <= ep; k++) {
                const int ind = sort_ind_g[k];
                if (!((double)curv_g[ind] < 0.1)) break;
                if ((double)range2g(ind) < 0.25) continue;
                if (M[ind - rbase] == 0) {
                    label_g[ind] = -1; if (nfl < kRingFlatCap) ring_flat[ring * kRingFlatCap + nfl++] = ind;
                    smallest++;
                    if (smallest >= 4) break;
                    M[ind - rbase] = 1;
                    for (int l = 1; l <= 5; l++) { if ((double)gap2g(ind + l, ind + l - 1) > 0.05) break; M[ind + l - rbase] = 1; }
                    for (int l = -1; l >= -5; l--) { if ((double)gap2g(ind + l, ind + l + 1) > 0.05) break; M[ind + l - rbase] = 1; }
                }
            }
        }
        st->ring_nedge[ring] = ne; st->ring_nsharp[ring] = nsh; st->ring_nflat[ring] = nfl;
    }
    __threadfence_block();
    __syncthreads();
    // less-flat list in index order (R:494-499); B.ord_b doubles as the list of global indices
    unsigned* lf = B.ord_b + rbase;
    int n_lf = 0;
    for (int k0 = rs; k0 <= re - 1; k0 += kRotBlock) {
        const int k = k0 + tid;
        bool keep = false;
        if (k <= re - 1) { const float4 p = full[k]; keep = !((double)(p.x * p.x + p.y * p.y + p.z * p.z) < 0.25) && label_g[k] <= 0; }
        int tot; const int off = block_excl_scan_1024(keep ? 1 : 0, scan, tot);
        if (keep) { lessflat_tmp[rbase + n_lf + off] = k; lf[n_lf + off] = (unsigned)k; }
        n_lf += tot;
    }
    __syncthreads();
    if (tid == 0) st->ring_nlf[ring] = n_lf;
    if (n_lf == 0) { if (tid == 0) st->ring_nsurf[ring] = 0; return; }
    // pcl::VoxelGrid(ds_v) (R:502-508): bounding box, voxel ids, stable radix sort of the list positions, in-order centroids
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = tid; q < n_lf; q += kRotBlock) {
        const float4 p = full[lf[q]];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { red[c][tid >> 6] = mn[c]; red[3 + c][tid >> 6] = mx[c]; }
    }
    __syncthreads();
    const float inv = 1.0f / P.ds_v;
    int min_b[3], div_b[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = red[c][0], b = red[3 + c][0];
        for (int w = 1; w < kRotBlock / 64; w++) { a = fminf(a, red[c][w]); b = fmaxf(b, red[3 + c][w]); }
        min_b[c] = (int)floorf(a * inv);
        div_b[c] = (int)floorf(b * inv) - min_b[c] + 1;
    }
    unsigned* vidx = B.vidx + rbase;
    unsigned* src = B.ord_a + rbase;
    unsigned* dst = B.ord_a + rbase;                  // set below (ping-pong with a second stretch)
    for (int q = tid; q < n_lf; q += kRotBlock) {
        const float4 p = full[lf[q]];
        const int i0 = (int)(floorf(p.x * inv) - (float)min_b[0]);
        const int i1 = (int)(floorf(p.y * inv) - (float)min_b[1]);
        const int i2 = (int)(floorf(p.z * inv) - (float)min_b[2]);
        vidx[q] = (unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
        src[q] = (unsigned)q;
    }
    const unsigned n_vox = (unsigned)div_b[0] * (unsigned)div_b[1] * (unsigned)div_b[2];
    int bits = 1; while (bits < 32 && (n_vox - 1u) >> bits) bits++;
    // second ping-pong buffer: the upper half of the ord_a area is not available (other rings), so the mark area's neighbour — a dedicated
    // stretch of the digit table buffer — is used: B.rcnt holds [ring][table | n ints]
    int* table = B.rcnt + B.rcnt_off[ring];
    const int n_slots = (n_lf + kRotBlock - 1) / kRotBlock;
    const int tsize = 32 * n_slots * (kRotBlock / 64);
    dst = reinterpret_cast<unsigned*>(table + tsize);
    __threadfence_block();
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int shift = 0; shift < bits; shift += 5) {
        for (int k = tid; k < tsize; k += kRotBlock) table[k] = 0;
        __threadfence_block();
        __syncthreads();
        for (int sl = 0; sl < n_slots; sl++) {
            const int i = sl * kRotBlock + tid;
            const bool act = i < n_lf;
            const int d = act ? (int)((vidx[src[i]] >> shift) & 31u) : 0;
            unsigned long long m = __ballot(act);
#pragma unroll
            for (int bb = 0; bb < 5; bb++) { const unsigned long long bal = __ballot(act && ((d >> bb) & 1)); m &= ((d >> bb) & 1) ? bal : ~bal; }
            if (act && __popcll(m & ((1ull << lane) - 1ull)) == 0) table[(d * n_slots + sl) * (kRotBlock / 64) + wave] = __popcll(m);
        }
        __threadfence_block();
        __syncthreads();
        int carry = 0;                                 // exclusive scan of the table, kRotBlock entries per round
        for (int k0 = 0; k0 < tsize; k0 += kRotBlock) {
            const int k = k0 + tid;
            const int v = k < tsize ? table[k] : 0;
            int tot; const int ex = block_excl_scan_1024(v, scan, tot);
            if (k < tsize) table[k] = carry + ex;
            carry += tot;
        }
        __threadfence_block();
        __syncthreads();
        for (int sl = 0; sl < n_slots; sl++) {
            const int i = sl * kRotBlock + tid;
            const bool act = i < n_lf;
            const unsigned q = act ? src[i] : 0u;
            const int d = act ? (int)((vidx[q] >> shift) & 31u) : 0;
            unsigned long long m = __ballot(act);
#pragma unroll
            for (int bb = 0; bb < 5; bb++) { const unsigned long long bal = __ballot(act && ((d >> bb) & 1)); m &= ((d >> bb) & 1) ? bal : ~bal; }
            if (act) dst[table[(d * n_slots + sl) * (kRotBlock / 64) + wave] + __popcll(m & ((1ull << lane) - 1ull))] = q;
        }
        __threadfence_block();
        __syncthreads();
        unsigned* t2 = src; src = dst; dst = t2;
    }
    int n_out = 0;
    for (int q0 = 0; q0 < n_lf; q0 += kRotBlock) {
        const int q = q0 + tid;
        const bool head = q < n_lf && (q == 0 || vidx[src[q]] != vidx[src[q - 1]]);
        int tot; const int off = block_excl_scan_1024(head ? 1 : 0, scan, tot);
        if (head) {
            const unsigned vox = vidx[src[q]];
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f; int cnt = 0;
            for (int m = q; m < n_lf && vidx[src[m]] == vox; m++) { const float4 p = full[lf[src[m]]]; sx += p.x; sy += p.y; sz += p.z; si += p.w; cnt++; }
            const float fn = (float)cnt;
            surf_tmp[rbase + n_out + off] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
            surf_cnt_tmp[rbase + n_out + off] = cnt;
        }
        n_out += tot;
    }
    if (tid == 0) st->ring_nsurf[ring] = n_out;
}

// ordered concatenation of the per-ring lists (rings ascending, then push order inside the ring): one block per ring
__global__ __launch_bounds__(256) void k_rot_compact(RotState* st, const float4* __restrict__ full,
                                                     const int* __restrict__ ring_edge, const int* __restrict__ ring_sharp, const int* __restrict__ ring_flat,
                                                     const int* __restrict__ lessflat_tmp, const float4* __restrict__ surf_tmp, const int* __restrict__ surf_cnt_tmp,
                                                     int* __restrict__ edge_idx, float4* __restrict__ edge_pts, int* __restrict__ sharp_idx, int* __restrict__ flat_idx,
                                                     int* __restrict__ lessflat_idx, float4* __restrict__ surf, int* __restrict__ surf_cnt) {
    __shared__ int off[5], tot[5];
    const int r = blockIdx.x;
    if (threadIdx.x < 5) {
        const int* arr = threadIdx.x == 0 ? st->ring_nedge : threadIdx.x == 1 ? st->ring_nsharp : threadIdx.x == 2 ? st->ring_nflat : threadIdx.x == 3 ? st->ring_nlf : st->ring_nsurf;
        int a = 0, t = 0;
        for (int k = 0; k < kMaxRings; k++) { int c = arr[k]; if (k < r) a += c; t += c; }
        off[threadIdx.x] = a; tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (r == 0 && threadIdx.x == 0) { st->n_edge = tot[0]; st->n_sharp = tot[1]; st->n_flat = tot[2]; st->n_lessflat = tot[3]; st->n_surf = tot[4]; }
    const int rb = st->ring_base[r];
    for (int k = threadIdx.x; k < st->ring_nedge[r]; k += blockDim.x) { int g = ring_edge[r * kRingEdgeCap + k]; edge_idx[off[0] + k] = g; edge_pts[off[0] + k] = full[g]; }
    for (int k = threadIdx.x; k < st->ring_nsharp[r]; k += blockDim.x) sharp_idx[off[1] + k] = ring_sharp[r * kRingSharpCap + k];
    for (int k = threadIdx.x; k < st->ring_nflat[r]; k += blockDim.x) flat_idx[off[2] + k] = ring_flat[r * kRingFlatCap + k];
    for (int k = threadIdx.x; k < st->ring_nlf[r]; k += blockDim.x) lessflat_idx[off[3] + k] = lessflat_tmp[rb + k];
    for (int k = threadIdx.x; k < st->ring_nsurf[r]; k += blockDim.x) { surf[off[4] + k] = surf_tmp[rb + k]; surf_cnt[off[4] + k] = surf_cnt_tmp[rb + k]; }
}

}  // namespace lili

// ================================================================================================
// C ABI
// ================================================================================================
namespace lili_detail {
struct RotBuffers {
    DevBuf in, valid, scan_id, ori_raw, block_hist, state, full, full_src, curv, label, sort_ind;
    DevBuf ring_edge, ring_sharp, ring_flat, lessflat_tmp, surf_tmp, surf_cnt_tmp;
    DevBuf edge_idx, edge_pts, sharp_idx, flat_idx, lessflat_idx, surf, surf_cnt;
    DevBuf big_mark, big_vidx, big_ord_a, big_ord_b, big_rcnt;   // working set of rings beyond the LDS budget (k_rot_select_big)
    lili::RotState host{};
    int n_in = 0;
    bool have = false;
    void release() {
        for (DevBuf* b : {&in, &valid, &scan_id, &ori_raw, &block_hist, &state, &full, &full_src, &curv, &label, &sort_ind, &ring_edge, &ring_sharp, &ring_flat,
                          &lessflat_tmp, &surf_tmp, &surf_cnt_tmp, &edge_idx, &edge_pts, &sharp_idx, &flat_idx, &lessflat_idx, &surf, &surf_cnt}) b->release();
    }
};
}  // namespace lili_detail

static lili_detail::RotBuffers* rot_of(lili_ctx* ctx) {
    if (!ctx->ext_rot) { ctx->ext_rot = new lili_detail::RotBuffers(); ctx->ext_rot_free = [](void* p) { auto* r = static_cast<lili_detail::RotBuffers*>(p); r->release(); delete r; }; }
    return static_cast<lili_detail::RotBuffers*>(ctx->ext_rot);
}

static int copy_out_f4(lili_ctx* ctx, const lili_feature_out* o, const float4* d_src, size_t count) {
    if (!o || !o->data || count == 0) return LILI_OK;
    size_t k = std::min(count, o->capacity);
    if (k == 0) return LILI_OK;
    size_t stride = o->stride ? o->stride : sizeof(float4);
    ARGCHK(stride >= sizeof(float4), "feature_out: stride must be >= 16");
    hipMemcpyKind kind = o->mem == LILI_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (stride == sizeof(float4)) HIPCHK(hipMemcpyAsync(o->data, d_src, k * sizeof(float4), kind, ctx->stream));      // one DMA; the 2-D form copies row by row
    else HIPCHK(hipMemcpy2DAsync(o->data, stride, d_src, sizeof(float4), sizeof(float4), k, kind, ctx->stream));
    return LILI_OK;
}

extern "C" {

int lili_extract_rot(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params,
                     lili_feature_out* full, lili_feature_out* edge, lili_feature_out* surf) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(scan && q_imu && q_lb && params, "extract_rot: null argument");
    ARGCHK(params->n_scans == 16 || params->n_scans == 32 || params->n_scans == 64, "extract_rot: n_scans must be 16, 32 or 64");
    ARGCHK(params->ds_rate >= 1, "extract_rot: ds_rate must be >= 1");
    ARGCHK(params->ds_v > 0, "extract_rot: ds_v must be positive");
    ARGCHK(scan->n <= 400000, "extract_rot: more than 400000 points (the reference's fixed arrays, R/src/Preprocessing.cpp:9-12)");
    HIPCHK(hipSetDevice(ctx->device));
    auto* R = rot_of(ctx);
    R->have = false;
    bool full_early = false;      // the full cloud's copy to the host was started behind k_rot_scatter (see there)
    int rc = lili_ingest_cloud(ctx, scan, R->in);
    if (rc != LILI_OK) return rc;
    const int n = (int)scan->n;
    R->n_in = n;
    HIPCHK(R->state.ensure(sizeof(RotState)));
    RotState* st = R->state.as<RotState>();
    hipLaunchKernelGGL(k_rot_init, dim3(1), dim3(64), 0, ctx->stream, st);
    if (n > 0) {
        const size_t cap = (size_t)n;
        const int nb = nblocks(n, kRotBlock);
        HIPCHK(R->valid.ensure(cap)); HIPCHK(R->scan_id.ensure(cap)); HIPCHK(R->ori_raw.ensure(cap * 4));
        HIPCHK(R->block_hist.ensure((size_t)nb * kMaxRings * 4));
        HIPCHK(R->full.ensure(cap * 16)); HIPCHK(R->full_src.ensure(cap * 4)); HIPCHK(R->curv.ensure(cap * 4)); HIPCHK(R->label.ensure(cap * 4)); HIPCHK(R->sort_ind.ensure(cap * 4));
        HIPCHK(R->ring_edge.ensure(kMaxRings * kRingEdgeCap * 4)); HIPCHK(R->ring_sharp.ensure(kMaxRings * kRingSharpCap * 4)); HIPCHK(R->ring_flat.ensure(kMaxRings * kRingFlatCap * 4));
        HIPCHK(R->lessflat_tmp.ensure(cap * 4)); HIPCHK(R->surf_tmp.ensure(cap * 16)); HIPCHK(R->surf_cnt_tmp.ensure(cap * 4));
        HIPCHK(R->edge_idx.ensure(kMaxRings * kRingEdgeCap * 4)); HIPCHK(R->edge_pts.ensure(kMaxRings * kRingEdgeCap * 16));
        HIPCHK(R->sharp_idx.ensure(kMaxRings * kRingSharpCap * 4)); HIPCHK(R->flat_idx.ensure(kMaxRings * kRingFlatCap * 4));
        HIPCHK(R->lessflat_idx.ensure(cap * 4)); HIPCHK(R->surf.ensure(cap * 16)); HIPCHK(R->surf_cnt.ensure(cap * 4));
        RotDev P{};
        P.n_scans = params->n_scans; P.ds_rate = params->ds_rate; P.ds_v = params->ds_v; P.near_thres = params->near_range;
        P.atan_mode = ctx->rot_atan;
        for (int i = 0; i < 4; i++) { P.q_imu[i] = q_imu[i]; P.q_lb[i] = q_lb[i]; }
        const float4* in = R->in.as<float4>();
        hipLaunchKernelGGL(k_rot_valid, dim3(std::min(nblocks(n, 256), 128)), dim3(256), 0, ctx->stream, in, n, P.near_thres, R->valid.as<unsigned char>(), st);
        hipLaunchKernelGGL(k_rot_classify, dim3(nb), dim3(kRotBlock), 0, ctx->stream, in, n, R->valid.as<unsigned char>(), P, st,
                           R->scan_id.as<signed char>(), R->ori_raw.as<float>(), R->block_hist.as<int>());
        hipLaunchKernelGGL(k_rot_ring_scan, dim3(1), dim3(kRotBlock), 0, ctx->stream, R->block_hist.as<int>(), nb, P, st);
        hipLaunchKernelGGL(k_rot_scatter, dim3(nb), dim3(kRotBlock), 0, ctx->stream, in, n, R->scan_id.as<signed char>(), R->ori_raw.as<float>(), P, st,
                           R->block_hist.as<int>(), R->full.as<float4>(), R->full_src.as<int>());
        // the deskewed cloud is final here: its copy to the host (3.2 MB for a 200 k-point scan, ~60 us) runs on a side stream under the feature
        // selection instead of behind it.  All n entries travel (the count is known only at the end); entries behind `count` are unspecified.
        if (full && full->data && full->mem == LILI_MEM_HOST && (full->stride == 0 || full->stride == sizeof(float4)) && full->capacity > 0) {
            if (!ctx->fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
            if (!ctx->side[1]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[1], hipStreamNonBlocking));
            if (!ctx->join_ev[1]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[1], hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->side[1], ctx->fork_ev, 0));
            HIPCHK(hipMemcpyAsync(full->data, R->full.as<float4>(), std::min((size_t)n, full->capacity) * sizeof(float4), hipMemcpyDeviceToHost, ctx->side[1]));
            HIPCHK(hipEventRecord(ctx->join_ev[1], ctx->side[1]));
            full_early = true;
        }
        hipLaunchKernelGGL(k_rot_curvature, dim3(nblocks(n, 256)), dim3(256), 0, ctx->stream, R->full.as<float4>(), st, R->curv.as<float>());
        hipLaunchKernelGGL(k_rot_rank, dim3(kMaxRings, 6, 12), dim3(256), 0, ctx->stream, R->curv.as<float>(), P, st, R->sort_ind.as<int>());
        hipLaunchKernelGGL(k_rot_select, dim3(kMaxRings), dim3(kRotBlock), sizeof(RingLds), ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                           R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                           R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>());
        hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                           R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                           R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>());
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(&R->host, st, sizeof(RotState), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (R->host.fallback_rings > 0) {      // rings beyond the LDS working set: second pass with global-memory arrays, then the concatenation again
        const size_t cap = (size_t)std::max(n, 1);
        RotBigScratch B{};
        size_t off = 0;
        for (int r = 0; r < kMaxRings; r++) {
            B.rcnt_off[r] = (int)off;
            const int rc_ = R->host.ring_count[r];
            if (rc_ > kRingLdsCap) off += (size_t)32 * (size_t)((rc_ + kRotBlock - 1) / kRotBlock) * (kRotBlock / 64) + (size_t)rc_ + 64;
        }
        HIPCHK(R->big_mark.ensure(cap + 64)); HIPCHK(R->big_vidx.ensure(cap * 4)); HIPCHK(R->big_ord_a.ensure(cap * 4)); HIPCHK(R->big_ord_b.ensure(cap * 4));
        HIPCHK(R->big_rcnt.ensure(std::max<size_t>(off, 1) * 4));
        B.mark = R->big_mark.as<signed char>(); B.vidx = R->big_vidx.as<unsigned>(); B.ord_a = R->big_ord_a.as<unsigned>(); B.ord_b = R->big_ord_b.as<unsigned>();
        B.rcnt = R->big_rcnt.as<int>();
        RotDev P{};
        P.n_scans = params->n_scans; P.ds_rate = params->ds_rate; P.ds_v = params->ds_v; P.near_thres = params->near_range; P.atan_mode = ctx->rot_atan;
        for (int i = 0; i < 4; i++) { P.q_imu[i] = q_imu[i]; P.q_lb[i] = q_lb[i]; }
        hipLaunchKernelGGL(k_rot_select_big, dim3(kMaxRings), dim3(kRotBlock), 0, ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                           R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                           R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), B);
        hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                           R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                           R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&R->host, st, sizeof(RotState), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    R->have = true;
    if (full) {
        full->count = (size_t)R->host.n_full;
        if (full_early) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->join_ev[1], 0));
        else { rc = copy_out_f4(ctx, full, R->full.as<float4>(), full->count); if (rc) return rc; }
    }
    if (edge) { edge->count = (size_t)R->host.n_edge; rc = copy_out_f4(ctx, edge, R->edge_pts.as<float4>(), edge->count); if (rc) return rc; }
    if (surf) { surf->count = (size_t)R->host.n_surf; rc = copy_out_f4(ctx, surf, R->surf.as<float4>(), surf->count); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}

// Intermediate products of the last lili_extract_rot (parity tests / debugging).  Any pointer may be NULL.
int lili_extract_rot_debug(lili_ctx* ctx, int32_t counts[8], int32_t* ring_start, int32_t* ring_end, int32_t* full_src, float* curvature, int32_t* label,
                           int32_t* edge_idx, int32_t* sharp_idx, int32_t* flat_idx, int32_t* lessflat_idx, int32_t* surf_cnt) {
    if (!ctx) return LILI_E_ARG;
    auto* R = rot_of(ctx);
    if (!R->have) return ctx->fail(LILI_E_STATE, "extract_rot_debug: run lili_extract_rot first");
    HIPCHK(hipSetDevice(ctx->device));
    const RotState& h = R->host;
    if (getenv("LILI_ROT_PHASES")) { fprintf(stderr, "rot_select phases (100 MHz ticks):"); for (int k = 1; k < 7; k++) fprintf(stderr, " %lld", h.tphase[k] - h.tphase[k-1]); fprintf(stderr, "\n"); }
    if (counts) { counts[0] = h.n_full; counts[1] = h.n_edge; counts[2] = h.n_sharp; counts[3] = h.n_flat; counts[4] = h.n_lessflat; counts[5] = h.n_surf; counts[6] = h.half_idx; counts[7] = h.first_valid; }
    if (ring_start) std::memcpy(ring_start, h.ring_start, sizeof(int) * kMaxRings);
    if (ring_end) std::memcpy(ring_end, h.ring_end, sizeof(int) * kMaxRings);
    auto dl = [&](void* dst, const DevBuf& src, size_t bytes) -> hipError_t { return (dst && bytes) ? hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess; };
    HIPCHK(dl(full_src, R->full_src, (size_t)h.n_full * 4)); HIPCHK(dl(curvature, R->curv, (size_t)h.n_full * 4)); HIPCHK(dl(label, R->label, (size_t)h.n_full * 4));
    HIPCHK(dl(edge_idx, R->edge_idx, (size_t)h.n_edge * 4)); HIPCHK(dl(sharp_idx, R->sharp_idx, (size_t)h.n_sharp * 4)); HIPCHK(dl(flat_idx, R->flat_idx, (size_t)h.n_flat * 4));
    HIPCHK(dl(lessflat_idx, R->lessflat_idx, (size_t)h.n_lessflat * 4)); HIPCHK(dl(surf_cnt, R->surf_cnt, (size_t)h.n_surf * 4));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}


// Device views of the last lili_extract_rot results (float4 x,y,z,intensity; valid until the next extract on this
// context): feed them back as lili_cloud{ptr, n, 16, 12, LILI_MEM_DEVICE} into lili_voxel_filter /
// lili_s2m_set_queries / lili_localmap_push so that a scan never leaves HBM between extraction and matching.
int lili_extract_rot_device(lili_ctx* ctx, lili_cloud* full, lili_cloud* edge, lili_cloud* surf) {
    if (!ctx) return LILI_E_ARG;
    auto* R = rot_of(ctx);
    if (!R->have) return ctx->fail(LILI_E_STATE, "extract_rot_device: run lili_extract_rot first");
    if (full) *full = lili_cloud{R->full.p, (size_t)R->host.n_full, 16, 12, LILI_MEM_DEVICE};
    if (edge) *edge = lili_cloud{R->edge_pts.p, (size_t)R->host.n_edge, 16, 12, LILI_MEM_DEVICE};
    if (surf) *surf = lili_cloud{R->surf.p, (size_t)R->host.n_surf, 16, 12, LILI_MEM_DEVICE};
    return LILI_OK;
}

}  // extern "C"
