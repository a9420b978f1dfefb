// LOAM-style feature extractor of LiLi-OM-ROT on gfx950 — replaces R/src/Preprocessing.cpp:277-509
// (R/ = LiLi-OM-ROT/) behind lili_extract_rot():
//   k_rot_classify   NaN / near-range filter, first & last surviving point (every workgroup for itself), elevation -> ring id, azimuth, `halfPassed` latch
//                    index, per-workgroup ring histogram                                                                     (R:280-294, 308-365)
//   k_rot_scatter    ring offsets from the histograms (every workgroup for itself: stable per-ring compaction = laserCloudScans[] + concatenation), relTime /
//                    intensity, IMU deskew (slerp, f64), scatter into the ring-concatenated cloud, packed voxel keys                  (R:367-382, 153-177)
//   k_rot_segments   one workgroup per work item.  Segment (ring, j): 11-tap curvature, binned rank sort by (curvature, index), greedy sharp / less-sharp / flat
//                    picks with +-5 neighbour suppression on one wave (break bits + mark ranges in registers); a segment whose pick may lie under marks of the
//                    segment before it waits for them and runs again.  Ordering (ring): the order pcl::VoxelGrid(ds_v) needs, on runs of candidates
//                                                                                                                             (R:385-394, 401-492, 502-508)
//   k_rot_ring       one workgroup per ring: joins the six segments, pick lists, less-flat list, in-order f32 VoxelGrid centroids (four lanes per voxel) — written
//                    where the SCAN's lists want them behind a look-back over the lower rings (no concatenation launch), optionally into a matcher slot as well
//                                                                                                                             (R:401-508)
//   (k_rot_compact: ordered concatenation of the per-ring lists — second passes and the fallback of a look-back that gave up; k_rot_voxel_order, k_rot_rank,
//   k_rot_select_big: scans beyond the packed voxel keys / rings beyond the LDS working set)
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
    int vox_overflow;     // a point's voxel coordinates did not fit the packed 32-bit key (k_rot_scatter): the radix ordering pass takes over
    int redo_segments;    // segments whose concurrent greedy run had to be repeated with the previous segment's marks (diagnostics)
    int fold_failed;      // a ring of k_rot_ring gave up waiting for the counts of the rings below it: the host repeats the concatenation (k_rot_compact)
    long long tphase[8];  // profiling: per-phase clock ticks of ring 0's workgroup (wall_clock64)
    int ring_ticks[kMaxRings];   // profiling: ticks k_rot_ring spent on each ring
    int order_ticks[kMaxRings];  // profiling: ticks of the ring's voxel-ordering item; seg_ticks: its slowest segment item
    int seg_ticks[kMaxRings];
    long long tjoin[4], tseg[4];   // profiling: inside the join of the probe ring (working set in LDS | border check | pick lists)
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

// Eigen 3.3 slerp of Identity towards b, in two parts: what depends on b alone (the angle and its sine: one acos and one sin, the same for every
// point of a scan — computed once per workgroup) and what depends on t.
struct SlerpConst { double theta, sin_theta; int linear; };
__device__ __forceinline__ SlerpConst qslerp_prepare(dq b) {
    const double one = 1.0 - 2.220446049250313e-16;
    const double absD = fabs(b.w);
    SlerpConst c{0.0, 1.0, 1};
    if (!(absD >= one)) { c.theta = acos(absD); c.sin_theta = sin(c.theta); c.linear = 0; }
    return c;
}
__device__ __forceinline__ dq qslerp_identity(double t, dq b, SlerpConst c) {
    double s0, s1;
    if (c.linear) { s0 = 1.0 - t; s1 = t; }
    else {
        s0 = sin((1.0 - t) * c.theta) / c.sin_theta;
        s1 = sin(t * c.theta) / c.sin_theta;
    }
    if (b.w < 0) s1 = -s1;
    return dq{s0 + s1 * b.w, s1 * b.x, s1 * b.y, s1 * b.z};
}
__device__ __forceinline__ float range2(float4 p) { return p.x * p.x + p.y * p.y + p.z * p.z; }
__device__ __forceinline__ bool rot_point_ok(float4 p, float thres) {      // R:131-134: finite and not inside the near range
    return isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && !(p.x * p.x + p.y * p.y + p.z * p.z < thres * thres);
}

// First and last surviving point of the scan (R:280-294), found by EVERY workgroup on its own instead of by a reduction launch: the whole block
// looks at 1024 points from the front and 1024 from the back per trip — the first trip ends the search unless the scan starts / ends with
// more than a thousand dropped points.  All threads return the same pair (0x7fffffff / -1 when nothing survives).
__device__ void rot_first_last_valid(const float4* __restrict__ in, int n, float thres, int& first, int& last) {
    __shared__ int s_first[kRotBlock / 64], s_last[kRotBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    first = 0x7fffffff; last = -1;
    for (int b = 0; b < n && (first == 0x7fffffff || last < 0); b += kRotBlock) {
        const int i = b + (int)threadIdx.x, k = n - 1 - b - (int)threadIdx.x;
        const bool oi = first == 0x7fffffff && i < n && rot_point_ok(in[i], thres);
        const bool ok = last < 0 && k >= 0 && rot_point_ok(in[k], thres);
        const unsigned long long mi = __ballot(oi), mk = __ballot(ok);
        if (lane == 0) {
            s_first[wave] = mi ? b + wave * 64 + (__ffsll((long long)mi) - 1) : 0x7fffffff;
            s_last[wave] = mk ? n - 1 - b - wave * 64 - (__ffsll((long long)mk) - 1) : -1;
        }
        __syncthreads();
        int f = 0x7fffffff, l = -1;
        for (int w = 0; w < kRotBlock / 64; w++) { f = min(f, s_first[w]); l = max(l, s_last[w]); }
        if (first == 0x7fffffff) first = f;
        if (last < 0) last = l;
        __syncthreads();
    }
}

__device__ __forceinline__ void start_end_ori(float4 a /*first surviving point*/, float4 b /*last*/, int am, float& startOri, float& endOri) {
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

// Launch 1 of 4.  Per point: NaN / near-range filter, ring id, raw azimuth; per workgroup of 1024 points: ring histogram and the first point that
// would latch `halfPassed` (R:351-357).  Nothing is read that another workgroup writes (start azimuth: rot_first_last_valid), nothing needs
// zeroing first.
__global__ __launch_bounds__(kRotBlock) void k_rot_classify(const float4* __restrict__ in, int n, RotDev P, RotState* st, signed char* __restrict__ scan_id,
                                                            float* __restrict__ ori_raw, int* __restrict__ block_hist /*[nb][64]*/, int* __restrict__ block_half /*[nb]*/) {
    __shared__ int hist[kMaxRings];
    __shared__ int half_min;
    if (threadIdx.x < kMaxRings) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) half_min = 0x7fffffff;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 p = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);      // (requested before the search below waits for its own loads)
    int first, last;
    rot_first_last_valid(in, n, P.near_thres, first, last);          // (its barriers also publish the two initialisations above)
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->first_valid = first; st->last_valid = last; st->vox_overflow = 0; }
    const float4 pa = last >= 0 ? in[first] : p, pb = last >= 0 ? in[last] : p;
    int id = -1;
    if (last >= 0 && i < n) {
        if (rot_point_ok(p, P.near_thres)) {
            id = ring_of(p, P.n_scans, P.atan_mode);
            if (id >= 0) {
                float startOri, endOri;
                start_end_ori(pa, pb, P.atan_mode, startOri, endOri);
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
    }
    if (i < n) scan_id[i] = (signed char)id;
    __syncthreads();
    if (threadIdx.x < kMaxRings) block_hist[blockIdx.x * kMaxRings + threadIdx.x] = hist[threadIdx.x];
    if (threadIdx.x == 0) block_half[blockIdx.x] = half_min;
}

// Launch 2 of 4.  Every workgroup sums the ring histograms itself (all of them: ring sizes; those of the workgroups before it: its own offsets) —
// 50 KB of L2 reads instead of a one-workgroup scan launch in between —, then: relTime / intensity, IMU deskew (slerp, f64), stable scatter into
// the ring-concatenated cloud (R:367-382, 153-177).  Workgroup 0 also writes the ring table of the scan state.
__global__ __launch_bounds__(kRotBlock) void k_rot_scatter(const float4* __restrict__ in, int n, int nb, const signed char* __restrict__ scan_id,
                                                           const float* __restrict__ ori_raw, RotDev P, RotState* st, const int* __restrict__ block_hist,
                                                           const int* __restrict__ block_half, float4* __restrict__ full, int* __restrict__ full_src,
                                                           unsigned* __restrict__ vkey, int* __restrict__ ring_ncand) {
    __shared__ int wave_hist[kRotBlock / 64][kMaxRings];
    __shared__ int part_pre[kRotBlock / 64][kMaxRings], part_tot[kRotBlock / 64][kMaxRings];
    __shared__ int ring_base[kMaxRings], ring_cnt[kMaxRings], my_base[kMaxRings];
    __shared__ int half_w[kRotBlock / 64];
    __shared__ SlerpConst slerp_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this thread's point and what k_rot_classify left about it: requested here, used behind the histogram sums (every dependent round trip of this launch is ~1-2 us)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int id = i < n ? (int)scan_id[i] : -1;
    const float ori_in = i < n ? ori_raw[i] : 0.f;                                  // (written for surviving points only)
    const float4 p_in = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int first_valid = st->first_valid, last_valid = st->last_valid;       // k_rot_classify, workgroup 0
    if (threadIdx.x == kRotBlock - 1) slerp_s = qslerp_prepare(dq{P.q_imu[0], P.q_imu[1], P.q_imu[2], P.q_imu[3]});
    for (int k = threadIdx.x; k < (kRotBlock / 64) * kMaxRings; k += blockDim.x) (&wave_hist[0][0])[k] = 0;
    {
        int pre = 0, tot = 0;
        constexpr int kW = kRotBlock / 64;
        for (int b0 = wave; b0 < nb; b0 += 16 * kW) {           // sixteen loads in flight per trip: ONE trip for scans of up to 262 144 points (a plain loop waits for every load before it issues the next)
            int c[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int b = b0 + u * kW; c[u] = b < nb ? block_hist[b * kMaxRings + lane] : 0; }
#pragma unroll
            for (int u = 0; u < 16; u++) { tot += c[u]; if (b0 + u * kW < (int)blockIdx.x) pre += c[u]; }
        }
        part_pre[wave][lane] = pre; part_tot[wave][lane] = tot;
        int hm = 0x7fffffff;
        for (int b = threadIdx.x; b < nb; b += kRotBlock) hm = min(hm, block_half[b]);
        for (int o = 32; o > 0; o >>= 1) hm = min(hm, __shfl_xor(hm, o));
        if (lane == 0) half_w[wave] = hm;
    }
    __syncthreads();
    if (wave == 0) {
        int pre = 0, tot = 0;
        for (int w = 0; w < kRotBlock / 64; w++) { pre += part_pre[w][lane]; tot += part_tot[w][lane]; }
        int inc = tot;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        ring_base[lane] = inc - tot; ring_cnt[lane] = tot; my_base[lane] = pre;
    }
    __syncthreads();
    int half_idx = 0x7fffffff;
    for (int w = 0; w < kRotBlock / 64; w++) half_idx = min(half_idx, half_w[w]);
    const float4 pa = last_valid >= 0 ? in[first_valid] : p_in, pb = last_valid >= 0 ? in[last_valid] : p_in;
    const SlerpConst slerp_c = slerp_s;
    if (blockIdx.x == 0) {
        if (threadIdx.x < kMaxRings) {
            const int r = threadIdx.x, bs = ring_base[r], c = ring_cnt[r];
            st->ring_count[r] = c; st->ring_base[r] = bs;
            st->ring_start[r] = r < P.n_scans ? bs + 5 : 0; st->ring_end[r] = r < P.n_scans ? bs + c - 6 : 0;       // R:379-381
            st->ring_nedge[r] = st->ring_nsharp[r] = st->ring_nflat[r] = st->ring_nlf[r] = st->ring_nsurf[r] = 0;
            ring_ncand[r] = 0; st->seg_ticks[r] = 0; st->order_ticks[r] = 0; st->ring_ticks[r] = 0;
        }
        if (threadIdx.x == 0) {
            st->n_full = ring_base[kMaxRings - 1] + ring_cnt[kMaxRings - 1]; st->half_idx = half_idx;
            st->n_edge = st->n_sharp = st->n_flat = st->n_lessflat = st->n_surf = 0; st->fallback_rings = 0; st->redo_segments = 0; st->fold_failed = 0;
        }
    }
    // stable rank of the point among the points of its ring inside this wave
    // (six ballots — the lanes that agree with this one on every bit of the ring id — instead of one trip per distinct ring: the 64 lanes of a wave
    // hold a spinning sensor's firing order, i.e. up to 64 different rings)
    int rank = 0;
    {
        unsigned long long m = __ballot(id >= 0);
#pragma unroll
        for (int b = 0; b < 6; b++) { const unsigned long long bal = __ballot((id >> b) & 1); m &= ((id >> b) & 1) ? bal : ~bal; }
        if (id >= 0) {
            rank = __popcll(m & ((1ull << lane) - 1ull));
            if (rank == 0) wave_hist[wave][id] = __popcll(m);
        }
    }
    __syncthreads();
    if (id < 0) return;
    // the points of this ring in the waves in front of this one: up to fifteen independent LDS reads per thread (a serial prefix by 64 threads and a second barrier before)
    int wave_base = 0;
#pragma unroll
    for (int w = 0; w < kRotBlock / 64; w++) wave_base += w < wave ? wave_hist[w][id] : 0;
    const int pos = ring_base[id] + my_base[id] + wave_base + rank;
    float startOri, endOri;
    start_end_ori(pa, pb, P.atan_mode, startOri, endOri);
    float ori = ori_in;
    if (i <= half_idx) {   // halfPassed was still false when the reference reached this point (R:350-358)
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
    const float4 p = p_in;
    int line = (int)intensity;
    double dt_i = (double)(intensity - (float)line);
    double ratio = dt_i / 0.1;
    if (ratio >= 1.0) ratio = 1.0;
    dq qimu{P.q_imu[0], P.q_imu[1], P.q_imu[2], P.q_imu[3]}, qlb{P.q_lb[0], P.q_lb[1], P.q_lb[2], P.q_lb[3]};
    dq qs = qslerp_identity(ratio, qimu, slerp_c);
    qs = qmul(qmul(qlb, qs), qinv(qlb));
    d3 r = qrot(qs, d3{(double)p.x, (double)p.y, (double)p.z});
    const float4 o = make_float4((float)r.x, (float)r.y, (float)r.z, intensity);
    full[pos] = o;
    full_src[pos] = i;
    // Key of the point for the VoxelGrid(ds_v) ordering of its ring (R:502-508): PCL orders the voxels by x + y * dx + z * dx * dy over the
    // bounding box = lexicographically by (floor(z / v), floor(y / v), floor(x / v)), whatever the box — packed here as 9 | 11 | 11 offset-binary
    // bits (+-153 m in z, +-614 m in x / y at v = 0.6).  0xffffffff: not a candidate of the less-flat list (outside [scanStartInd,
    // scanEndInd) or nearer than 0.5 m, R:494-499; the picked edge points leave the list later, in k_rot_ring).
    const int k = pos - ring_base[id];
    unsigned key = 0xffffffffu;
    if (k >= 5 && k <= ring_cnt[id] - 7 && !((double)range2(o) < 0.25)) {
        const float inv = 1.0f / P.ds_v;
        const float f0 = floorf(o.x * inv), f1 = floorf(o.y * inv), f2 = floorf(o.z * inv);
        if (f0 >= -1024.f && f0 < 1024.f && f1 >= -1024.f && f1 < 1024.f && f2 >= -256.f && f2 < 256.f)
            key = ((unsigned)((int)f2 + 256) << 22) | ((unsigned)((int)f1 + 1024) << 11) | (unsigned)((int)f0 + 1024);
        else { key = 0u; atomicOr(&st->vox_overflow, 1); }
    }
    vkey[pos] = key;
}

// 11-tap curvature (R:385-394): strictly left-to-right f32 sums.  `P` is indexed so that P[k] is the point itself.
__device__ __forceinline__ float curvature11(const float4* __restrict__ P, int k) {
    float dX = P[k - 5].x + P[k - 4].x + P[k - 3].x + P[k - 2].x + P[k - 1].x - 10 * P[k].x + P[k + 1].x + P[k + 2].x + P[k + 3].x + P[k + 4].x + P[k + 5].x;
    float dY = P[k - 5].y + P[k - 4].y + P[k - 3].y + P[k - 2].y + P[k - 1].y - 10 * P[k].y + P[k + 1].y + P[k + 2].y + P[k + 3].y + P[k + 4].y + P[k + 5].y;
    float dZ = P[k - 5].z + P[k - 4].z + P[k - 3].z + P[k - 2].z + P[k - 1].z - 10 * P[k].z + P[k + 1].z + P[k + 2].z + P[k + 3].z + P[k + 4].z + P[k + 5].z;
    return dX * dX + dY * dY + dZ * dZ;
}

// ------------------------------------------------------------------------------------------------
// per-ring selection
// ------------------------------------------------------------------------------------------------
constexpr int kRingLdsCap = 4096;   // ring points held in LDS (SYN: 3125, HDL-64E: ~2100, VLP-16: ~1800)
constexpr int kSegCap = 720;                        // (multiple of 16)  window of ONE segment: its points (<= cap / 6 + 1), the ring's first / last five with segments 0 / 5, +-5 margins

__device__ __forceinline__ float gap2(const float4* __restrict__ P, int a, int b) {   // R:435-438
    float dX = P[a].x - P[b].x, dY = P[a].y - P[b].y, dZ = P[a].z - P[b].z;
    return dX * dX + dY * dY + dZ * dZ;
}
__device__ __forceinline__ float range2(const float4* __restrict__ P, int k) { return P[k].x * P[k].x + P[k].y * P[k].y + P[k].z * P[k].z; }

constexpr int kSegEdge = 10, kSegFlat = 4;
constexpr int kGapWords = (kSegCap + 63) / 64;      // 12
constexpr int kProbeRing = 12;                     // LILI_ROT_PHASES: the ring whose k_rot_ring phases are recorded
constexpr int kStage3Blocks = 480;                 // workgroups of k_rot_segments (two per CU are resident)
static_assert(kStage3Blocks >= 7 * kMaxRings, "k_rot_segments: every work item its own workgroup, item i in workgroup i (a segment waits for the workgroup before it)");

// One segment's working set in LDS, indexed in WINDOW coordinates (window = the segment, +-5 points, and the ring's first / last five with
// its first / last segment; for a nearly empty ring the window is the whole ring).
struct SegLds {
    float4 pts[kSegCap];
    float curv[kSegCap];
    alignas(16) float key[kSegCap + 16];   // the segment's curvatures, 16-byte aligned and padded with +inf (rank sort)
    int sort_ind[kSegCap];
    int hrank[kSegCap];                    // rank sort: element numbers in bin order
    int cnt[kRotBlock + 1];                // rank sort: elements per bin, then the bins' first positions
    int scan[kRotBlock / 64 + 1];
    signed char mark[kSegCap + 16];        // cloudNeighborPicked of this segment: [sp - 5, ep + 5]
    signed char label[kSegCap];
    int edge[kSegEdge], flat[kSegFlat], ne, nf;      // picks in push order
    unsigned long long gapw[kGapWords];    // bit k & 63 of word k >> 6: gap2(k + 1, k) > 0.05 — the break test of every suppression (R:441-452), once per window
};
// What k_rot_segments hands to k_rot_ring per (ring, segment): picks as RING-LOCAL indices, counts, and the marks spilled into the next segment.
struct SegOut { int edge[kSegEdge]; int flat[kSegFlat]; int ne, nf, spill, pad; };

// The greedy picks of ONE segment (R:413-492), run by one lane.  `M` = the segment's cloudNeighborPicked, M[k] for k in [sp - 5, ep + 5]
// (cleared by the caller; `spill` = bit l set <=> element sp + l was already marked by the previous segment's picks).  Writes the labels
// of its picks, the pick lists in push order and its marks.
__device__ void greedy_segment(SegLds& L, int sp, int ep, unsigned spill) {
    const float4* Pp = L.pts;
    signed char* M = L.mark;
    for (int l = 0; l < 5; l++) if ((spill >> l) & 1u) M[sp + l] = 1;
    int ne = 0, nf = 0;
    int largest = 0;
    for (int k = ep; k >= sp; k--) {                                    // R:413-453
        int ind = L.sort_ind[k];
        if (!((double)L.curv[ind] > 2.0)) break;                        // sorted: nothing further can qualify
        if (M[ind] == 0) {
            largest++;
            if (largest <= 2) { L.label[ind] = 2; L.edge[ne++] = ind; }
            else if (largest <= 10) { L.label[ind] = 1; L.edge[ne++] = ind; }
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
            L.label[ind] = -1; L.flat[nf++] = ind;
            smallest++;
            if (smallest >= 4) break;                                   // before the suppression (R:468-470)
            M[ind] = 1;
            for (int l = 1; l <= 5; l++) { if ((double)gap2(Pp, ind + l, ind + l - 1) > 0.05) break; M[ind + l] = 1; }
            for (int l = -1; l >= -5; l--) { if ((double)gap2(Pp, ind + l, ind + l + 1) > 0.05) break; M[ind + l] = 1; }
        }
    }
    L.ne = ne; L.nf = nf;
}

#define LILI_ROT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// The same picks by a whole WAVE (all 64 lanes call it with the same arguments), without a memory round trip per pick (round 6; the version of rounds 3-5 kept
// cloudNeighborPicked as bytes in LDS and spent ~0.4 us per pick in four dependent LDS round trips: eligibility, broadcast, two points per gap test, marks):
//   * 64 candidates of the sorted order are examined at a time, the first eligible one = first set bit of a ballot;
//   * the break tests of a suppression (R:441-452) are bits of L.gapw (evaluated for the whole window before the call), a word per lane, fetched by v_readlane: the
//     marks of a pick at `ind` are the range [ind - nb, ind + nf], nb / nf = the zero bits below / from `ind` (at most five) — scalar arithmetic;
//   * a lane learns that ITS candidate got marked by comparing it with the range of every pick (no mark array); the ranges stay in registers (lane r: range r) for the
//     candidates of a later batch and for the marks spilled behind `ep`.
// Picks, labels and lists are exactly those of greedy_segment: a pick only ever ADDS marks and the candidates are visited in the same order.  Returns the marks
// spilled into the next segment (bit l: element ep + 1 + l).  Caller: LILI_ROT_WAVE_SYNC() before reading the lists.
__device__ unsigned greedy_segment_wave(SegLds& L, int sp, int ep, unsigned spill = 0u) {
    const int lane = threadIdx.x & 63;
    const unsigned long long Gw = lane < kGapWords ? L.gapw[lane] : 0ull;
    const int Glo = (int)(unsigned)Gw, Ghi = (int)(unsigned)(Gw >> 32);
    int ra = 0, rb = -1, nr = 0;                     // lane r: marks of pick r (nr picks so far, uniform)
    int epick = 0, fpick = 0;                        // lane q: q-th edge / flat pick
    const auto reach = [&](int ind, int& a, int& b) {       // `ind` uniform
        const int p0 = ind - 5, w = __builtin_amdgcn_readfirstlane(p0 >> 6), sh = p0 & 63;     // (p0 >= 0: the window starts five points before the segment)
        const unsigned long long g0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(Ghi, w) << 32) | (unsigned)__builtin_amdgcn_readlane(Glo, w);
        const unsigned long long g1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(Ghi, w + 1) << 32) | (unsigned)__builtin_amdgcn_readlane(Glo, w + 1);
        const unsigned bits = (unsigned)((g0 >> sh) | (sh ? g1 << (64 - sh) : 0ull)) & 1023u;      // bit i: gap2(p0 + i + 1, p0 + i) > 0.05
        const unsigned fw = bits >> 5, bw = bits & 31u;
        const int nf = fw ? __ffs((int)fw) - 1 : 5;            // l = 1 .. 5: gap(ind + l, ind + l - 1) = bit 4 + l
        const int nb = bw ? __clz((int)bw) - 27 : 5;           // l = -1 .. -5: gap(ind + l, ind + l + 1) = bit 5 + l; highest set bit h: 4 - h marks
        a = ind - nb; b = ind + nf;
    };
    const auto marked = [&](int cand) -> bool {                // is `cand` (per lane) under a mark of the picks so far / of the previous segment's spill?
        bool d = cand >= sp && cand < sp + 5 && ((spill >> (cand - sp)) & 1u);
        for (int r = 0; r < nr; r++) { const int a = __builtin_amdgcn_readlane(ra, r), b = __builtin_amdgcn_readlane(rb, r); d = d || (cand >= a && cand <= b); }
        return d;
    };
    int ne = 0, nf_ = 0, largest = 0;
    bool done = false;
    for (int k0 = ep; k0 >= sp && !done; k0 -= 64) {                        // R:413-453, 64 candidates at a time
        const int k = k0 - lane;
        const bool in = k >= sp;
        const int ind = in ? L.sort_ind[k] : sp;
        const bool big = in && (double)L.curv[ind] > 2.0;
        bool dead = marked(ind);
        const unsigned long long stop = __ballot(!big);                     // first lane that ends the loop (curvature too small, or the segment's end)
        const unsigned long long live = stop ? ((stop & (0ull - stop)) - 1ull) : ~0ull;       // lanes before it
        unsigned long long todo = live;
        while (todo) {
            const unsigned long long el = __ballot(big && !dead) & todo;
            if (!el) break;
            const int l = __ffsll((long long)el) - 1;
            const int pick = __builtin_amdgcn_readlane(ind, l);
            largest++;
            if (largest > 10) { done = true; break; }
            if (lane == ne) epick = pick;
            ne++;
            int a, b; reach(pick, a, b);
            if (lane == nr) { ra = a; rb = b; }
            nr++;
            dead = dead || (ind >= a && ind <= b);
            todo &= ~((2ull << l) - 1ull);                                  // lanes behind the pick
        }
        if (stop) done = true;
    }
    int smallest = 0;
    done = false;
    for (int k0 = sp; k0 <= ep && !done; k0 += 64) {                        // R:456-492
        const int k = k0 + lane;
        const bool in = k <= ep;
        const int ind = in ? L.sort_ind[k] : sp;
        const bool small = in && (double)L.curv[ind] < 0.1;
        const bool far = in && !((double)range2(L.pts, ind) < 0.25);
        bool dead = marked(ind);
        const unsigned long long stop = __ballot(!small);
        const unsigned long long live = stop ? ((stop & (0ull - stop)) - 1ull) : ~0ull;
        unsigned long long todo = live;
        while (todo) {
            const unsigned long long el = __ballot(small && far && !dead) & todo;
            if (!el) break;
            const int l = __ffsll((long long)el) - 1;
            const int pick = __builtin_amdgcn_readlane(ind, l);
            if (lane == nf_) fpick = pick;
            nf_++;
            smallest++;
            if (smallest >= 4) { done = true; break; }                      // before the suppression (R:468-470)
            int a, b; reach(pick, a, b);
            if (lane == nr) { ra = a; rb = b; }
            nr++;
            dead = dead || (ind >= a && ind <= b);
            todo &= ~((2ull << l) - 1ull);
        }
        if (stop) done = true;
    }
    if (lane < ne) { L.edge[lane] = epick; L.label[epick] = lane < 2 ? 2 : 1; }
    if (lane < nf_) { L.flat[lane] = fpick; L.label[fpick] = -1; }
    if (lane == 0) { L.ne = ne; L.nf = nf_; }
    // marks behind the segment's end
    bool sp_l = false;
    const int pos = ep + 1 + lane;
    for (int r = 0; r < nr; r++) { const int a = __builtin_amdgcn_readlane(ra, r), b = __builtin_amdgcn_readlane(rb, r); sp_l = sp_l || (pos >= a && pos <= b); }
    return (unsigned)(__ballot(sp_l) & 31ull);
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

// Rank sort of one segment of one ring by (curvature, index) — std::sort of R:409-410 with ties broken by index — for rings BEYOND the LDS
// working set (k_rot_segments sorts the others itself).  One workgroup per (ring, segment, chunk of 256 elements); sort_ind holds GLOBAL
// indices (into `full`).
__global__ __launch_bounds__(256) void k_rot_rank(const float* __restrict__ curv, RotDev P, const RotState* __restrict__ st, int* __restrict__ sort_ind) {
    __shared__ float seg[kRingLdsCap];
    const int ring = blockIdx.x, j = blockIdx.y;
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    if (ring >= P.n_scans || re - rs < 6 || ring % P.ds_rate != 0 || st->ring_count[ring] <= kRingLdsCap) return;
    const int sp = rs + (re - rs) * j / 6, ep = rs + (re - rs) * (j + 1) / 6 - 1;
    const int len = ep - sp + 1;
    if (len <= 0) return;
    if ((int)blockIdx.z * 64 >= len) return;
    const bool in_lds = len <= kRingLdsCap;
    if (in_lds) for (int m = threadIdx.x; m < len; m += 256) seg[m] = curv[sp + m];
    __syncthreads();
    // four lanes per element, each counting a quarter of the segment (interleaved by 4): the partial ranks meet in two shuffles
    const int sub = threadIdx.x & 3;
    for (int e = blockIdx.z * 64 + (threadIdx.x >> 2); e < len; e += gridDim.z * 64) {
        const float ck = in_lds ? seg[e] : curv[sp + e];
        int rank = 0;
        for (int m = sub; m < len; m += 4) { const float cm = in_lds ? seg[m] : curv[sp + m]; rank += (cm < ck || (cm == ck && m < e)) ? 1 : 0; }
        rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2);
        if (sub == 0) sort_ind[sp + rank] = sp + e;
    }
}

// Working set of the voxel ordering of one ring inside k_rot_segments.
struct OrderLds {
    unsigned key[kRingLdsCap + 8];         // dense voxel number per ring point (0xffffffff: not a candidate)
    unsigned short rstart[kRingLdsCap], rend[kRingLdsCap];   // runs in index order: first / last point
    unsigned rkey[kRingLdsCap];            //                       voxel number
    int cnt[2 * kRotBlock + 1];            // runs per bin (2048 bins on the top bits of the ring's voxel numbers), then their exclusive prefix in place
    unsigned short bidx[kRingLdsCap];      // runs grouped by bin (any order inside a bin): run number; their voxel numbers reuse `key`
    unsigned wmin[3][kRotBlock / 64], wmax[3][kRotBlock / 64];
    int scan[kRotBlock / 64 + 1];
};

// Working set of the radix ordering (k_rot_voxel_order).
struct SortLds {
    float4 pts[kRingLdsCap];
    unsigned vidx[kRingLdsCap];            // voxel index of the q-th candidate
    unsigned short cand[kRingLdsCap];      // ring-local index of the q-th candidate
    unsigned short ord_a[kRingLdsCap];     // radix-sort ping-pong: positions q in the candidate list, ordered by (voxel, q)
    unsigned short ord_b[kRingLdsCap];
    int rcnt[32 * 4 * (kRotBlock / 64)];   // radix pass: counts / offsets [digit][slot][wave]
    int scan[kRotBlock / 64 + 1];
    float red[6][kRotBlock / 64];
};

struct RotRingScratch {        // products of k_rot_segments for k_rot_ring, all indexed like `full` unless noted
    SegOut* seg_out;           // [64 * 6]
    int* ring_ncand;           // [64]
    int* sorted_k;             // the i-th run of the ring in (voxel, first index) order, at [ring_base + i]: ring-local index of its first point
    int* sorted_len;           //   its number of points (consecutive indices)
    unsigned* sorted_vox;      //   its voxel number
    unsigned long long* seg_final;   // [64 * 6] {tag << 48 | marks finally spilled into the next segment}: k_rot_segments, segment j waiting for segment j - 1 (nullptr: no waiting)
    unsigned tag;                    // this extraction's (1 .. 65535)
};

// The same ordering by a stable LSD radix sort with the voxel numbering of the ring's bounding box (what round 2 ran for every ring): second pass,
// only for scans in which a point's voxel coordinates did not fit the packed key (RotState::vox_overflow: more than 614 m / v * 0.6 from the
// sensor).  One workgroup per ring; positions (16-bit) sorted on 5-bit digits in LDS, ranks inside a wave from five ballots.
__global__ __launch_bounds__(kRotBlock) void k_rot_voxel_order(const float4* __restrict__ full, RotDev P, RotState* st, RotRingScratch X) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ring = blockIdx.x, tid = threadIdx.x;
    const int rbase = st->ring_base[ring], rcount = st->ring_count[ring];
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    if (!(ring < P.n_scans && re - rs >= 6 && ring % P.ds_rate == 0) || rcount > kRingLdsCap) return;
    const int s0 = rs - rbase, e0 = re - rbase;
    {
        SortLds& L = *reinterpret_cast<SortLds*>(smem);
        // ================= voxel ordering of the ring's candidates
        for (int k = tid; k < rcount; k += kRotBlock) L.pts[k] = full[rbase + k];
        __syncthreads();
        int n_c = 0;
        for (int k0 = s0; k0 <= e0 - 1; k0 += kRotBlock) {
            const int k = k0 + tid;
            const bool c = k <= e0 - 1 && !((double)range2(L.pts, k) < 0.25);
            int tot; const int off = block_excl_scan_1024(c ? 1 : 0, L.scan, tot);
            if (c) L.cand[n_c + off] = (unsigned short)k;
            n_c += tot;
        }
        __syncthreads();
        if (tid == 0) X.ring_ncand[ring] = n_c;
        if (n_c == 0) return;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int q = tid; q < n_c; q += kRotBlock) {
            const float4 p = L.pts[L.cand[q]];
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
        for (int q = tid; q < n_c; q += kRotBlock) {
            const float4 p = L.pts[L.cand[q]];
            const int i0 = (int)(floorf(p.x * inv) - (float)min_b[0]);
            const int i1 = (int)(floorf(p.y * inv) - (float)min_b[1]);
            const int i2 = (int)(floorf(p.z * inv) - (float)min_b[2]);
            L.vidx[q] = (unsigned)(i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1]);
            L.ord_a[q] = (unsigned short)q;
        }
        const unsigned n_vox = (unsigned)div_b[0] * (unsigned)div_b[1] * (unsigned)div_b[2];    // PCL itself rejects grids beyond 2^31 cells
        int bits = 1; while (bits < 32 && (n_vox - 1u) >> bits) bits++;
        unsigned short* src = L.ord_a;
        unsigned short* dst = L.ord_b;
        __syncthreads();
        // Ranks inside a wave come from five ballots (the lanes that hold the same digit), across waves / slots from one scan of the
        // [digit][slot][wave] count table: 4 block barriers per pass and ceil(bits / 5) passes.
        {
            const int wave = tid >> 6, lane = tid & 63;
            for (int shift = 0; shift < bits; shift += 5) {
                L.rcnt[tid] = 0; L.rcnt[tid + kRotBlock] = 0;
                __syncthreads();
                int dig[4], rk[4], qq[4];
#pragma unroll
                for (int sl = 0; sl < 4; sl++) {
                    const int i = sl * kRotBlock + tid;
                    const bool act = i < n_c;
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
                    if (i < n_c) dst[L.rcnt[(dig[sl] * 4 + sl) * (kRotBlock / 64) + wave] + rk[sl]] = (unsigned short)qq[sl];
                }
                __syncthreads();
                unsigned short* t2 = src; src = dst; dst = t2;
            }
        }
        for (int i = tid; i < n_c; i += kRotBlock) {
            const int q = src[i], k = L.cand[q];
            X.sorted_k[rbase + i] = k; X.sorted_len[rbase + i] = 1; X.sorted_vox[rbase + i] = L.vidx[q];       // (every candidate a run of its own)
        }
    }
}

// Launch 3 of 4: everything about a ring that does not need the picks of the WHOLE ring.  Work items (ring, by):
//   by < 6 — segment j:
//   * the 11-tap curvatures of its stretch of the ring (R:385-394; the six workgroups cover the ring, also of rings that are not selected:
//     the curvature array is complete);
//   * rank sort of the segment by (curvature, index) (R:409-410): one lane per element and half of the keys, 16-byte LDS broadcast reads;
//   * wave 0: greedy sharp / less-sharp / flat picks with +-5 suppression (R:401-492).  The reference runs the six segments one after
//     the other and a suppression may reach across a border (A4 iv) — only FORWARD matters, and only through the first five elements of
//     the next segment: every segment records the marks it spills, k_rot_ring checks them against the next segment's picks and redoes
//     that segment serially if one of its picks was among them (rare: a top-10 curvature within five points of both sides of a border).
//   by == 6 — the ordering pcl::VoxelGrid(ds_v) needs for the ring (R:502-508).  The grid runs over the less-flat points = the points of
//     [scanStartInd, scanEndInd) at range >= 0.5 m that were NOT picked as edge points; the picks are not known yet, so ALL candidates are
//     ordered by (voxel key, index) — a stable order, so dropping the picked ones later leaves exactly the order of the less-flat list.
//     The keys (k_rot_scatter wrote them) are binned on the top 11 bits of the ring's key range with LDS atomics (one histogram, one
//     scan, one scatter), then every key counts its exact rank among the few keys of its bin: ~6 us and 7 barriers per ring, where the
//     LSD radix sort of round 2 (4-5 passes x 6 barriers, 15-25 us; still there as k_rot_voxel_order) was the longest step of the
//     extractor.  (Counting every key against every key of the ring — no barriers at all — was tried first: 81 us x CU of compares per
//     ring, fine for 16 selected rings spread over the chip, slower than the radix sort for 64.)
// The curvature rank sort splits the keys into those before every element of the wave (count key' <= key), those behind (key' < key) and
// the wave's own stretch (index compared too): two to three operations per pair instead of five.
// One work item of k_rot_segments: by < 6 — segment `by` of `ring`; by >= 6 — voxel-ordering chunk by - 6.
__device__ void rot_stage3_item(int ring, int by, unsigned char* smem, const float4* __restrict__ full, const unsigned* __restrict__ vkey_g, const RotDev& P, RotState* st,
                                float* __restrict__ curv_g, int* __restrict__ sort_ind_g, int* __restrict__ label_g, const RotRingScratch& X) {
    const int tid = threadIdx.x;
    const int n = st->n_full;
    const int rbase = st->ring_base[ring], rcount = st->ring_count[ring];
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    const bool selected = ring < P.n_scans && re - rs >= 6 && ring % P.ds_rate == 0;      // R:402
    const bool in_lds = selected && rcount <= kRingLdsCap;                                 // longer rings: k_rot_rank + k_rot_select_big
    const int s0 = rs - rbase, e0 = re - rbase;   // local scanStartInd / scanEndInd
    if (by >= 6) {
        // ================= voxel ordering of the ring: RUNS of consecutive candidates in one voxel, ordered by (voxel, first index)
        if (!in_lds) return;
        const long long t_rb = wall_clock64();
        OrderLds& O = *reinterpret_cast<OrderLds*>(smem);
        constexpr int kPer = kRingLdsCap / kRotBlock;                                  // points per thread, CONTIGUOUS: m = kPer * tid + i
        unsigned key[kPer];
        unsigned lo3[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi3[3] = {0u, 0u, 0u};      // x, y, z fields of the candidates' keys
        {
            const uint4 raw = kPer * tid + 3 < rcount && ((rbase & 3) == 0) ? *reinterpret_cast<const uint4*>(vkey_g + rbase + kPer * tid) : make_uint4(0, 0, 0, 0);
            const bool fast = kPer * tid + 3 < rcount && ((rbase & 3) == 0);
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                const int m = kPer * tid + i;
                key[i] = fast ? (i == 0 ? raw.x : i == 1 ? raw.y : i == 2 ? raw.z : raw.w) : (m < rcount ? vkey_g[rbase + m] : 0xffffffffu);   // (k_rot_scatter: 0xffffffff = not a candidate)
                if (key[i] != 0xffffffffu) {
                    const unsigned f[3] = {key[i] & 2047u, (key[i] >> 11) & 2047u, key[i] >> 22};
#pragma unroll
                    for (int c = 0; c < 3; c++) { lo3[c] = min(lo3[c], f[c]); hi3[c] = max(hi3[c], f[c]); }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) for (int o = 32; o > 0; o >>= 1) { lo3[c] = min(lo3[c], (unsigned)__shfl_xor((int)lo3[c], o)); hi3[c] = max(hi3[c], (unsigned)__shfl_xor((int)hi3[c], o)); }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) { O.wmin[c][tid >> 6] = lo3[c]; O.wmax[c][tid >> 6] = hi3[c]; }
        }
        O.cnt[tid] = 0; O.cnt[tid + kRotBlock] = 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; c++) for (int w = 0; w < kRotBlock / 64; w++) { lo3[c] = min(lo3[c], O.wmin[c][w]); hi3[c] = max(hi3[c], O.wmax[c][w]); }
        if (lo3[0] == 0xffffffffu) return;                                             // no candidate (ring_ncand stays 0)
        // dense voxel number over the ring's box (what PCL computes; below 2^31 by the packing)
        const unsigned dx = hi3[0] - lo3[0] + 1, dy = hi3[1] - lo3[1] + 1, dz = hi3[2] - lo3[2] + 1;
        const unsigned range = dx * dy * dz - 1u;
        const int shift = max(0, 32 - __clz((int)range) - 11);                        // the bins: top 11 bits, (range >> shift) < 2048
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            if (key[i] != 0xffffffffu) key[i] = ((key[i] >> 22) - lo3[2]) * (dx * dy) + (((key[i] >> 11) & 2047u) - lo3[1]) * dx + ((key[i] & 2047u) - lo3[0]);
            O.key[kPer * tid + i] = key[i];
        }
        __syncthreads();
        // run heads and tails among this thread's points (the neighbours' keys come from LDS), numbered in index order by one scan:
        // the r-th head and the r-th tail belong to the same run
        const unsigned prev = tid > 0 ? O.key[kPer * tid - 1] : 0xffffffffu, next = tid < kRotBlock - 1 ? O.key[kPer * tid + kPer] : 0xffffffffu;
        int nh = 0, nt = 0;
        bool is_head[kPer], is_tail[kPer];
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const unsigned before = i ? key[i - 1] : prev, after = i + 1 < kPer ? key[i + 1] : next;
            is_head[i] = key[i] != 0xffffffffu && before != key[i];
            is_tail[i] = key[i] != 0xffffffffu && after != key[i];
            nh += is_head[i] ? 1 : 0; nt += is_tail[i] ? 1 : 0;
        }
        int n_runs;
        {
            int tot; const int ex = block_excl_scan_1024(nh | (nt << 16), O.scan, tot);
            n_runs = tot & 0xffff;
            int h = ex & 0xffff, t = ex >> 16;
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                if (is_head[i]) { O.rstart[h] = (unsigned short)(kPer * tid + i); O.rkey[h] = key[i]; h++; }
                if (is_tail[i]) { O.rend[t] = (unsigned short)(kPer * tid + i); t++; }
            }
        }
        __syncthreads();
        if (tid == 0) X.ring_ncand[ring] = n_runs;
        // one pass of binning of the RUNS on the top bits of the voxel number, then exact ranks by (voxel, first index) inside the bins
        int slot[kPer], rbin[kPer];
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int r = tid + i * kRotBlock;
            rbin[i] = r < n_runs ? (int)(O.rkey[r] >> shift) : -1;
            slot[i] = rbin[i] >= 0 ? atomicAdd(&O.cnt[rbin[i]], 1) : 0;
        }
        __syncthreads();
        {
            const int a = O.cnt[2 * tid], b2 = O.cnt[2 * tid + 1];
            int tot; const int ex = block_excl_scan_1024(a + b2, O.scan, tot);
            O.cnt[2 * tid] = ex; O.cnt[2 * tid + 1] = ex + a;          // (the scan's barriers lie between the reads above and these writes)
            if (tid == kRotBlock - 1) O.cnt[2 * kRotBlock] = tot;
        }
        __syncthreads();
        unsigned* bkey = O.key;                                                        // (the per-point keys are no longer needed)
#pragma unroll
        for (int i = 0; i < kPer; i++) if (rbin[i] >= 0) {
            const int r = tid + i * kRotBlock, at = O.cnt[rbin[i]] + slot[i];
            bkey[at] = O.rkey[r]; O.bidx[at] = (unsigned short)r;                    // (runs are numbered in index order: r orders equal voxels)
        }
        __syncthreads();
        if (ring == 0 && tid == 0) st->tphase[3] = (wall_clock64() - t_rb) << 32;
#pragma unroll
        for (int i = 0; i < kPer; i++) if (rbin[i] >= 0) {
            const int r = tid + i * kRotBlock;
            const unsigned rk = O.rkey[r];
            const int t0 = O.cnt[rbin[i]], t1 = O.cnt[rbin[i] + 1];
            int rank = t0;
            int t = t0;
            for (; t + 4 <= t1; t += 4) {                                               // four independent loads in flight
                const unsigned k0 = bkey[t], k1 = bkey[t + 1], k2 = bkey[t + 2], k3 = bkey[t + 3];
                const int i0 = O.bidx[t], i1 = O.bidx[t + 1], i2 = O.bidx[t + 2], i3 = O.bidx[t + 3];
                rank += ((k0 < rk || (k0 == rk && i0 < r)) ? 1 : 0) + ((k1 < rk || (k1 == rk && i1 < r)) ? 1 : 0)
                      + ((k2 < rk || (k2 == rk && i2 < r)) ? 1 : 0) + ((k3 < rk || (k3 == rk && i3 < r)) ? 1 : 0);
            }
            for (; t < t1; t++) { const unsigned bk = bkey[t]; const int bi = O.bidx[t]; rank += (bk < rk || (bk == rk && bi < r)) ? 1 : 0; }
            X.sorted_k[rbase + rank] = O.rstart[r]; X.sorted_len[rbase + rank] = (int)O.rend[r] - (int)O.rstart[r] + 1; X.sorted_vox[rbase + rank] = rk;
        }
        __syncthreads();
        if (ring == 0 && tid == 0) st->tphase[3] |= (wall_clock64() - t_rb);
        if (tid == 0) st->order_ticks[ring] = (int)(wall_clock64() - t_rb);
        return;
    }
    // ================= segment j
    const int j = by;
    if (!in_lds) {       // not selected, or beyond the LDS working set: this sixth of the ring's curvatures and labels only
        const int lo = (int)((long long)rcount * j / 6), hi = (int)((long long)rcount * (j + 1) / 6);
        for (int k = lo + tid; k < hi; k += kRotBlock) {
            const int g = rbase + k;
            curv_g[g] = (g >= 5 && g < n - 5) ? curvature11(full, g) : 0.f;
            label_g[g] = 0;                                                                 // R:393
        }
        return;
    }
    SegLds& L = *reinterpret_cast<SegLds*>(smem);
    const int sp = s0 + (e0 - s0) * j / 6, ep = s0 + (e0 - s0) * (j + 1) / 6 - 1;           // ring-local, inclusive (R:404-405)
    const int lo = j == 0 ? 0 : sp, hi = j == 5 ? rcount : ep + 1;                           // the stretch of the ring this workgroup answers for
    const int w0 = lo - 5;                                                                   // window coordinate w <-> ring-local k = w0 + w
    const int wlen = hi + 5 - w0;
    const long long t_seg = wall_clock64();
    if (ring == 0 && j == 0 && tid == 0) st->tphase[0] = t_seg;
    for (int w = tid; w < wlen; w += kRotBlock) {
        const int g = rbase + w0 + w;
        L.pts[w] = (g >= 0 && g < n) ? full[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        L.label[w] = 0; L.mark[w] = 0;
    }
    if (tid < 16) L.mark[wlen + tid] = 0;
    L.cnt[tid] = 0;
    __syncthreads();
    if (ring == 0 && j == 0 && tid == 0) st->tseg[0] = wall_clock64();
    const int spw = sp - w0, epw = ep - w0, len = ep - sp + 1;
    {   // the break tests of the suppressions, one bit per gap of the window (wlen <= kSegCap < kRotBlock: one per thread)
        const bool brk = tid + 1 < wlen && (double)gap2(L.pts, tid + 1, tid) > 0.05;
        const unsigned long long bal = __ballot(brk);
        if ((tid & 63) == 0 && (tid >> 6) < kGapWords) L.gapw[tid >> 6] = bal;
    }
    for (int w = 5 + tid; w < wlen - 5; w += kRotBlock) {
        const int g = rbase + w0 + w;
        const float c = (g >= 5 && g < n - 5) ? curvature11(L.pts, w) : 0.f;
        L.curv[w] = c;
        curv_g[g] = c;
    }
    __syncthreads();
    // Rank sort of [sp, ep] by (curvature, index) (len <= kSegCap: one element per thread).  Round 6: the elements are binned first — a curvature is a sum of
    // squares, its float bits order like its value, bin = exponent and two mantissa bits (LDS atomics, one block scan) — and every element counts its exact rank among
    // the few elements of its bin.  (Rounds 3-5 counted every key against every key: len^2 = 122 k compares at two instructions each, 6.2-6.9 us of a segment's 15;
    // a segment whose curvatures all fall into one bin — a perfectly flat wall — still costs that.)
    if (ring == 0 && j == 0 && tid == 0) st->tseg[1] = wall_clock64();
    {
        const int m = tid;
        const bool act = m < len;
        const float ck = act ? L.curv[spw + m] : 0.f;
        const int bkt = act ? (int)((__float_as_uint(ck) >> 21) & (unsigned)(kRotBlock - 1)) : 0;
        const int slot = act ? atomicAdd(&L.cnt[bkt], 1) : 0;            // (cleared with the window's labels; arrival order inside a bin is arbitrary: the ranks below do not depend on it)
        __syncthreads();
        int tot; const int ex = block_excl_scan_1024(L.cnt[tid], L.scan, tot);
        L.cnt[tid] = ex;
        if (tid == kRotBlock - 1) L.cnt[kRotBlock] = tot;
        __syncthreads();
        if (act) { const int at = L.cnt[bkt] + slot; L.key[at] = ck; L.hrank[at] = m; }
        __syncthreads();
        if (act) {
            const int t0 = L.cnt[bkt], t1 = L.cnt[bkt + 1];
            int rank = t0, t = t0;
            for (; t + 4 <= t1; t += 4) {
                const float u0 = L.key[t], u1 = L.key[t + 1], u2 = L.key[t + 2], u3 = L.key[t + 3];
                const int m0 = L.hrank[t], m1 = L.hrank[t + 1], m2 = L.hrank[t + 2], m3 = L.hrank[t + 3];
                rank += ((u0 < ck || (u0 == ck && m0 < m)) ? 1 : 0) + ((u1 < ck || (u1 == ck && m1 < m)) ? 1 : 0) + ((u2 < ck || (u2 == ck && m2 < m)) ? 1 : 0) + ((u3 < ck || (u3 == ck && m3 < m)) ? 1 : 0);
            }
            for (; t < t1; t++) { const float u = L.key[t]; const int mi = L.hrank[t]; rank += (u < ck || (u == ck && mi < m)) ? 1 : 0; }
            L.sort_ind[spw + rank] = spw + m; sort_ind_g[rbase + sp + rank] = rbase + sp + m;
        }
    }
    __syncthreads();
    const bool par_seg = (e0 - s0) >= 64;      // every segment longer than the reach of a suppression (5): spills stop in the next segment
    if (ring == 0 && j == 0 && tid == 0) st->tphase[1] = wall_clock64();
    if (tid < 64) {
        if (par_seg) {
            unsigned spill = greedy_segment_wave(L, spw, epw);
            LILI_ROT_WAVE_SYNC();
            // The picks above assumed that segment j - 1 left no mark in this one (the reference runs the six one after the other, R:401-492).  Only a pick among the
            // segment's FIRST FIVE elements can sit under such a mark: then — rare — this segment waits for what its predecessor finally spills (one 64-bit word per
            // segment, {tag | spill}, agent-scope relaxed like RotFold's; the predecessor is the workgroup before this one: resident or done) and, if a pick is hit,
            // runs again with those marks while everything still lies in LDS.  (Rounds 3-5 left this to k_rot_ring: re-staging and a second run on ITS critical path,
            // 6.5-10.7 us whenever any ring of the scan had such a pick.)  A spin that gives up publishes what it has; k_rot_ring's border check still catches the hit.
            if (X.seg_final) {
                const int q = tid;
                int d = -1;
                if (q < L.ne) d = L.edge[q] - spw; else if (q >= 16 && q - 16 < L.nf) d = L.flat[q - 16] - spw;
                const bool first5 = d >= 0 && d < 5;
                if (j > 0 && __ballot(first5)) {
                    unsigned long long w = 0ull;
                    if (tid == 0) {
                        int spins = 0;
                        do {
                            w = __hip_atomic_load(&X.seg_final[ring * 6 + j - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (++spins > (1 << 20)) w = (unsigned long long)(X.tag & 0xffffu) << 48;      // never hang: as if nothing were spilled (k_rot_ring checks again)
                        } while ((unsigned)(w >> 48) != (X.tag & 0xffffu));
                    }
                    const unsigned pspill = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(w & 31ull));
                    if (__ballot(first5 && ((pspill >> d) & 1u))) {
                        if (q < L.ne) L.label[L.edge[q]] = 0; else if (q >= 16 && q - 16 < L.nf) L.label[L.flat[q - 16]] = 0;
                        LILI_ROT_WAVE_SYNC();
                        spill = greedy_segment_wave(L, spw, epw, pspill);
                        LILI_ROT_WAVE_SYNC();
                        if (tid == 0) atomicAdd(&st->redo_segments, 1);
                    }
                }
                if (tid == 0) __hip_atomic_store(&X.seg_final[ring * 6 + j], ((unsigned long long)(X.tag & 0xffffu) << 48) | (unsigned long long)spill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            SegOut& O = X.seg_out[ring * 6 + j];
            if (tid < L.ne) O.edge[tid] = L.edge[tid] + w0;
            if (tid < L.nf) O.flat[tid] = L.flat[tid] + w0;
            if (tid == 0) { O.ne = L.ne; O.nf = L.nf; O.spill = (int)spill; O.pad = 0; }
        }
        if (ring == 0 && j == 0 && tid == 0) st->tphase[2] = wall_clock64();
    }
    __syncthreads();
    // (a nearly empty ring takes the reference's serial order through all six segments in k_rot_ring; its labels stay 0 here)
    for (int k = lo + tid; k < hi; k += kRotBlock) label_g[rbase + k] = L.label[k - w0];
    if (tid == 0) atomicMax(&st->seg_ticks[ring], (int)(wall_clock64() - t_seg));
}

// The launch: a fixed number of workgroups (two per CU fit) share the work items — 6 segments for each of the 64 rings, then the
// voxel ordering of each selected ring (which rings have work only the device knows: a grid with one workgroup per POSSIBLE item spent
// most of its time dispatching workgroups that had nothing to do, ~27 ns each).
__global__ __launch_bounds__(kRotBlock) void k_rot_segments(const float4* __restrict__ full, const unsigned* __restrict__ vkey_g, RotDev P, RotState* st, float* __restrict__ curv_g,
                                                            int* __restrict__ sort_ind_g, int* __restrict__ label_g, RotRingScratch X) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (every work item has a workgroup of its own — kStage3Blocks >= 7 * kMaxRings —: a segment's workgroup knows its item from its index and starts at once; only
    // the workgroups of the ordering items first have to find out which rings are selected)
    if (blockIdx.x < 6 * kMaxRings) { rot_stage3_item(blockIdx.x / 6, blockIdx.x % 6, smem, full, vkey_g, P, st, curv_g, sort_ind_g, label_g, X); return; }
    __shared__ int chunk_pref[kMaxRings + 1];
    if (threadIdx.x < 64) {
        const int r = threadIdx.x;
        const int rc = st->ring_count[r], span = rc - 11;                                   // candidates live in [5, rcount - 7]
        const int c = (r < P.n_scans && span >= 6 && r % P.ds_rate == 0 && rc <= kRingLdsCap && st->vox_overflow == 0) ? 1 : 0;   // (overflow: k_rot_voxel_order in a second pass)
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (r >= o) inc += t; }
        chunk_pref[r + 1] = inc;
        if (r == 0) chunk_pref[0] = 0;
    }
    __syncthreads();
    const int c = (int)blockIdx.x - 6 * kMaxRings;
    if (c >= chunk_pref[kMaxRings]) return;
    int r = 0;
    while (chunk_pref[r + 1] <= c) r++;
    rot_stage3_item(r, 6 + c - chunk_pref[r], smem, full, vkey_g, P, st, curv_g, sort_ind_g, label_g, X);
}

struct RingLds {
    SegLds seg;                              // redo of one segment / the serial run of a nearly empty ring
    signed char label[kRingLdsCap];
    float4 pts[kRingLdsCap];                 // the ring's points
    unsigned rk[kRingLdsCap + 8];            // runs in (voxel, first index) order: voxel number (+ sentinels)
    unsigned short rs[kRingLdsCap + 8], rl[kRingLdsCap + 8];   //                   first point, number of points
    SegOut so[6];
    int scan[kRotBlock / 64 + 1];
    int flag, spill, hits;
    int off[5];                              // lists of the lower rings: edge, sharp, flat, less-flat, surf
    int tot[2];                              // last ring: the scan's edge / surf totals
    unsigned long long pmask[kRingLdsCap / 64];                // bit k: ring point k is an edge pick (leaves the less-flat list)
    unsigned short task[kRingLdsCap], vcnt[kRingLdsCap];      // VoxelGrid: the o-th centroid of the ring belongs to the voxel that starts with run task[o] and has vcnt[o] points
};

// Round 6 — k_rot_ring puts a ring's lists where the SCAN's lists want them itself (no concatenation launch behind it): every ring publishes its five counts as ONE
// 64-bit word {tag of this extraction | counts} (agent-scope relaxed atomic: the word is its own flag, as in k_scan_lookback_t) and sums the words of the rings
// below it — 64 workgroups, one per CU, dispatched in index order, a ring only ever waits for lower ones; a bounded spin gives up, raises `give_up` and the host repeats
// the concatenation with k_rot_compact.  The last ring's workgroup has then seen every word: it writes the totals and the page-locked mirror of the state.
// Word: surf [12:0] | less-flat [25:13] | flat [30:26] | sharp [34:31] | edge [40:35] | beyond-LDS ring [41] | tag [63:48].
struct RotFold {
    unsigned long long* words;               // [kMaxRings]; nullptr: lists stay per ring (second passes: k_rot_compact follows)
    unsigned tag;                            // 1 .. 65535
    int* edge_idx; float4* edge_pts; int* sharp_idx; int* flat_idx; int* lessflat_idx; float4* surf; int* surf_cnt;
    RotState* mirror;                        // page-locked copy of the state as the device sees it, or nullptr
    int* give_up;                            // page-locked word behind the mirror, or nullptr
    // optional second destination of the surf / edge lists: the query arrays of a matcher slot (lili_query_sink; rows behind the lists NaN up to the capacities, the
    // slot's pose set by the last ring's workgroup) — a frame whose matcher is enqueued behind this launch needs no launch in between
    float4* q_surf; int cap_surf; float4* q_edge; int cap_edge; SlotState* q_state; double q_pose[7];
};

// Launch 4 of 4, one workgroup per ring: joins the six segments (spill check, R:401-492 order), the ring's pick lists, the less-flat list in index order
// (R:494-499) and the VoxelGrid centroids of the less-flat points in the order k_rot_segments prepared (f32 sums in list order, like pcl's CentroidPoint;
// PCL >= 1.8 semantics) — into the per-ring scratch lists and, with `F`, into the scan's lists.
__global__ __launch_bounds__(kRotBlock) void k_rot_ring(const float4* __restrict__ full, const float* __restrict__ curv_g, const int* __restrict__ sort_ind_g, RotDev P, RotState* st,
                                                        int* __restrict__ label_g, int* __restrict__ ring_edge /*[64][60]*/,
                                                        int* __restrict__ ring_sharp /*[64][12]*/, int* __restrict__ ring_flat /*[64][24]*/,
                                                        int* __restrict__ lessflat_tmp /*[n]*/, float4* __restrict__ surf_tmp /*[n]*/,
                                                        int* __restrict__ surf_cnt_tmp /*[n]*/, RotRingScratch X, RotFold F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RingLds& L = *reinterpret_cast<RingLds*>(smem);
    const int ring = blockIdx.x, tid = threadIdx.x;
    const int rbase = st->ring_base[ring], rcount = st->ring_count[ring];
    const int rs = st->ring_start[ring], re = st->ring_end[ring];
    const int n_runs = X.ring_ncand[ring];       // (k_rot_segments ordered the runs of every selected ring that fits LDS, nearly empty ones included)
    const bool selected = !(ring >= P.n_scans || re - rs < 6 || ring % P.ds_rate != 0);      // R:402
    const bool big = selected && rcount > kRingLdsCap;     // does not fit the LDS working set: k_rot_select_big takes this ring (global-memory arrays)
    if (!F.words && (!selected || big)) {
        if (big && tid == 0) atomicAdd(&st->fallback_rings, 1);
        return;
    }
    int ne = 0, nsh = 0, nfl = 0, n_lf = 0, n_out = 0;
    // what waits in registers for the offsets of the scan's lists
    unsigned lf_keep = 0; int lf_off = 0, lf_k0 = 0;
    long long t_start = 0;
    float4 my_edge = make_float4(0.f, 0.f, 0.f, 0.f);
    int pj = 0, pq = 0, p_ne = 0, p_nsh = 0, p_nfl = 0, p_mine = 0, p_minf = 0;     // this thread's slot of the pick lists: segment, position, picks of the segments before
    if (selected && !big) {
        t_start = wall_clock64();
        if (ring == kProbeRing && tid == 0) st->tphase[4] = t_start;
        const int s0 = rs - rbase, e0 = re - rbase;
        const bool par_seg = (e0 - s0) >= 64;
        {   // the ring's working set: EVERY load of the thread requested before the first is used (a `for (k = tid; k < rcount; k += 1024)` loop waits for each trip's loads
            // before it issues the next trip's: five to six dependent round trips of ~2 us where this is one)
            constexpr int kPer = kRingLdsCap / kRotBlock;
            int lab[kPer]; float4 pt[kPer]; unsigned vk[kPer]; int vs[kPer], vl[kPer];
            int so_w = 0;
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                const int k = tid + i * kRotBlock;
                const bool in = k < rcount, inr = k < n_runs;
                lab[i] = in ? label_g[rbase + k] : 0; pt[i] = in ? full[rbase + k] : make_float4(0.f, 0.f, 0.f, 0.f);
                vk[i] = inr ? X.sorted_vox[rbase + k] : 0xffffffffu; vs[i] = inr ? X.sorted_k[rbase + k] : 0; vl[i] = inr ? X.sorted_len[rbase + k] : 0;
            }
            if (par_seg && tid < 6 * (int)(sizeof(SegOut) / sizeof(int))) so_w = reinterpret_cast<const int*>(X.seg_out + ring * 6)[tid];
#pragma unroll
            for (int i = 0; i < kPer; i++) {
                const int k = tid + i * kRotBlock;
                if (k < rcount) { L.label[k] = (signed char)lab[i]; L.pts[k] = pt[i]; }
                if (k < n_runs + 4) { L.rk[k] = vk[i]; L.rs[k] = (unsigned short)vs[i]; L.rl[k] = (unsigned short)vl[i]; }      // (sentinels end the last voxel; n_runs + 4 <= rcount - 7)
            }
            if (par_seg && tid < 6 * (int)(sizeof(SegOut) / sizeof(int))) reinterpret_cast<int*>(L.so)[tid] = so_w;
        }
        if (tid == 0) L.hits = 0;
        if (tid < kRingLdsCap / 64) L.pmask[tid] = 0ull;
        __syncthreads();
        if (ring == kProbeRing && tid == 0) st->tjoin[0] = wall_clock64();
        bool relabel = false;
        // stages segment j's window (or the whole of a nearly empty ring) for a greedy run: points from the ring's copy in LDS, curvatures and sorted order as
        // k_rot_segments left them, the break bits of the window (wlen <= kSegCap: one element per thread)
        auto stage = [&](int lo, int hi, int sp, int ep) {       // ring-local [lo, hi) answered for, segment [sp, ep]
            const int w0 = lo - 5, wlen = hi + 5 - w0;
            for (int w = tid; w < wlen; w += kRotBlock) {
                const int kk = w0 + w;
                const bool in = kk >= 0 && kk < rcount;
                L.seg.pts[w] = in ? L.pts[kk] : make_float4(0.f, 0.f, 0.f, 0.f);
                L.seg.curv[w] = in ? curv_g[rbase + kk] : 0.f;
                L.seg.label[w] = 0; L.seg.mark[w] = 0;
            }
            for (int k = sp + tid; k <= ep; k += kRotBlock) L.seg.sort_ind[k - w0] = sort_ind_g[rbase + k] - rbase - w0;
            if (tid < 16) L.seg.mark[wlen + tid] = 0;
            {
                const int ka = w0 + tid, kb = ka + 1;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 pa = (tid < wlen && ka >= 0 && ka < rcount) ? L.pts[ka] : z, pb = (tid + 1 < wlen && kb >= 0 && kb < rcount) ? L.pts[kb] : z;
                const float dX = pb.x - pa.x, dY = pb.y - pa.y, dZ = pb.z - pa.z;
                const bool brk = tid + 1 < wlen && (double)(dX * dX + dY * dY + dZ * dZ) > 0.05;
                const unsigned long long bal = __ballot(brk);
                if ((tid & 63) == 0 && (tid >> 6) < kGapWords) L.seg.gapw[tid >> 6] = bal;
            }
            return w0;
        };
        if (!par_seg) {                             // a nearly empty ring (< 75 points): the reference's order, one shared mark array
            const int w0 = stage(0, rcount, s0, e0 - 1);
            __syncthreads();
            if (tid == 0) {
                for (int j = 0; j < 6; j++) {
                    greedy_segment(L.seg, s0 + (e0 - s0) * j / 6 - w0, s0 + (e0 - s0) * (j + 1) / 6 - 1 - w0, 0u);
                    SegOut& O = L.so[j];
                    for (int q = 0; q < L.seg.ne; q++) O.edge[q] = L.seg.edge[q] + w0;
                    for (int q = 0; q < L.seg.nf; q++) O.flat[q] = L.seg.flat[q] + w0;
                    O.ne = L.seg.ne; O.nf = L.seg.nf;
                }
            }
            __syncthreads();
            for (int k = tid; k < rcount; k += kRotBlock) L.label[k] = L.seg.label[k - w0];
            relabel = true;
            __syncthreads();
        } else {
            // Did a pick of segment j fall on a mark the picks of segment j - 1 spilled across the border?  All five borders at once (one pick per thread); only then
            // the walk over the borders from the first such segment on: a redo changes what THAT segment spills — its successor is checked again —, nothing else.
            // `pick_hit(j, q)`: is pick q of segment j (q < 10: edge, else flat) among the marks segment j - 1 spills?
            const auto pick_hit = [&](int j, int q) -> bool {
                const unsigned spill = (unsigned)L.so[j - 1].spill;
                const int sp = s0 + (e0 - s0) * j / 6;
                int d = -1;
                if (q < kSegEdge) { if (q < L.so[j].ne) d = L.so[j].edge[q] - sp; }
                else if (q - kSegEdge < kSegFlat) { if (q - kSegEdge < L.so[j].nf) d = L.so[j].flat[q - kSegEdge] - sp; }
                return spill && d >= 0 && d < 5 && ((spill >> d) & 1u);
            };
            if (tid < 5 * 16 && pick_hit(1 + (tid >> 4), tid & 15)) atomicOr(&L.hits, 1 << (1 + (tid >> 4)));
            __syncthreads();
            if (ring == kProbeRing && tid == 0) st->tjoin[1] = wall_clock64();
            const int hits = L.hits;
            int j = hits ? __ffs(hits) - 1 : 6;
            while (j < 6) {                          // (uniform)
                const int sp = s0 + (e0 - s0) * j / 6, ep = s0 + (e0 - s0) * (j + 1) / 6 - 1;
                const int w0 = stage(sp, j == 5 ? rcount : ep + 1, sp, ep);
                __syncthreads();
                if (tid < 64) {                      // wave 0: the segment again, with the marks of its predecessor; its lists and labels replaced; is the next border hit now?
                    const unsigned spill = greedy_segment_wave(L.seg, sp - w0, ep - w0, (unsigned)L.so[j - 1].spill);
                    LILI_ROT_WAVE_SYNC();
                    SegOut& O = L.so[j];
                    const int one = O.ne, onf = O.nf;
                    if (tid < one) L.label[O.edge[tid]] = 0;
                    if (tid >= 16 && tid - 16 < onf) L.label[O.flat[tid - 16]] = 0;
                    LILI_ROT_WAVE_SYNC();
                    if (tid < L.seg.ne) { const int e = L.seg.edge[tid]; O.edge[tid] = e + w0; L.label[e + w0] = L.seg.label[e]; }
                    if (tid >= 16 && tid - 16 < L.seg.nf) { const int f = L.seg.flat[tid - 16]; O.flat[tid - 16] = f + w0; L.label[f + w0] = L.seg.label[f]; }
                    if (tid == 0) { O.ne = L.seg.ne; O.nf = L.seg.nf; O.spill = (int)spill; atomicAdd(&st->redo_segments, 1); }
                    LILI_ROT_WAVE_SYNC();
                    int next = 6;
                    if (j < 5) {
                        const bool h = tid < 16 && pick_hit(j + 1, tid);
                        const int later = hits & ~((2 << (j + 1)) - 1);           // borders behind j + 1: their predecessors are unchanged
                        next = __ballot(h) ? j + 1 : (later ? __ffs(later) - 1 : 6);
                    }
                    if (tid == 0) L.flag = next;
                }
                relabel = true;
                __syncthreads();
                j = L.flag;          // (the next trip writes it behind its own first barrier)
            }
        }
        if (ring == kProbeRing && tid == 0) st->tjoin[2] = wall_clock64();
        // ---- the ring's pick lists in the reference's push order (segment by segment): one pick per thread
        {
            int sne[6], snf[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { sne[j] = L.so[j].ne; snf[j] = L.so[j].nf; }
            pj = min(tid >> 4, 5); pq = tid & 15;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                ne += sne[j]; nsh += min(sne[j], 2); nfl += snf[j];
                if (j < pj) { p_ne += sne[j]; p_nsh += min(sne[j], 2); p_nfl += snf[j]; }
                if (j == pj) { p_mine = sne[j]; p_minf = snf[j]; }
            }
            if (tid >= 6 * 16) { p_mine = 0; p_minf = 0; }
            if (pq < kSegEdge) {
                if (pq < p_mine) {
                    const int k = L.so[pj].edge[pq], g = rbase + k;
                    if (pq < 2) ring_sharp[ring * kRingSharpCap + p_nsh + pq] = g;
                    ring_edge[ring * kRingEdgeCap + p_ne + pq] = g;
                    atomicOr(&L.pmask[k >> 6], 1ull << (k & 63));          // (read in step 1 of the VoxelGrid, behind the barriers of the less-flat scan)
                }
            } else if (pq - kSegEdge < p_minf) ring_flat[ring * kRingFlatCap + p_nfl + pq - kSegEdge] = rbase + L.so[pj].flat[pq - kSegEdge];
            if (tid == 0) { st->ring_nedge[ring] = ne; st->ring_nsharp[ring] = nsh; st->ring_nflat[ring] = nfl; }
        }
        if (relabel) for (int k = tid; k < rcount; k += kRotBlock) label_g[rbase + k] = L.label[k];
        // ---- less-flat list in index order (R:494-499): every thread a contiguous stretch of the candidates, ONE block scan
        if (ring == kProbeRing && tid == 0) st->tphase[5] = wall_clock64();
        {
            const int per = (e0 - s0 + kRotBlock - 1) / kRotBlock;            // <= 4: the ring fits kRingLdsCap
            lf_k0 = s0 + per * tid;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < kRingLdsCap / kRotBlock; i++) {
                const int k = lf_k0 + i;
                if (i < per && k <= e0 - 1 && L.label[k] <= 0 && !((double)range2(L.pts[k]) < 0.25)) { lf_keep |= 1u << i; cnt++; }
            }
            lf_off = block_excl_scan_1024(cnt, L.scan, n_lf);
            int o = rbase + lf_off;
#pragma unroll
            for (int i = 0; i < kRingLdsCap / kRotBlock; i++) if ((lf_keep >> i) & 1u) lessflat_tmp[o++] = rbase + lf_k0 + i;
        }
        if (tid == 0) st->ring_nlf[ring] = n_lf;
        // the edge picks leave the ring's copy (nobody reads it between the scan above and step 2 of the VoxelGrid, which then adds +0.0f for them); their points stay
        // with the threads that write the pick lists
        if (pq < kSegEdge && pq < p_mine) { const int k = L.so[pj].edge[pq]; my_edge = L.pts[k]; L.pts[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
        // ---- pcl::VoxelGrid(ds_v) on the less-flat points (R:502-508): the runs of candidates in (voxel, first index) order — a voxel's points in list order are
        // its runs one after the other — minus the picked points (R:494-499).  Step 1 here: which voxels have a centroid (their members' labels only) and where it
        // goes — every thread a contiguous stretch of the runs, ONE block scan.  Step 2 (the sums) follows the exchange of the counts below.
        if (ring == kProbeRing && tid == 0) st->tphase[6] = wall_clock64();
        {
            const int rper = (n_runs + kRotBlock - 1) / kRotBlock;          // <= 4
            int cn[kRingLdsCap / kRotBlock];
            int nout_t = 0;
#pragma unroll
            for (int c = 0; c < kRingLdsCap / kRotBlock; c++) {
                const int i = rper * tid + c;
                const bool first = c < rper && i < n_runs && (i == 0 || L.rk[i - 1] != L.rk[i]);     // first run of its voxel
                int cnt = 0;
                if (first) {
                    const unsigned vox = L.rk[i];
                    unsigned nk = vox; int na = L.rs[i], nl = L.rl[i];          // (the next run's descriptor is loaded while this run is counted)
                    for (int j = i; nk == vox; j++) {
                        const int a = na, e = na + nl;                           // the run's points [a, e): its length minus the edge picks among them (bits of pmask)
                        nk = L.rk[j + 1]; na = L.rs[j + 1]; nl = L.rl[j + 1];
                        cnt += e - a;
                        const int w0 = a >> 6, w1 = (e - 1) >> 6;
                        for (int w = w0; w <= w1; w++) {
                            unsigned long long mk = L.pmask[w];
                            if (!mk) continue;
                            const int lo = w == w0 ? (a & 63) : 0, hi = w == w1 ? ((e - 1) & 63) : 63;
                            mk >>= lo;
                            if (hi - lo < 63) mk &= (1ull << (hi - lo + 1)) - 1ull;
                            cnt -= __popcll(mk);
                        }
                    }
                }
                cn[c] = cnt;                                  // (a voxel whose points were all picked as edge points has no centroid)
                nout_t += cnt > 0 ? 1 : 0;
            }
            int o = block_excl_scan_1024(nout_t, L.scan, n_out);
#pragma unroll
            for (int c = 0; c < kRingLdsCap / kRotBlock; c++) if (cn[c] > 0) { L.task[o] = (unsigned short)(rper * tid + c); L.vcnt[o] = (unsigned short)cn[c]; o++; }
        }
        if (tid == 0) st->ring_nsurf[ring] = n_out;
        if (ring == kProbeRing && tid == 0) st->tphase[7] = wall_clock64();
    }
    // ---- the scan's lists: this ring's counts out, the lower rings' counts in
    if (F.words) {
        const unsigned long long own = ((unsigned long long)(F.tag & 0xffffu) << 48) | ((unsigned long long)(big ? 1 : 0) << 41) | ((unsigned long long)ne << 35) | ((unsigned long long)nsh << 31)
                                     | ((unsigned long long)nfl << 26) | ((unsigned long long)n_lf << 13) | (unsigned long long)n_out;
        if (tid == 0) __hip_atomic_store(&F.words[ring], own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 64 && ((selected && !big) || ring == kMaxRings - 1)) {
            unsigned long long w = tid == ring ? own : 0ull;
            if (tid < ring) {
                int spins = 0;
                do {
                    w = __hip_atomic_load(&F.words[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > (1 << 20)) {      // cannot happen while lower rings run; never hang the GPU — the host repeats the concatenation (k_rot_compact)
                        atomicOr(&st->fold_failed, 1);
                        if (F.give_up) *F.give_up = 1;
                        w = (unsigned long long)(F.tag & 0xffffu) << 48;
                    }
                } while ((unsigned)(w >> 48) != (F.tag & 0xffffu));
            }
            const int f[6] = {(int)((w >> 35) & 63u), (int)((w >> 31) & 15u), (int)((w >> 26) & 31u), (int)((w >> 13) & 0x1fffu), (int)(w & 0x1fffu), (int)((w >> 41) & 1u)};
            int below[6], all[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                int b = tid < ring ? f[c] : 0, a = tid <= ring ? f[c] : 0;
                for (int o = 32; o > 0; o >>= 1) { b += __shfl_xor(b, o); a += __shfl_xor(a, o); }
                below[c] = b; all[c] = a;
            }
            if (tid < 5) L.off[tid] = tid == 0 ? below[0] : tid == 1 ? below[1] : tid == 2 ? below[2] : tid == 3 ? below[3] : below[4];
            if (ring == kMaxRings - 1) {      // every word has been seen here: totals, and the state's page-locked copy (fields other workgroups of THIS launch write come from the words)
                if (tid == 0) { st->n_edge = all[0]; st->n_sharp = all[1]; st->n_flat = all[2]; st->n_lessflat = all[3]; st->n_surf = all[4]; st->fallback_rings = all[5]; }
                if (F.mirror) {
                    const int* src = reinterpret_cast<const int*>(st);
                    int* dst = reinterpret_cast<int*>(F.mirror);
                    constexpr int kHead = (int)(offsetof(RotState, ring_nedge) / sizeof(int)), kTail = (int)(offsetof(RotState, vox_overflow) / sizeof(int));
                    for (int k = tid; k < kHead; k += 64) dst[k] = src[k];                                   // first / last point, ring tables (k_rot_classify, k_rot_scatter)
                    for (int k = kTail + tid; k < (int)(sizeof(RotState) / sizeof(int)); k += 64) dst[k] = src[k];      // overflow flag; diagnostics (those of this launch may be stale)
                    F.mirror->ring_nedge[tid] = f[0]; F.mirror->ring_nsharp[tid] = f[1]; F.mirror->ring_nflat[tid] = f[2]; F.mirror->ring_nlf[tid] = f[3]; F.mirror->ring_nsurf[tid] = f[4];
                    if (tid == 0) {
                        F.mirror->n_edge = all[0]; F.mirror->n_sharp = all[1]; F.mirror->n_flat = all[2]; F.mirror->n_lessflat = all[3]; F.mirror->n_surf = all[4];
                        F.mirror->fallback_rings = all[5];
                    }
                }
                if (tid == 0) { L.tot[0] = all[0]; L.tot[1] = all[4]; }
            }
        }
        if (ring == kMaxRings - 1 && (F.q_surf || F.q_edge || F.q_state)) {      // the matcher slot behind this launch: padding rows and the pose (lili_s2m_pose_set's fields)
            __syncthreads();
            const int tot_e = L.tot[0], tot_s = L.tot[1];
            const float qn = __builtin_nanf("");
            if (F.q_surf) for (int k = tot_s + tid; k < F.cap_surf; k += kRotBlock) F.q_surf[k] = make_float4(qn, qn, qn, 0.f);
            if (F.q_edge) for (int k = tot_e + tid; k < F.cap_edge; k += kRotBlock) F.q_edge[k] = make_float4(qn, qn, qn, 0.f);
            if (F.q_state) {
                if (tid < 7) F.q_state->pose[tid] = F.q_pose[tid];
                if (tid < 6) F.q_state->last_delta[tid] = 0.0;
                if (tid == 7) { F.q_state->n_res[0] = 0; F.q_state->n_res[1] = 0; F.q_state->gn_status = 0; F.q_state->iters = 0; F.q_state->cnt_word = 0ull; }
            }
        }
    }
    if (!selected || big) return;
    __syncthreads();              // (L.vo; with F.words also L.off)
    // ---- step 2 of the VoxelGrid: the centroids.  CentroidPoint's f32 sums are sequential by definition, so a ring lasts as long as its fullest voxel (next to the
    // sensor a 0.6 m voxel holds 100-200 points of a ring).  FOUR lanes per centroid, a component each: one load and ONE dependent addition per member — the picked
    // members were zeroed in the ring's copy above (adding +0.0f is exact: a sum that starts at +0.0f is never -0.0f), the count comes from step 1.  Rounds 3-5: one lane
    // per voxel, label tests and four sums per member: 3-12 us per ring.
    {
        const int quad = tid >> 2, comp = tid & 3;
        const float* PC = reinterpret_cast<const float*>(L.pts);
        float* tmp_c = reinterpret_cast<float*>(surf_tmp + rbase);
        float* fin_c = F.words ? reinterpret_cast<float*>(F.surf + L.off[4]) : nullptr;
        for (int o = quad; o < n_out; o += kRotBlock / 4) {
            const int i = L.task[o], cnt = L.vcnt[o];
            const unsigned vox = L.rk[i];
            unsigned nk = vox; int na = L.rs[i], nl = L.rl[i];
            float s = 0.f;
            for (int j = i; nk == vox; j++) {
                const int a = na, e = na + nl;
                nk = L.rk[j + 1]; na = L.rs[j + 1]; nl = L.rl[j + 1];
                int k = a;
                for (; k + 8 <= e; k += 8) {
                    const float v0 = PC[4 * k + comp], v1 = PC[4 * k + 4 + comp], v2 = PC[4 * k + 8 + comp], v3 = PC[4 * k + 12 + comp];
                    const float v4 = PC[4 * k + 16 + comp], v5 = PC[4 * k + 20 + comp], v6 = PC[4 * k + 24 + comp], v7 = PC[4 * k + 28 + comp];
                    s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
                }
                for (; k < e; k++) s += PC[4 * k + comp];
            }
            const float c = s / (float)cnt;
            tmp_c[4 * o + comp] = c;
            if (comp == 0) surf_cnt_tmp[rbase + o] = cnt;
            if (fin_c) {
                fin_c[4 * o + comp] = c; if (comp == 0) F.surf_cnt[L.off[4] + o] = cnt;
                if (F.q_surf && L.off[4] + o < F.cap_surf) reinterpret_cast<float*>(F.q_surf + L.off[4] + o)[comp] = c;
            }
        }
    }
    if (tid == 0) st->ring_ticks[ring] = (int)(wall_clock64() - t_start);
    if (!F.words) return;
    if (pq < kSegEdge) {
        if (pq < p_mine) {
            const int k = L.so[pj].edge[pq], at = L.off[0] + p_ne + pq;
            F.edge_idx[at] = rbase + k; F.edge_pts[at] = my_edge;
            if (F.q_edge && at < F.cap_edge) F.q_edge[at] = my_edge;
            if (pq < 2) F.sharp_idx[L.off[1] + p_nsh + pq] = rbase + k;
        }
    } else if (pq - kSegEdge < p_minf) F.flat_idx[L.off[2] + p_nfl + pq - kSegEdge] = rbase + L.so[pj].flat[pq - kSegEdge];
    {
        int o = L.off[3] + lf_off;
#pragma unroll
        for (int i = 0; i < kRingLdsCap / kRotBlock; i++) if ((lf_keep >> i) & 1u) F.lessflat_idx[o++] = rbase + lf_k0 + i;
    }
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
                                                     int* __restrict__ lessflat_idx, float4* __restrict__ surf, int* __restrict__ surf_cnt,
                                                     RotState* __restrict__ mirror /*optional: page-locked host copy of *st, written by the first workgroup (round 4: no copy launch before the call's synchronisation)*/) {
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
    if (r == 0 && mirror) {          // every other field of *st was final before this launch; the five totals are thread 0's, ordered by the barrier
        __syncthreads();
        const int* src = reinterpret_cast<const int*>(st);
        int* dst = reinterpret_cast<int*>(mirror);
        for (int k = threadIdx.x; k < (int)(sizeof(RotState) / sizeof(int)); k += blockDim.x) dst[k] = src[k];
    }
    const int rb = st->ring_base[r];
    for (int k = threadIdx.x; k < st->ring_nedge[r]; k += blockDim.x) { int g = ring_edge[r * kRingEdgeCap + k]; edge_idx[off[0] + k] = g; edge_pts[off[0] + k] = full[g]; }
    for (int k = threadIdx.x; k < st->ring_nsharp[r]; k += blockDim.x) sharp_idx[off[1] + k] = ring_sharp[r * kRingSharpCap + k];
    for (int k = threadIdx.x; k < st->ring_nflat[r]; k += blockDim.x) flat_idx[off[2] + k] = ring_flat[r * kRingFlatCap + k];
    for (int k = threadIdx.x; k < st->ring_nlf[r]; k += blockDim.x) lessflat_idx[off[3] + k] = lessflat_tmp[rb + k];
    for (int k = threadIdx.x; k < st->ring_nsurf[r]; k += blockDim.x) { surf[off[4] + k] = surf_tmp[rb + k]; surf_cnt[off[4] + k] = surf_cnt_tmp[rb + k]; }
}

// Round 4 — the feature lists into a caller's PAGE-LOCKED buffers by a kernel of the library's own, before the host has seen the counts (-23 us per call: the count round
// trip, two sized copies and a second synchronisation): `count` records each (the counts are read where k_rot_compact left them), rows of `stride` bytes whose first 16 are written.
__global__ __launch_bounds__(256) void k_rot_send(const RotState* __restrict__ st, const float4* __restrict__ edge_pts, char* __restrict__ edge_dst, int edge_cap, int edge_stride,
                                                  const float4* __restrict__ surf, char* __restrict__ surf_dst, int surf_cap, int surf_stride) {
    const int ne = edge_dst ? min(st->n_edge, edge_cap) : 0, ns = surf_dst ? min(st->n_surf, surf_cap) : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ne + ns; i += gridDim.x * blockDim.x) {
        if (i < ne) *reinterpret_cast<float4*>(edge_dst + (size_t)i * edge_stride) = edge_pts[i];
        else *reinterpret_cast<float4*>(surf_dst + (size_t)(i - ne) * surf_stride) = surf[i - ne];
    }
}

}  // namespace lili

// ================================================================================================
// C ABI
// ================================================================================================
namespace lili_detail {
struct RotBuffers {
    DevBuf in, scan_id, ori_raw, block_hist, block_half, state, full, full_src, curv, label, sort_ind;
    DevBuf vkey, seg_out, ring_ncand, sorted_k, sorted_vox, sorted_len;      // k_rot_scatter -> k_rot_segments -> k_rot_ring
    DevBuf ring_edge, ring_sharp, ring_flat, lessflat_tmp, surf_tmp, surf_cnt_tmp;
    DevBuf edge_idx, edge_pts, sharp_idx, flat_idx, lessflat_idx, surf, surf_cnt;
    DevBuf fold_words;          // k_rot_ring: one status word per ring (RotFold); behind them k_rot_segments' word per segment (RotRingScratch::seg_final)
    unsigned fold_tag = 0;
    bool prev_ok = false, redone = false;      // the previous extraction completed (prev_edge / prev_surf: its list lengths); the last completion ran a second pass (lists rewritten)
    int prev_edge = 0, prev_surf = 0;
    DevBuf big_mark, big_vidx, big_ord_a, big_ord_b, big_rcnt;   // working set of rings beyond the LDS budget (k_rot_select_big)
    lili::RotState host{};
    lili::RotState* h_state = nullptr;       // page-locked mirror of the device state, written by k_rot_compact; h_state_dev: the same memory as the device sees it
    lili::RotState* h_state_dev = nullptr;
    int n_in = 0;
    bool have = false;
    // an extraction that has been enqueued but not completed (lili_extract_rot_enqueue / _complete: the front-end frame pipeline overlaps its kernels with the local map's)
    struct Pending { bool on = false; lili_rot_params params{}; double q_imu[4] = {1, 0, 0, 0}, q_lb[4] = {1, 0, 0, 0}; lili::RotRingScratch X{}; bool full_early = false, sent_edge = false, sent_surf = false;
                     unsigned long long gen = 0; } pend;
    void release() {
        for (DevBuf* b : {&in, &scan_id, &ori_raw, &block_hist, &block_half, &state, &full, &full_src, &curv, &label, &sort_ind, &vkey, &seg_out, &ring_ncand, &sorted_k, &sorted_vox,
                          &sorted_len, &big_mark, &big_vidx, &big_ord_a, &big_ord_b, &big_rcnt, &ring_edge, &ring_sharp, &ring_flat,
                          &lessflat_tmp, &surf_tmp, &surf_cnt_tmp, &edge_idx, &edge_pts, &sharp_idx, &flat_idx, &lessflat_idx, &surf, &surf_cnt, &fold_words}) b->release();
        if (h_state) { (void)hipHostFree(h_state); h_state = nullptr; h_state_dev = nullptr; }
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

// The extraction up to (not including) its synchronisation: every kernel of the first pass is on the context's stream, the state travels to the page-locked mirror
// with the concatenation kernel.  rot_complete takes the counts (and runs the rare second passes).  `side`: the side stream a copy into the caller's full-cloud
// buffer was started on (the caller of this function drains it on every exit), or nullptr.
static int rot_enqueue(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params,
                       lili_feature_out* full, lili_feature_out* edge, lili_feature_out* surf, hipStream_t* side, const lili_query_sink* sink = nullptr) {
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
    bool sent_edge = false, sent_surf = false;      // the feature lists already lie in the caller's (page-locked) buffers
    // a scan that is already in HBM as float4 rows is read in place (no staging copy, one launch less)
    const bool in_place = scan->mem == LILI_MEM_DEVICE && scan->stride == sizeof(float4) && scan->aux_offset == 12 && (reinterpret_cast<uintptr_t>(scan->data) & 15) == 0;
    int rc = in_place ? LILI_OK : lili_ingest_cloud(ctx, scan, R->in);
    if (rc != LILI_OK) return rc;
    const int n = (int)scan->n;
    R->n_in = n;
    HIPCHK(R->state.ensure(sizeof(RotState)));
    RotState* st = R->state.as<RotState>();
    if (!R->h_state) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&R->h_state), sizeof(RotState) + 64, hipHostMallocDefault));      // (+ the word k_rot_ring raises when a look-back gave up)
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, R->h_state, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }
        R->h_state_dev = static_cast<RotState*>(d);
    }
    if (n == 0) HIPCHK(hipMemsetAsync(st, 0, sizeof(RotState), ctx->stream));      // (the kernels below write every field they later read; an empty scan launches none)
    RotRingScratch X{};
    if (n > 0) {
        const size_t cap = (size_t)n;
        const int nb = nblocks(n, kRotBlock);
        HIPCHK(R->scan_id.ensure(cap)); HIPCHK(R->ori_raw.ensure(cap * 4));
        HIPCHK(R->block_hist.ensure((size_t)nb * kMaxRings * 4)); HIPCHK(R->block_half.ensure((size_t)nb * 4));
        HIPCHK(R->full.ensure(cap * 16)); HIPCHK(R->full_src.ensure(cap * 4)); HIPCHK(R->curv.ensure(cap * 4)); HIPCHK(R->label.ensure(cap * 4)); HIPCHK(R->sort_ind.ensure(cap * 4));
        HIPCHK(R->seg_out.ensure(kMaxRings * 6 * sizeof(SegOut))); HIPCHK(R->ring_ncand.ensure(kMaxRings * 4));
        HIPCHK(R->sorted_k.ensure(cap * 4)); HIPCHK(R->sorted_vox.ensure(cap * 4)); HIPCHK(R->sorted_len.ensure(cap * 4)); HIPCHK(R->vkey.ensure((cap + 16) * 4));
        HIPCHK(R->ring_edge.ensure(kMaxRings * kRingEdgeCap * 4)); HIPCHK(R->ring_sharp.ensure(kMaxRings * kRingSharpCap * 4)); HIPCHK(R->ring_flat.ensure(kMaxRings * kRingFlatCap * 4));
        HIPCHK(R->lessflat_tmp.ensure(cap * 4)); HIPCHK(R->surf_tmp.ensure(cap * 16)); HIPCHK(R->surf_cnt_tmp.ensure(cap * 4));
        HIPCHK(R->edge_idx.ensure(kMaxRings * kRingEdgeCap * 4)); HIPCHK(R->edge_pts.ensure(kMaxRings * kRingEdgeCap * 16));
        HIPCHK(R->sharp_idx.ensure(kMaxRings * kRingSharpCap * 4)); HIPCHK(R->flat_idx.ensure(kMaxRings * kRingFlatCap * 4));
        HIPCHK(R->lessflat_idx.ensure(cap * 4)); HIPCHK(R->surf.ensure(cap * 16)); HIPCHK(R->surf_cnt.ensure(cap * 4));
        if (!R->fold_words.p) { HIPCHK(R->fold_words.ensure(kMaxRings * 7 * 8)); HIPCHK(hipMemsetAsync(R->fold_words.p, 0, kMaxRings * 7 * 8, ctx->stream)); }
        X.seg_out = R->seg_out.as<SegOut>(); X.ring_ncand = R->ring_ncand.as<int>(); X.sorted_k = R->sorted_k.as<int>(); X.sorted_vox = R->sorted_vox.as<unsigned>();
        X.sorted_len = R->sorted_len.as<int>();
        R->fold_tag = R->fold_tag >= 0xffffu ? 1u : R->fold_tag + 1u;      // tag of this extraction's status words (k_rot_segments, k_rot_ring)
        X.seg_final = ctx->rot_segment_wait ? R->fold_words.as<unsigned long long>() + kMaxRings : nullptr; X.tag = R->fold_tag;
        RotDev P{};
        P.n_scans = params->n_scans; P.ds_rate = params->ds_rate; P.ds_v = params->ds_v; P.near_thres = params->near_range;
        P.atan_mode = ctx->rot_atan;
        for (int i = 0; i < 4; i++) { P.q_imu[i] = q_imu[i]; P.q_lb[i] = q_lb[i]; }
        const float4* in = in_place ? static_cast<const float4*>(scan->data) : R->in.as<float4>();
        // four launches (round 2: nine, rounds 3-5: five): classify | scatter | segments + voxel ordering (7 workgroups per ring) | ring (writes the scan's lists itself since round 6)
        hipLaunchKernelGGL(k_rot_classify, dim3(nb), dim3(kRotBlock), 0, ctx->stream, in, n, P, st, R->scan_id.as<signed char>(), R->ori_raw.as<float>(),
                           R->block_hist.as<int>(), R->block_half.as<int>());
        hipLaunchKernelGGL(k_rot_scatter, dim3(nb), dim3(kRotBlock), 0, ctx->stream, in, n, nb, R->scan_id.as<signed char>(), R->ori_raw.as<float>(), P, st,
                           R->block_hist.as<int>(), R->block_half.as<int>(), R->full.as<float4>(), R->full_src.as<int>(), R->vkey.as<unsigned>(), X.ring_ncand);
        // the deskewed cloud is final here: its copy to the host (3.2 MB for a 200 k-point scan, ~60 us) runs on a side stream under the feature
        // selection instead of behind it.  All n entries travel (the count is known only at the end); entries behind `count` are unspecified.
        // Only into PAGE-LOCKED memory (lili_host_alloc / hipHostMalloc / hipHostRegister): a copy into pageable memory is staged by the runtime and
        // blocks the host right here, before k_rot_segments is even launched — then the plain copy at the end is the better one.
        const bool full_pinned = full && full->data && full->mem == LILI_MEM_HOST && lili_pinned_dev_ptr(full->data, 16) != nullptr;
        if (full_pinned && (full->stride == 0 || full->stride == sizeof(float4)) && full->capacity > 0) {
            if (!ctx->fork_ev) HIPCHK(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
            if (!ctx->side[1]) HIPCHK(hipStreamCreateWithFlags(&ctx->side[1], hipStreamNonBlocking));
            if (!ctx->join_ev[1]) HIPCHK(hipEventCreateWithFlags(&ctx->join_ev[1], hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->fork_ev, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ctx->side[1], ctx->fork_ev, 0));
            // (the runtime's copy: a thin copy kernel of the library's own was measured 21 us per call slower)
            HIPCHK(hipMemcpyAsync(full->data, R->full.as<float4>(), std::min((size_t)n, full->capacity) * sizeof(float4), hipMemcpyDeviceToHost, ctx->side[1]));
            *side = ctx->side[1];
            HIPCHK(hipEventRecord(ctx->join_ev[1], ctx->side[1]));
            full_early = true;
        }
        hipLaunchKernelGGL(k_rot_segments, dim3(kStage3Blocks), dim3(kRotBlock), std::max(sizeof(SegLds), sizeof(OrderLds)), ctx->stream, R->full.as<float4>(), R->vkey.as<unsigned>(), P, st, R->curv.as<float>(),
                           R->sort_ind.as<int>(), R->label.as<int>(), X);
        if (ctx->rot_fold) {   // the ring stage writes the scan's lists itself (RotFold): four launches
            RotFold F{};
            F.words = R->fold_words.as<unsigned long long>();
            F.tag = R->fold_tag;
            F.edge_idx = R->edge_idx.as<int>(); F.edge_pts = R->edge_pts.as<float4>(); F.sharp_idx = R->sharp_idx.as<int>(); F.flat_idx = R->flat_idx.as<int>();
            F.lessflat_idx = R->lessflat_idx.as<int>(); F.surf = R->surf.as<float4>(); F.surf_cnt = R->surf_cnt.as<int>();
            F.mirror = R->h_state_dev;
            F.give_up = R->h_state_dev ? reinterpret_cast<int*>(R->h_state_dev + 1) : nullptr;
            if (sink) { F.q_surf = sink->q_surf; F.cap_surf = sink->cap_surf; F.q_edge = sink->q_edge; F.cap_edge = sink->cap_edge; F.q_state = sink->state; for (int i = 0; i < 7; i++) F.q_pose[i] = sink->pose[i]; }
            if (R->h_state) *reinterpret_cast<volatile int*>(R->h_state + 1) = 0;
            hipLaunchKernelGGL(k_rot_ring, dim3(kMaxRings), dim3(kRotBlock), sizeof(RingLds), ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                               R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                               R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), X, F);
        } else {               // option "rot_fold" = 0 (A/B, and the path a ring's given-up look-back falls back to): per-ring lists, then the concatenation launch of rounds 3-5
            if (R->h_state) *reinterpret_cast<volatile int*>(R->h_state + 1) = 0;
            hipLaunchKernelGGL(k_rot_ring, dim3(kMaxRings), dim3(kRotBlock), sizeof(RingLds), ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                               R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                               R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), X, RotFold{});
            hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                               R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                               R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>(), R->h_state_dev);
        }
        HIPCHK(hipGetLastError());
        // page-locked feature buffers are written right behind the concatenation, `count` records each, before the host has seen the counts: the state's read-back below
        // is then the call's only synchronisation (it was: read-back, two sized copies, second synchronisation).  The second passes further down redo the copies.
        const auto pinned16 = [](const lili_feature_out* o) -> char* {
            if (!o || !o->data || o->capacity == 0 || o->mem != LILI_MEM_HOST) return nullptr;
            const size_t stride = o->stride ? o->stride : sizeof(float4);
            return stride % 16 == 0 ? static_cast<char*>(lili_pinned_dev_ptr(o->data, 16)) : nullptr;
        };
        char* de = pinned16(edge); char* ds = pinned16(surf);
        if (de || ds) {
            hipLaunchKernelGGL(k_rot_send, dim3(16), dim3(256), 0, ctx->stream, st, R->edge_pts.as<float4>(), de, de ? (int)std::min(edge->capacity, (size_t)0x7fffffff) : 0,
                               de ? (int)(edge->stride ? edge->stride : sizeof(float4)) : 16, R->surf.as<float4>(), ds, ds ? (int)std::min(surf->capacity, (size_t)0x7fffffff) : 0,
                               ds ? (int)(surf->stride ? surf->stride : sizeof(float4)) : 16);
            HIPCHK(hipGetLastError());
            sent_edge = de != nullptr; sent_surf = ds != nullptr;
        }
        if (full_early) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->join_ev[1], 0));      // the read-back's synchronisation then also covers the side stream's copy
    }
    R->pend.on = true; R->pend.params = *params; R->pend.X = X; R->pend.full_early = full_early; R->pend.sent_edge = sent_edge; R->pend.sent_surf = sent_surf;
    for (int i = 0; i < 4; i++) { R->pend.q_imu[i] = q_imu[i]; R->pend.q_lb[i] = q_lb[i]; }
    R->pend.gen = ctx->readback_gen;
    return LILI_OK;
}

// `synced`: the context's stream has been synchronised since rot_enqueue (the state lies in the page-locked mirror already)
static int rot_complete(lili_ctx* ctx, lili_feature_out* full, lili_feature_out* edge, lili_feature_out* surf, bool synced) {
    auto* R = rot_of(ctx);
    if (!R->pend.on) return ctx->fail(LILI_E_STATE, "extract_rot: nothing enqueued");
    R->pend.on = false;
    const int n = R->n_in;
    const lili_rot_params* params = &R->pend.params;
    const double* q_imu = R->pend.q_imu; const double* q_lb = R->pend.q_lb;
    RotRingScratch X = R->pend.X;
    const bool full_early = R->pend.full_early;
    bool sent_edge = R->pend.sent_edge, sent_surf = R->pend.sent_surf;
    RotState* st = R->state.as<RotState>();
    int rc = LILI_OK;
    static const bool phases = getenv("LILI_ROT_PHASES") != nullptr;      // (the mirror's diagnostics of the last launch may be stale: the profiling tool reads the state itself)
    if (n > 0 && R->h_state_dev && !phases) {          // the state came with the ring kernel's last workgroup: wait, read it where it landed
        if (!synced) HIPCHK(hipStreamSynchronize(ctx->stream));
        std::memcpy(&R->host, R->h_state, sizeof(RotState));
        if (*reinterpret_cast<volatile int*>(R->h_state + 1)) R->host.fold_failed = 1;
    } else { int rb = lili_readback_add(ctx, &R->host, st, sizeof(RotState)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    if (n == 0) { R->host.first_valid = R->host.half_idx = 0x7fffffff; R->host.last_valid = -1; }
    if (n > 0 && (R->host.vox_overflow || R->host.fallback_rings > 0 || R->host.fold_failed)) sent_edge = sent_surf = false;      // the lists are about to change: copied again below
    if (n > 0 && R->host.fold_failed && !R->host.vox_overflow && R->host.fallback_rings == 0) {      // a ring gave up its look-back (never seen): the per-ring lists are complete, concatenate them
        hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                           R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                           R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>(), nullptr);
        HIPCHK(hipGetLastError());
        { int rb = lili_readback_add(ctx, &R->host, st, sizeof(RotState)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    }
    if (n > 0 && R->host.vox_overflow) {   // voxel coordinates beyond the packed keys: order by the radix pass, then the ring stage and the concatenation again
        RotDev P{};
        P.n_scans = params->n_scans; P.ds_rate = params->ds_rate; P.ds_v = params->ds_v; P.near_thres = params->near_range; P.atan_mode = ctx->rot_atan;
        for (int i = 0; i < 4; i++) { P.q_imu[i] = q_imu[i]; P.q_lb[i] = q_lb[i]; }
        hipLaunchKernelGGL(k_rot_voxel_order, dim3(kMaxRings), dim3(kRotBlock), sizeof(SortLds), ctx->stream, R->full.as<float4>(), P, st, X);
        hipLaunchKernelGGL(k_rot_ring, dim3(kMaxRings), dim3(kRotBlock), sizeof(RingLds), ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                           R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                           R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), X, RotFold{});
        hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                           R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                           R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>(), nullptr);
        HIPCHK(hipGetLastError());
        { int rb = lili_readback_add(ctx, &R->host, st, sizeof(RotState)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    }
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
        hipLaunchKernelGGL(k_rot_rank, dim3(kMaxRings, 6, 12), dim3(256), 0, ctx->stream, R->curv.as<float>(), P, st, R->sort_ind.as<int>());
        hipLaunchKernelGGL(k_rot_select_big, dim3(kMaxRings), dim3(kRotBlock), 0, ctx->stream, R->full.as<float4>(), R->curv.as<float>(), R->sort_ind.as<int>(), P, st,
                           R->label.as<int>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(), R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(),
                           R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), B);
        hipLaunchKernelGGL(k_rot_compact, dim3(kMaxRings), dim3(256), 0, ctx->stream, st, R->full.as<float4>(), R->ring_edge.as<int>(), R->ring_sharp.as<int>(),
                           R->ring_flat.as<int>(), R->lessflat_tmp.as<int>(), R->surf_tmp.as<float4>(), R->surf_cnt_tmp.as<int>(), R->edge_idx.as<int>(),
                           R->edge_pts.as<float4>(), R->sharp_idx.as<int>(), R->flat_idx.as<int>(), R->lessflat_idx.as<int>(), R->surf.as<float4>(), R->surf_cnt.as<int>(), nullptr);
        HIPCHK(hipGetLastError());
        { int rb = lili_readback_add(ctx, &R->host, st, sizeof(RotState)); if (rb == LILI_OK) rb = lili_readback_finish(ctx); if (rb != LILI_OK) return rb; }
    }
    R->have = true;
    R->redone = n > 0 && (R->host.vox_overflow || R->host.fallback_rings > 0 || R->host.fold_failed);
    R->prev_ok = n > 0; R->prev_edge = R->host.n_edge; R->prev_surf = R->host.n_surf;
    bool more = false;      // copies enqueued after the state's read-back: a second synchronisation
    if (full) {
        full->count = (size_t)R->host.n_full;
        if (!full_early && full->data && full->count && full->capacity) { rc = copy_out_f4(ctx, full, R->full.as<float4>(), full->count); if (rc) return rc; more = true; }
    }
    if (edge) { edge->count = (size_t)R->host.n_edge; if (!sent_edge && edge->data && edge->count && edge->capacity) { rc = copy_out_f4(ctx, edge, R->edge_pts.as<float4>(), edge->count); if (rc) return rc; more = true; } }
    if (surf) { surf->count = (size_t)R->host.n_surf; if (!sent_surf && surf->data && surf->count && surf->capacity) { rc = copy_out_f4(ctx, surf, R->surf.as<float4>(), surf->count); if (rc) return rc; more = true; } }
    if (more) HIPCHK(hipStreamSynchronize(ctx->stream));
    return LILI_OK;
}


int lili_extract_rot(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params,
                     lili_feature_out* full, lili_feature_out* edge, lili_feature_out* surf) {
    if (!ctx) return LILI_E_ARG;
    // Whatever way this call ends, no DMA into the caller's buffer may outlive it: every return between the early copy and its join —
    // a HIP error, a failed read-back, the second passes — drains the side stream first.
    struct DrainSide { hipStream_t s = nullptr; ~DrainSide() { if (s) (void)hipStreamSynchronize(s); } } drain_side;
    int rc = rot_enqueue(ctx, scan, q_imu, q_lb, params, full, edge, surf, &drain_side.s);
    if (rc != LILI_OK) return rc;
    rc = rot_complete(ctx, full, edge, surf, false);
    if (rc == LILI_OK) drain_side.s = nullptr;          // joined into the context's stream before the state's read-back and drained with it
    return rc;
}
}  // extern "C"
// lili_pipeline.hip: the extraction without outputs and without its synchronisation; the counts afterwards (no wait of its own if a read-back has synchronised the
// context's stream in between)
int lili_extract_rot_enqueue(lili_ctx* ctx, const lili_cloud* scan, const double q_imu[4], const double q_lb[4], const lili_rot_params* params, const lili_query_sink* sink) {
    if (!ctx) return LILI_E_ARG;
    if (sink && !(ctx->rot_fold && scan && scan->n > 0)) return ctx->fail(LILI_E_STATE, "extract_rot_enqueue: a query sink needs the ring stage that writes the scan's lists (rot_fold) and a non-empty scan");
    hipStream_t side = nullptr;
    return rot_enqueue(ctx, scan, q_imu, q_lb, params, nullptr, nullptr, nullptr, &side, sink);
}
void lili_extract_rot_prev(lili_ctx* ctx, int* prev_edge, int* prev_surf) {
    auto* R = rot_of(ctx);
    *prev_edge = R->prev_ok ? R->prev_edge : 0; *prev_surf = R->prev_ok ? R->prev_surf : 0;
}
int lili_extract_rot_complete(lili_ctx* ctx) {
    if (!ctx) return LILI_E_ARG;
    auto* R = rot_of(ctx);
    return rot_complete(ctx, nullptr, nullptr, nullptr, R->pend.on && ctx->readback_gen != R->pend.gen);
}
bool lili_extract_rot_redone(lili_ctx* ctx) { auto* R = rot_of(ctx); return R->redone; }
extern "C" {

// Intermediate products of the last lili_extract_rot (parity tests / debugging).  Any pointer may be NULL.
int lili_extract_rot_debug(lili_ctx* ctx, int32_t counts[8], int32_t* ring_start, int32_t* ring_end, int32_t* full_src, float* curvature, int32_t* label,
                           int32_t* edge_idx, int32_t* sharp_idx, int32_t* flat_idx, int32_t* lessflat_idx, int32_t* surf_cnt) {
    if (!ctx) return LILI_E_ARG;
    auto* R = rot_of(ctx);
    if (!R->have) return ctx->fail(LILI_E_STATE, "extract_rot_debug: run lili_extract_rot first");
    HIPCHK(hipSetDevice(ctx->device));
    const RotState& h = R->host;
    if (getenv("LILI_ROT_PHASES")) { fprintf(stderr, "ring 0, 100 MHz ticks: segment 0 (curvature + rank sort | greedy)"); fprintf(stderr, " %lld %lld; voxel ordering (binning | total) %lld %lld", h.tphase[1] - h.tphase[0], h.tphase[2] - h.tphase[1], h.tphase[3] >> 32, h.tphase[3] & 0xffffffff);
                                    fprintf(stderr, "; ring (join | less-flat | centroids)"); for (int k = 5; k < 8; k++) fprintf(stderr, " %lld", h.tphase[k] - h.tphase[k-1]);
                                    fprintf(stderr, "\n  from segment 0's start to k_rot_ring's start: %lld", h.tphase[4] - h.tphase[0]);
                                    fprintf(stderr, "\n  segment 0 of ring 0 (points in LDS | gap bits + curvature | rank sort): %lld %lld %lld", h.tseg[0] - h.tphase[0], h.tseg[1] - h.tseg[0], h.tphase[1] - h.tseg[1]);
                                    fprintf(stderr, "\n  segments redone in k_rot_ring: %d; join of the probe ring (working set | border check | redo | pick lists): %lld %lld %lld %lld", h.redo_segments, h.tjoin[0] - h.tphase[4], h.tjoin[1] - h.tjoin[0], h.tjoin[2] - h.tjoin[1], h.tphase[5] - h.tjoin[2]);
                                    fprintf(stderr, "\n  k_rot_ring ticks per ring:"); for (int k = 0; k < kMaxRings; k++) if (h.ring_ticks[k]) fprintf(stderr, " %d", h.ring_ticks[k]);
                                    fprintf(stderr, "\n  ordering ticks per ring:"); for (int k = 0; k < kMaxRings; k++) if (h.order_ticks[k]) fprintf(stderr, " %d", h.order_ticks[k]);
                                    fprintf(stderr, "\n  slowest segment per ring:"); for (int k = 0; k < kMaxRings; k++) if (h.seg_ticks[k]) fprintf(stderr, " %d", h.seg_ticks[k]); fprintf(stderr, "\n"); }
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
