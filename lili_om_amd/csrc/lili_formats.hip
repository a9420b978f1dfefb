// Callers and data formats either side of the hot path:
//   lili_livox_custom_to_cloud  Livox CustomMsg points -> pcl::PointXYZINormal records (device kernel)
//   lili_imu_integrate          gyro integration over one scan (host; its result is a kernel argument)
//   lili_marg_add_lidar         Gram record -> MarginalizationInfo's A, b blocks (host)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "lili_ctx.h"

namespace lili {

// One thread per CustomPoint (19 serialised bytes, unaligned): L/src/FormatConvert.cpp:13-23.
//   float s = float(offset_time / (float)time_end);            uint32 -> float, float division
//   pt.intensity = line + s * 0.1;                              int + double, stored as float
//   pt.curvature = 0.1 * reflectivity;                          double, stored as float
// PointXYZINormal's constructor leaves data[3] = 1.0f and zeroes the normal and the padding (pcl/point_types.hpp).
__global__ void k_custom_to_pcl(const unsigned char* __restrict__ in, int n, int stride, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto rd32 = [](const unsigned char* p) { return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24); };
    const unsigned time_end = rd32(in + (size_t)(n - 1) * stride);   // points.back().offset_time
    const unsigned char* p = in + (size_t)i * stride;
    const unsigned ot = rd32(p);
    const float x = __uint_as_float(rd32(p + 4)), y = __uint_as_float(rd32(p + 8)), z = __uint_as_float(rd32(p + 12));
    const unsigned refl = p[16], line = p[18];
    const float s = (float)ot / (float)time_end;
    const float intensity = (float)((double)(int)line + (double)s * 0.1);
    const float curvature = (float)(0.1 * (double)(int)refl);
    float4* o = reinterpret_cast<float4*>(out + (size_t)i * 12);
    o[0] = make_float4(x, y, z, 1.0f);
    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    o[2] = make_float4(intensity, curvature, 0.f, 0.f);
}

}  // namespace lili
using namespace lili;

extern "C" {

int lili_livox_custom_to_cloud(lili_ctx* ctx, const void* points, size_t point_num, size_t stride, int in_mem, void* out, int out_mem) {
    if (!ctx) return LILI_E_ARG;
    ARGCHK(point_num == 0 || (points && out), "custom_to_cloud: null buffer");
    ARGCHK(stride >= 19, "custom_to_cloud: a CustomPoint has 19 serialised bytes");
    ARGCHK(point_num < ((size_t)1 << 31) / 48, "custom_to_cloud: too many points");
    ARGCHK((in_mem == LILI_MEM_HOST || in_mem == LILI_MEM_DEVICE) && (out_mem == LILI_MEM_HOST || out_mem == LILI_MEM_DEVICE), "custom_to_cloud: bad mem");
    if (point_num == 0) return LILI_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const unsigned char* src = reinterpret_cast<const unsigned char*>(points);
    if (in_mem == LILI_MEM_HOST) {
        HIPCHK(ctx->staging.ensure(point_num * stride));
        HIPCHK(hipMemcpyAsync(ctx->staging.p, points, point_num * stride, hipMemcpyHostToDevice, ctx->stream));
        src = ctx->staging.as<unsigned char>();
    }
    float* dst = reinterpret_cast<float*>(out);
    if (out_mem == LILI_MEM_HOST) {
        HIPCHK(ctx->fmt_out.ensure(point_num * 48));
        dst = ctx->fmt_out.as<float>();
    } else ARGCHK((reinterpret_cast<uintptr_t>(out) & 15) == 0, "custom_to_cloud: device output must be 16-byte aligned");
    hipLaunchKernelGGL(k_custom_to_pcl, dim3((unsigned)((point_num + 255) / 256)), dim3(256), 0, ctx->stream, src, (int)point_num, (int)stride, dst);
    HIPCHK(hipGetLastError());
    if (out_mem == LILI_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(out, dst, point_num * 48, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else if (in_mem == LILI_MEM_HOST) HIPCHK(hipStreamSynchronize(ctx->stream));   // the staging copy read the caller's buffer
    return LILI_OK;
}

void lili_imu_reset(lili_imu_state* st) {
    if (!st) return;
    std::memset(st, 0, sizeof(*st));
    st->t_cur = -1.0;
}

// q <- q * deltaQ(theta), Eigen's quaternion product order, no normalisation (L:129-133; math_tools.h:125-138)
static void imu_solve_rotation(double q[4], double gyr0[3], double dt, const double w[3]) {
    double th[3];
    for (int k = 0; k < 3; k++) { double un = 0.5 * (gyr0[k] + w[k]); th[k] = un * dt; }
    const double bw = 1.0, bx = th[0] / 2.0, by = th[1] / 2.0, bz = th[2] / 2.0;
    const double aw = q[0], ax = q[1], ay = q[2], az = q[3];
    q[0] = aw * bw - ax * bx - ay * by - az * bz;
    q[1] = aw * bx + ax * bw + ay * bz - az * by;
    q[2] = aw * by + ay * bw + az * bx - ax * bz;
    q[3] = aw * bz + az * bw + ax * by - ay * bx;
    for (int k = 0; k < 3; k++) gyr0[k] = w[k];
}

int lili_imu_integrate(lili_imu_state* st, const double* stamps, const double* gyr, size_t n, double t_scan_next, double q_out[4]) {
    if (!st || !q_out || (n > 0 && (!stamps || !gyr))) return LILI_E_ARG;
    double q[4] = {1.0, 0.0, 0.0, 0.0};                      // q_iMU is reset to identity after every scan (L:403)
    if (st->idx == 0 && st->first == 0 && st->t_cur == 0.0) st->t_cur = -1.0;   // zero-initialised state
    if (n > 0) {                                              // `if (imu_buf.size() > 0) processIMU(time_scan_next)` (L:230-231)
        if (st->t_cur < 0) st->t_cur = stamps[0];             // imuHandler, L:178-179 (first message ever)
        if (!st->first) { st->first = 1; for (int k = 0; k < 3; k++) st->gyr0[k] = gyr[k]; }   // L:182-190
        double r[3] = {0.0, 0.0, 0.0};
        int64_t i = st->idx;
        if (i >= (int64_t)n) i--;                             // L:138-139
        while (stamps[i] < t_scan_next) {                     // L:140-154
            double t = stamps[i];
            if (st->t_cur < 0) st->t_cur = t;
            double dt = t - st->t_cur;
            st->t_cur = stamps[i];
            for (int k = 0; k < 3; k++) r[k] = gyr[3 * i + k];
            imu_solve_rotation(q, st->gyr0, dt, r);
            i++;
            if (i >= (int64_t)n) break;
        }
        if (i < (int64_t)n) {                                 // L:156-167: interpolate the rate at the scan boundary
            double dt1 = t_scan_next - st->t_cur;
            double dt2 = stamps[i] - t_scan_next;
            double w1 = dt2 / (dt1 + dt2);
            double w2 = dt1 / (dt1 + dt2);
            for (int k = 0; k < 3; k++) r[k] = w1 * r[k] + w2 * gyr[3 * i + k];
            imu_solve_rotation(q, st->gyr0, dt1, r);
        }
        st->t_cur = t_scan_next;                              // L:169-170
        st->idx = i;
    }
    if (std::isnan(q[0]) || std::isnan(q[1]) || std::isnan(q[2]) || std::isnan(q[3])) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }   // L:232-234
    for (int k = 0; k < 4; k++) q_out[k] = q[k];
    return LILI_OK;
}

int lili_marg_add_lidar(const double gram[64], double* A, size_t ld, double* b, size_t pos, size_t idx_t, size_t idx_q) {
    if (!gram || !A || !b) return LILI_E_ARG;
    if (ld < pos || idx_t + 3 > pos || idx_q + 3 > pos) return LILI_E_ARG;
    const int g[6] = {0, 1, 2, 4, 5, 6};                       // t, then rightCols(3) of the (w,x,y,z) quaternion Jacobian
    const size_t at[6] = {idx_t, idx_t + 1, idx_t + 2, idx_q, idx_q + 1, idx_q + 2};
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) A[at[i] * ld + at[j]] += gram[g[i] * 8 + g[j]];
        b[at[i]] += gram[g[i] * 8 + 7];
    }
    return LILI_OK;
}

}  // extern "C"
