// Probe (run through gpurun): operand / result lane maps and issue cost of the two f64 MFMA forms on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o tools/_probe/probe_mfma && tools/_probe/probe_mfma
// For every pair (la, lb): A = 1 on lane la only, B = 1 on lane lb only -> which result lanes / registers become 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void k_map4(int* out /*[64*64]: out lane or -1*/) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) | ((__popcll(m) - 1) << 8) : -1;
        }
}
__global__ void k_map16(int* out /*[64*64]: lane | reg << 8 or -1*/) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++)
        for (int lb = 0; lb < 64; lb++) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            v4f64 c = {0, 0, 0, 0};
            v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            int found = -1;
            for (int r = 0; r < 4; r++) {
                unsigned long long m = __ballot(d[r] != 0.0);
                if (m) found = (__ffsll((long long)m) - 1) | (r << 8);
            }
            if (lane == 0) out[la * 64 + lb] = found;
        }
}
template <int KIND, int DEP>
__global__ void k_time(long long* out, double* sink, int n) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        if (KIND == 16) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            if (DEP) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); }
            else { c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); }
        } else {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            if (DEP) { s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); }
            else { s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0); s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0); }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3;
}
// plain f64 FMA issue cost for comparison: 4 independent chains
__global__ void k_fma(long long* out, double* sink, int n) {
    double a = 1.0 + threadIdx.x * 1e-9, x0 = 1, x1 = 2, x2 = 3, x3 = 4;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) { x0 = __fma_rn(x0, a, 1e-9); x1 = __fma_rn(x1, a, 1e-9); x2 = __fma_rn(x2, a, 1e-9); x3 = __fma_rn(x3, a, 1e-9); }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

int main() {
    int* d; hipMalloc(&d, 64 * 64 * sizeof(int));
    std::vector<int> h(64 * 64);
    hipLaunchKernelGGL(k_map4, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    printf("4x4x4_4b: for A lane la (rows) the B lanes that pair with it and the result lane\n");
    for (int la = 0; la < 64; la++) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d%s", lb, h[la * 64 + lb] & 255, (h[la * 64 + lb] >> 8) ? "+" : "");
        printf("\n");
    }
    hipLaunchKernelGGL(k_map16, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    printf("16x16x4: A lanes 0,1,16,17 only\n");
    for (int la : {0, 1, 16, 17}) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf(" B%02d->D%02d.r%d", lb, h[la * 64 + lb] & 255, h[la * 64 + lb] >> 8);
        printf("\n");
    }
    long long* dt; double* sink; hipMalloc(&dt, 8); hipMalloc(&sink, 8 * 1024 * 1024);
    const int n = 2000;
    auto run = [&](const char* name, auto kern, int blocks, int threads, double per) {
        long long t = 0;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dt, sink, n);
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf("%-44s %8.1f clock64 ticks per instruction (one wave's view)\n", name, (double)t / (n * per));
    };
    run("mfma_f64_16x16x4 dependent, 1 wave/SIMD", k_time<16, 1>, 1, 64, 4);
    run("mfma_f64_16x16x4 independent x4, 1 wave/SIMD", k_time<16, 0>, 1, 64, 4);
    run("mfma_f64_16x16x4 independent, 4 waves/SIMD", k_time<16, 0>, 1, 1024, 4);
    run("mfma_f64_4x4x4 dependent, 1 wave/SIMD", k_time<4, 1>, 1, 64, 4);
    run("mfma_f64_4x4x4 independent x4, 1 wave/SIMD", k_time<4, 0>, 1, 64, 4);
    run("mfma_f64_4x4x4 independent, 4 waves/SIMD", k_time<4, 0>, 1, 1024, 4);
    run("v_fma_f64 4 chains, 1 wave/SIMD", k_fma, 1, 64, 4);
    run("v_fma_f64 4 chains, 4 waves/SIMD", k_fma, 1, 1024, 4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
    int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, shader clock %d kHz\n", clk, sclk);
    return 0;
}
