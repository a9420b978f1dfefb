// tools/randrun_probe.hip — what does the MEMORY side of the dense-map association cost, with no selection arithmetic at all?  
//
// k_associate_fine on configs[2] variant B: 200 000 queries, each reads two range words and then ONE contiguous run of ~32 float4 (512 B) at an
// unrelated place of a 1.1 GB array (the queries are ~8 fine cells apart: no two of them share a line).  This probe issues exactly that pattern —
// N "queries", each a run of RUN bytes at a 16-byte aligned offset of a buffer far larger than L2 + Infinity Cache — and nothing else:
//   lane : one lane per query, 4 independent 16-byte loads per trip (the kernel's walk), two trips in flight
//   quad : 4 lanes per query, lane k loads float4 k of every 64-byte chunk (one line look-up per query and trip instead of four)
//   x16  : 16 lanes per query, 256 bytes per trip
// each in the order given (random) and with the offsets sorted by address (what a spatially sorted query list would give).
// Prints microseconds per launch (HIP events, median of 9) and GB/s of useful bytes.  Under rocprofv3 --pmc FETCH_SIZE / TCC_MISS_sum the same
// binary calibrates bytes per L2 miss for this pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_probe/randrun_probe tools/randrun_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ float eat(float4 p) { return p.x + p.y + p.z + p.w; }

__global__ __launch_bounds__(64) void k_lane(const float4* __restrict__ buf, const unsigned* __restrict__ off, int n, int chunks, float* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float4* p = buf + off[i];
    float s = 0.f;
    float4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    for (int c = 1; c <= chunks; c++) {
        const float4* q = p + 4 * c;        // (the kernel also requests one chunk past the end)
        float4 b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
        s += eat(a0) + eat(a1) + eat(a2) + eat(a3);
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    }
    if (s == 12345.678f) out[i] = s;
}
template <int L>
__global__ __launch_bounds__(64) void k_coop(const float4* __restrict__ buf, const unsigned* __restrict__ off, int n, int run_f4, float* __restrict__ out) {
    const int qi = (blockIdx.x * 64 + threadIdx.x) / L, l = threadIdx.x % L;
    if (qi >= n) return;
    const float4* p = buf + off[qi] + l;
    float s = 0.f;
    float4 a = p[0];
    for (int c = L; c < run_f4 + L; c += L) {
        float4 b = p[c];
        s += eat(a);
        a = b;
    }
    if (s == 12345.678f) out[qi] = s;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 200000;
    const int run_f4 = argc > 2 ? std::atoi(argv[2]) : 32;
    const size_t buf_f4 = (size_t)(argc > 3 ? std::atoi(argv[3]) : 1100) * (1u << 20) / 16;
    float4* buf; unsigned* off; float* out;
    CK(hipMalloc(&buf, buf_f4 * 16)); CK(hipMalloc(&off, (size_t)n * 4)); CK(hipMalloc(&out, (size_t)n * 4));
    CK(hipMemset(buf, 0, buf_f4 * 16));
    std::mt19937_64 rng(12345);
    std::vector<unsigned> h(n);
    for (auto& v : h) v = (unsigned)(rng() % (buf_f4 - (size_t)run_f4 - 64));
    std::vector<unsigned> hs = h;
    std::sort(hs.begin(), hs.end());
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        std::vector<float> t;
        for (int r = 0; r < 11; r++) {
            CK(hipEventRecord(e0, st)); launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        const double us = t[t.size() / 2];
        std::printf("%-28s %8.2f us   %7.1f GB/s useful (%d x %d B)\n", name, us, (double)n * run_f4 * 16 / us * 1e-3, n, run_f4 * 16);
    };
    for (int sorted = 0; sorted < 2; sorted++) {
        CK(hipMemcpy(off, sorted ? hs.data() : h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        const char* tag = sorted ? "sorted" : "random";
        char name[64];
        std::snprintf(name, sizeof name, "lane  %s", tag);
        timeit(name, [&] { hipLaunchKernelGGL(k_lane, dim3((n + 63) / 64), dim3(64), 0, st, buf, off, n, run_f4 / 4, out); });
        std::snprintf(name, sizeof name, "quad  %s", tag);
        timeit(name, [&] { hipLaunchKernelGGL(k_coop<4>, dim3((n * 4 + 63) / 64), dim3(64), 0, st, buf, off, n, run_f4, out); });
        std::snprintf(name, sizeof name, "x16   %s", tag);
        timeit(name, [&] { hipLaunchKernelGGL(k_coop<16>, dim3((n * 16 + 63) / 64), dim3(64), 0, st, buf, off, n, run_f4, out); });
    }
    return 0;
}
