// tools/launch_floor.hip — where do the microseconds between small dependent launches go?  
//
// Every case enqueues N dependent launches on ONE stream and reports
//   wall     : (host time from first enqueue to stream drained) / N
//   events   : HIP events around the whole batch / N
//   in-kernel: s_memrealtime (100 MHz) stamped by lane 0 at the first and last instruction of every launch ->
//              kernel body time, and the GAP from the end of launch i to the start of launch i + 1 (median / p90).
// No profiler involved: rocprofv3 --kernel-trace gives every dispatch its own completion signal, which is itself a
// suspect for the 4.4 us "duration" of an 8-lane kernel in profiles/r03_kernel_stats.csv.
//
// Cases: (a) fresh non-blocking stream, trivial 1 x 64 kernel; (a2) 256 x 256 trivial; (b) null stream; (c) the library's own
// k_pose_copy through lili_s2m_pose_copy on a context's stream; (c2) the same after lili_p2p-style allocations (uncached /
// fine-grained device memory + mapped host memory) exist in the process; (d) trivial kernel behind a kernel that dirtied
// 8 MB; (d2) the same with non-temporal stores; (e) 200 trivial launches captured in a hipGraph; (f) high-priority stream;
// (g) a kernel with a 400-byte by-value argument block (k_associate's size).
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_probe/launch_floor tools/launch_floor.hip -I include -L lili_om_amd -llili_hip -Wl,-rpath,$PWD/lili_om_amd
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lili_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Big { double v[50]; };     // 400 bytes of by-value kernel arguments

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

__global__ void k_trivial(unsigned long long* stamps, int i) {
    unsigned long long t0 = now();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * i] = t0;
        stamps[2 * i + 1] = now();
    }
}
__global__ void k_trivial_big(unsigned long long* stamps, int i, Big b) {
    unsigned long long t0 = now();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * i] = t0 + (b.v[49] == 123.25 ? 1 : 0);
        stamps[2 * i + 1] = now();
    }
}
template <bool NT> __global__ void k_dirty(float4* buf, size_t n, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 x = {v, v, v, v};
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<v4*>(&buf[i]));
        else *reinterpret_cast<v4*>(&buf[i]) = x;
    }
}

static void stats(const char* name, int n, double wall_us, float ev_ms, const std::vector<unsigned long long>& st, int stride = 1) {
    std::vector<double> body, gap;
    for (int i = 0; i < n; i++) body.push_back((st[2 * i * stride + 1] - st[2 * i * stride]) * 0.01);
    for (int i = 1; i < n; i++) gap.push_back((double)((long long)st[2 * i * stride] - (long long)st[2 * (i - 1) * stride + 1]) * 0.01);
    std::sort(body.begin(), body.end()); std::sort(gap.begin(), gap.end());
    auto q = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[(size_t)(f * (v.size() - 1))]; };
    std::printf("%-58s wall %6.2f us  events %6.2f us  | body med %5.2f  gap end->start: med %5.2f  p10 %5.2f  p90 %5.2f us\n", name, wall_us / n, ev_ms * 1e3 / n,
                q(body, 0.5), q(gap, 0.5), q(gap, 0.1), q(gap, 0.9));
}

template <class F> static void run_case(const char* name, hipStream_t s, int n, unsigned long long* d_st, F launch, int stride = 1) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; i++) launch(i % n);          // warm-up (code object load, queue creation)
    CK(hipStreamSynchronize(s));
    CK(hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * 2 * n * stride, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; i++) launch(i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * n * stride);
    CK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    stats(name, n, wall, ms, st, stride);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, sizeof(unsigned long long) * 2 * N * 2));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    run_case("(a) fresh non-blocking stream, trivial 1 x 64", s, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_st, i); });
    run_case("(a2) same stream, trivial 256 x 256", s, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, s, d_st, i); });
    run_case("(a3) same stream, trivial 3125 x 64", s, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(3125), dim3(64), 0, s, d_st, i); });
    run_case("(b) null stream, trivial 1 x 64", nullptr, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, nullptr, d_st, i); });
    Big big{}; big.v[49] = 1.0;
    run_case("(g) 400-byte by-value argument block, 1 x 64", s, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial_big, dim3(1), dim3(64), 0, s, d_st, i, big); });
    {
        hipStream_t hp; int lo, hi;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&hp, hipStreamNonBlocking, hi));
        run_case("(f) high-priority stream, trivial 1 x 64", hp, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, hp, d_st, i); });
        CK(hipStreamDestroy(hp));
    }
    {   // (d) trivial behind a kernel that dirtied 8 MB: stamps of the trivial launches only (stride 1 over the trivial ones)
        float4* buf; size_t n4 = (8u << 20) / 16;
        CK(hipMalloc(&buf, n4 * 16));
        run_case("(d) [8 MB plain stores] -> trivial: pair", s, N / 2, d_st, [&](int i) {
            hipLaunchKernelGGL(k_dirty<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, buf, n4, (float)i);
            hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_st, i); });
        run_case("(d2) [8 MB non-temporal stores] -> trivial: pair", s, N / 2, d_st, [&](int i) {
            hipLaunchKernelGGL(k_dirty<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, buf, n4, (float)i);
            hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_st, i); });
        run_case("(d3) [8 MB plain stores] alone (for the subtraction)", s, N / 2, d_st, [&](int i) {
            hipLaunchKernelGGL(k_dirty<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, buf, n4, (float)i); });
        CK(hipFree(buf));
    }
    {   // (e) 200 trivial launches captured in a graph, replayed N / 200 times
        const int G = 200;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < G; i++) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_st, i);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        int reps = std::max(1, N / G);
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * G);
        CK(hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
        stats("(e) hipGraph of 200 trivial 1 x 64 launches (last replay's stamps)", G, wall / reps, ms / reps, st);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    {   // (c) the library's k_pose_copy on a context's own stream and on a caller-provided stream; no stamps inside (product kernel): wall / events only
        lili_ctx* ctx = nullptr;
        if (lili_ctx_create(&ctx, 0, nullptr) != LILI_OK) { std::fprintf(stderr, "lili_ctx_create failed\n"); return 1; }
        double t[3] = {0.1, 0.2, 0.3}, q[4] = {1, 0, 0, 0};
        lili_s2m_pose_set(ctx, 1, t, q);
        auto time_lib = [&](const char* name, lili_ctx* c) {
            for (int i = 0; i < 50; i++) lili_s2m_pose_copy(c, 0, 1);
            lili_sync(c);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) lili_s2m_pose_copy(c, 0, 1);
            double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            lili_sync(c);
            double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%-58s wall %6.2f us  (host enqueue alone %5.2f us per call)\n", name, wall / N, enq / N);
        };
        time_lib("(c) lili_s2m_pose_copy, context's own stream", ctx);
        // (c2) allocations of the kinds lili_p2p makes: uncached device memory + mapped host memory
        void* unc = nullptr; void* hm = nullptr;
        hipError_t e1 = hipExtMallocWithFlags(&unc, 1 << 16, hipDeviceMallocUncached);
        hipError_t e2 = hipHostMalloc(&hm, 1 << 16, hipHostMallocMapped);
        std::printf("    uncached device allocation: %s, mapped host allocation: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        time_lib("(c2) the same with uncached + mapped-host allocations alive", ctx);
        run_case("(a') fresh stream again, trivial 1 x 64 (after c2's allocations)", s, N, d_st, [&](int i) { hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, d_st, i); });
        if (unc) CK(hipFree(unc));
        if (hm) CK(hipHostFree(hm));
        lili_ctx_destroy(ctx);
        lili_ctx* c2 = nullptr;
        if (lili_ctx_create(&c2, 0, (void*)s) == LILI_OK) {
            lili_s2m_pose_set(c2, 1, t, q);
            time_lib("(c3) lili_s2m_pose_copy, caller's non-blocking stream", c2);
            lili_ctx_destroy(c2);
        }
    }
    CK(hipStreamDestroy(s));
    CK(hipFree(d_st));
    return 0;
}
