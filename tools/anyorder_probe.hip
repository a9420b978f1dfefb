// Does hipExtAnyOrderLaunch drop the barrier between two kernels of ONE stream on this box (hip_ext.h says "not supported on GFX9xx" for the module-launch variant)?
// And what does a granule hand-off between two kernels that run concurrently cost (option "overlap_gn" of liblili_hip: the reduction + GN kernel publishes the
// pose, the association launched behind it without a barrier polls for it)?
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/_probe/anyorder_probe && tools/_probe/anyorder_probe
// Kernel A (1 x 64): stamps its start, spins `spin_us`, publishes a keyed 16-byte granule, stamps its end.  Kernel B (3125 x 64): every wave stamps its start, polls the
// granule, stamps the moment it saw it.  B is launched (a) the plain way, (b) with hipExtAnyOrderLaunch.  Printed per case: B's first / median / last start relative to
// A's start and to A's end, and the granule latency (A's publish stamp -> B's observation: min / median / max over the waves).  s_memrealtime = 100 MHz, chip-wide.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__global__ void k_a(long long* st, double* granule, unsigned long long key, int spin_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = now();
    st[0] = t0;
    while (now() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(4);
    const unsigned long long lo = 0x3ff0000000000000ull, hi = lo ^ key;
    const u32x4 d = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    st[1] = now();
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(granule), "v"(d) : "memory");
    st[2] = now();
}
__global__ void k_b(long long* start, long long* seen, const double* granule, unsigned long long key) {
    const long long t0 = now();
    bool ok = false;
    long long t1 = 0;
    for (unsigned sweep = 0; sweep < (1u << 20) && !ok; sweep++) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(granule);
        const unsigned long long lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (lo ^ hi) == key;
        t1 = now();
        if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; seen[blockIdx.x] = ok ? t1 : -1; }
}

int main() {
    const int nb = 3125;
    long long *d_st, *d_start, *d_seen; double* d_gr;
    hipMalloc(&d_st, 64); hipMalloc(&d_start, nb * 8); hipMalloc(&d_seen, nb * 8); hipMalloc(&d_gr, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<long long> st(3), start(nb), seen(nb);
    for (int mode = 0; mode < 2; mode++) for (int spin_us : {5, 20}) for (int rep = 0; rep < 3; rep++) {
        const unsigned long long key = 0x9E3779B97F4A7C15ull * (unsigned long long)(1 + rep + 10 * mode + 100 * spin_us);
        hipMemsetAsync(d_gr, 0, 64, s);
        hipStreamSynchronize(s);
        hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, d_st, d_gr, key, spin_us * 100);
        if (mode) hipExtLaunchKernelGGL(k_b, dim3(nb), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d_start, d_seen, (const double*)d_gr, key);
        else hipLaunchKernelGGL(k_b, dim3(nb), dim3(64), 0, s, d_start, d_seen, (const double*)d_gr, key);
        hipError_t e = hipStreamSynchronize(s);
        hipMemcpy(st.data(), d_st, 24, hipMemcpyDeviceToHost); hipMemcpy(start.data(), d_start, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(seen.data(), d_seen, nb * 8, hipMemcpyDeviceToHost);
        std::vector<long long> a = start, b;
        int lost = 0;
        for (int i = 0; i < nb; i++) { if (seen[i] < 0) lost++; else b.push_back(seen[i] - st[1]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        auto us = [](long long t) { return (double)t * 0.01; };
        std::printf("%s spin %2d us rep %d (%s): B starts %+7.2f / %+7.2f / %+7.2f us after A's start (first / median / last), A ran %.2f us; first B start is %+.2f us vs A's END; "
                    "granule seen %.2f / %.2f / %.2f us after its store was issued (min / median / max), %d waves gave up\n",
                    mode ? "any-order" : "plain    ", spin_us, rep, hipGetErrorString(e), us(a.front() - st[0]), us(a[nb / 2] - st[0]), us(a.back() - st[0]), us(st[2] - st[0]), us(a.front() - st[2]),
                    b.empty() ? -1.0 : us(b.front()), b.empty() ? -1.0 : us(b[b.size() / 2]), b.empty() ? -1.0 : us(b.back()), lost);
    }
    return 0;
}
