// What does it cost to ask the runtime whether a host pointer is page-locked?  (the Livox extractor's direct paths must not trust a cached answer: a freed
// page-locked buffer's address can come back as pageable memory)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_probe/ptr_attr_cost tools/ptr_attr_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main() {
    void* pin = nullptr; hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault);
    void* page = std::malloc(1 << 20);
    void* dev = nullptr; hipMalloc(&dev, 1 << 20);
    auto time = [&](const char* name, void* p, int mode) {
        auto t0 = std::chrono::steady_clock::now();
        int ok = 0;
        for (int i = 0; i < 2000; i++) {
            if (mode == 0) { hipPointerAttribute_t a{}; if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost) ok++; else (void)hipGetLastError(); }
            else { void* d = nullptr; if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) ok++; else (void)hipGetLastError(); }
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        std::printf("%-44s %8.2f us per call (%d / 2000 say page-locked)\n", name, us, ok);
    };
    time("hipPointerGetAttributes, page-locked", pin, 0);
    time("hipPointerGetAttributes, pageable", page, 0);
    time("hipPointerGetAttributes, device", dev, 0);
    time("hipHostGetDevicePointer, page-locked", pin, 1);
    time("hipHostGetDevicePointer, pageable", page, 1);
    char* mid = static_cast<char*>(pin) + 4096;
    time("hipPointerGetAttributes, inside page-locked", mid, 0);
    return 0;
}
