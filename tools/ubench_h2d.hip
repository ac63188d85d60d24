// tools/ubench_h2d.hip -- how fast does a pageable / registered host buffer reach HBM, on which stream?
// hipcc --offload-arch=gfx950 -O2 tools/ubench_h2d.hip -o tools/bin/ubench_h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 512ull << 20;
    float *h = (float *)malloc(bytes);
    memset(h, 1, bytes);
    float *d; CK(hipMalloc(&d, bytes));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1)); double t1 = now(); CK(hipStreamSynchronize(s1)); double t2 = now();
        printf("pageable, stream s1     : call %.2f ms, total %.2f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, bytes / (t2 - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, 0)); double t1 = now(); CK(hipStreamSynchronize(0)); double t2 = now();
        printf("pageable, null stream   : call %.2f ms, total %.2f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, (t2 - t0) * 1e3, bytes / (t2 - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int p = 0; p < 8; ++p) CK(hipMemcpyAsync((char *)d + p * (bytes / 8), (char *)h + p * (bytes / 8), bytes / 8, hipMemcpyHostToDevice, s2));
        CK(hipStreamSynchronize(s2)); double t2 = now();
        printf("pageable, 8 panels on s2: total %.2f ms  (%.1f GB/s)\n", (t2 - t0) * 1e3, bytes / (t2 - t0) / 1e9);
    }
    {
        double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double t1 = now();
        printf("hipHostRegister 512 MiB : %.2f ms\n", (t1 - t0) * 1e3);
        for (int rep = 0; rep < 3; ++rep) {
            double a = now(); CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1)); double b = now(); CK(hipStreamSynchronize(s1)); double c = now();
            printf("registered, stream s1   : call %.2f ms, total %.2f ms  (%.1f GB/s)\n", (b - a) * 1e3, (c - a) * 1e3, bytes / (c - a) / 1e9);
        }
        t0 = now(); CK(hipHostUnregister(h)); t1 = now();
        printf("hipHostUnregister       : %.2f ms\n", (t1 - t0) * 1e3);
    }
    return 0;
}
