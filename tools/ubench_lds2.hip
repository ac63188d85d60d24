// tools/ubench_lds2.hip -- the guide's LDS read rates, reproduced or refuted.
// Bare streams of ds_read_b128 / ds_read_b64 / ds_read_b32 written in inline assembly: ONE address register, 16 reads at immediate offsets per
// s_waitcnt lgkmcnt(0), NO VALU between the reads (tools/ubench_lds.hip's legs all carried address arithmetic or an add per read).  Linear addresses
// (lane i reads bytes [W i, W i + W) of a window: every bank once per lane group = conflict-free for each width).  Timed with BOTH clocks:
// s_memtime (shader cycles) -> bytes per clock per CU, s_memrealtime (100 MHz) -> the shader clock the part really ran at, and TB/s chip-wide.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds2.hip -o tools/bin/ubench_lds2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ inline unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

// W = bytes per lane (16 / 8 / 4).  16 reads per wait; the window of one read is 64 W bytes, consecutive reads are 64 W apart (offsets fit 16 bits).
template <int W>
__global__ void k(unsigned *out, int iters, unsigned long long *clk) {
    extern __shared__ unsigned lds[];
    for (int e = threadIdx.x; e < 16 * 64 * W / 4 + 4096; e += blockDim.x) lds[e] = e;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned addr = lane * W + (threadIdx.x >> 6) * 64;      // waves start 64 B apart (same banks pattern, different rows)
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = memtime(), r0 = realtime();
    for (int it = 0; it < iters; ++it) {
        if (W == 16) {
            u32x4 a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
            asm volatile(
                "ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n"
                "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n"
                "ds_read_b128 %8, %16 offset:8192\n ds_read_b128 %9, %16 offset:9216\n ds_read_b128 %10, %16 offset:10240\n ds_read_b128 %11, %16 offset:11264\n"
                "ds_read_b128 %12, %16 offset:12288\n ds_read_b128 %13, %16 offset:13312\n ds_read_b128 %14, %16 offset:14336\n ds_read_b128 %15, %16 offset:15360\n"
                "s_waitcnt lgkmcnt(0)"
                : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7), "=v"(a8), "=v"(a9), "=v"(a10), "=v"(a11), "=v"(a12),
                  "=v"(a13), "=v"(a14), "=v"(a15)
                : "v"(addr));
            acc ^= a0.x ^ a15.w;
        } else if (W == 8) {
            u32x2 a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
            asm volatile(
                "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:512\n ds_read_b64 %2, %16 offset:1024\n ds_read_b64 %3, %16 offset:1536\n"
                "ds_read_b64 %4, %16 offset:2048\n ds_read_b64 %5, %16 offset:2560\n ds_read_b64 %6, %16 offset:3072\n ds_read_b64 %7, %16 offset:3584\n"
                "ds_read_b64 %8, %16 offset:4096\n ds_read_b64 %9, %16 offset:4608\n ds_read_b64 %10, %16 offset:5120\n ds_read_b64 %11, %16 offset:5632\n"
                "ds_read_b64 %12, %16 offset:6144\n ds_read_b64 %13, %16 offset:6656\n ds_read_b64 %14, %16 offset:7168\n ds_read_b64 %15, %16 offset:7680\n"
                "s_waitcnt lgkmcnt(0)"
                : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7), "=v"(a8), "=v"(a9), "=v"(a10), "=v"(a11), "=v"(a12),
                  "=v"(a13), "=v"(a14), "=v"(a15)
                : "v"(addr));
            acc ^= a0.x ^ a15.y;
        } else {
            unsigned a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
            asm volatile(
                "ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:256\n ds_read_b32 %2, %16 offset:512\n ds_read_b32 %3, %16 offset:768\n"
                "ds_read_b32 %4, %16 offset:1024\n ds_read_b32 %5, %16 offset:1280\n ds_read_b32 %6, %16 offset:1536\n ds_read_b32 %7, %16 offset:1792\n"
                "ds_read_b32 %8, %16 offset:2048\n ds_read_b32 %9, %16 offset:2304\n ds_read_b32 %10, %16 offset:2560\n ds_read_b32 %11, %16 offset:2816\n"
                "ds_read_b32 %12, %16 offset:3072\n ds_read_b32 %13, %16 offset:3328\n ds_read_b32 %14, %16 offset:3584\n ds_read_b32 %15, %16 offset:3840\n"
                "s_waitcnt lgkmcnt(0)"
                : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7), "=v"(a8), "=v"(a9), "=v"(a10), "=v"(a11), "=v"(a12),
                  "=v"(a13), "=v"(a14), "=v"(a15)
                : "v"(addr));
            acc ^= a0 ^ a15;
        }
    }
    const unsigned long long t1 = memtime(), r1 = realtime();
    __syncthreads();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int W>
static void run(int waves_per_cu, int blocks_per_cu, int iters) {
    const int threads = 64 * waves_per_cu / blocks_per_cu, blocks = 256 * blocks_per_cu;
    unsigned *out; unsigned long long *clk;
    CK(hipMalloc(&out, sizeof(unsigned) * (size_t)blocks * threads)); CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
    const size_t lds = 16 * 64 * W + 16384 + 4096;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<W><<<blocks, threads, lds>>>(out, iters, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<W><<<blocks, threads, lds>>>(out, iters, clk);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    cyc /= blocks; real /= blocks;
    const double bytes_block = (double)iters * 16 * 64 * W * (threads / 64);
    const double bpc_cu = bytes_block * blocks_per_cu / cyc;
    const double mhz = cyc / (real / 100.0);      // s_memrealtime ticks at 100 MHz
    printf("ds_read_b%-3d  %2d waves/CU in %d block(s)/CU: %7.1f B/clk/CU  (%5.1f %% of 256)  shader clock %6.0f MHz  %6.1f TB/s chip-wide in-kernel, %6.1f TB/s by HIP events\n",
           8 * W, waves_per_cu, blocks_per_cu, bpc_cu, 100.0 * bpc_cu / 256.0, mhz, bpc_cu * mhz * 1e6 * 256 / 1e12, bytes_block * blocks / (ms * 1e-3) / 1e12);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
    const int iters = 20000;
    for (int w : {4, 8, 16}) run<16>(w, 1, iters);
    run<16>(32, 2, iters);
    run<16>(16, 4, iters);
    for (int w : {4, 8, 16}) run<8>(w, 1, iters);
    run<8>(32, 2, iters);
    for (int w : {4, 16}) run<4>(w, 1, iters);
    run<4>(32, 2, iters);
    return 0;
}
