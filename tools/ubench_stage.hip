// tools/ubench_stage.hip -- what a node update's TABLE STAGING costs per CU, chip-wide, for the ways round 6 weighs:
//   A  vgpr     the walk kernel's present scheme: slice s + 1 travels through 7 x 16 B of registers per thread while slice s is walked; per slice
//               barrier, ds_write pass, barrier (8 slices of 112 KB per node update)
//   B  dma      LDS-DMA (global_load_lds_dwordx4), DOUBLE-BUFFERED half-slices: 16 half-slices of 56 KB (7 tables x 256 codes x 32 B) land in two
//               64 KB buffers (256-byte lines of eight 32-byte slots, the eighth left alone: 14 of 16 lanes active per line), one barrier per half-slice
//   B' dma full the same with whole 1 KiB wave instructions (the free slot overwritten: what the transfer alone costs)
//   C  gather   no staging: every active vector reads its 7 table rows (512 B each, u16 levels) + its unary row straight from L2 / HBM,
//               half a wave per vector (16 B per lane)
// Every variant runs as 256 blocks x 1024 threads with 160 KB of LDS (one block per CU, all CUs at once, as in the kernel), `work` dummy item steps
// per slice (LDS reads of the staged table) to stand in for a sparse node update's walk.  Prints microseconds per node update.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o tools/bin/ubench_stage ; tools/bin/ubench_stage [nodes] [active]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NT = 1024, TAB = 7 * 256 * 4;              // 16-byte entries of one 32-candidate slice (7 tables x 256 codes x 64 B)
constexpr int HTAB = 7 * 256 * 2;                        // ... of one 16-candidate half-slice

__device__ inline unsigned lds_sum(const u32x4 *tab, int entries, int steps, unsigned seed) {
    unsigned acc = 0;
    for (int t = 0; t < steps; ++t) {
        seed = seed * 1664525u + 1013904223u;
        const u32x4 v = tab[(seed >> 8) % (unsigned)entries];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    return acc;
}

// A: register-staged slices (two barriers + a ds_write pass per slice).  ROT: the blocks of an XCD start a node update at different slices (the order
// of the slices does not matter to the minimum), so that the 32 CUs of an XCD do not ask its L2 for the same lines at the same moment.
template <int ROT>
__global__ __launch_bounds__(NT) void stage_vgpr(const u32x4 *__restrict__ Tq, int nodes, int work, unsigned *sink) {
    extern __shared__ u32x4 lds[];
    unsigned acc = 0;
    const int s0 = ROT ? (int)((blockIdx.x >> 3) & 7u) : 0;
    for (int nu = 0; nu < nodes; ++nu) {
        const u32x4 *src = Tq + (size_t)(nu & 7) * 8 * TAB;
        u32x4 nxt[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) nxt[r] = src[(size_t)s0 * TAB + threadIdx.x + r * NT];
        for (int s = 0; s < 8; ++s) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 7; ++r) lds[threadIdx.x + r * NT] = nxt[r];
            __syncthreads();
            if (s + 1 < 8) {
#pragma unroll
                for (int r = 0; r < 7; ++r) nxt[r] = src[(size_t)((s + 1 + s0) & 7) * TAB + threadIdx.x + r * NT];
            }
            acc += lds_sum(lds, TAB, work, threadIdx.x + nu);
        }
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// B: LDS-DMA, double-buffered half-slices.  MASKED: 14 lanes per 256-byte line (slot 7 of every line is left alone); else whole 1 KiB wave instructions.
template <int MASKED>
__global__ __launch_bounds__(NT) void stage_dma(const u32x4 *__restrict__ Tq, int nodes, int work, unsigned *sink) {
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned acc = 0;
    auto issue = [&](const u32x4 *src, int buf) {
        // a half-slice = 256 lines; wave w owns lines [16 w, 16 w + 16)
        if (MASKED) {
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                const int line = wave * 16 + l;
                if (lane < 14)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src + line * 14 + lane),
                                                     (__attribute__((address_space(3))) void *)(lds + buf * 4096 + line * 16), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int e0 = (wave * 4 + l) * 64;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src + e0 + lane),
                                                 (__attribute__((address_space(3))) void *)(lds + buf * 4096 + e0), 16, 0, 0);
            }
        }
    };
    for (int nu = 0; nu < nodes; ++nu) {
        const u32x4 *src = Tq + (size_t)(nu & 7) * 8 * TAB;
        issue(src, 0);
        for (int s = 0; s < 16; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                       // half-slice s has landed for everyone; everyone has left half-slice s - 1
            if (s + 1 < 16) issue(src + (size_t)(s + 1) * (MASKED ? HTAB : 4096), (s + 1) & 1);
            acc += lds_sum(lds + (s & 1) * 4096, 4096, work, threadIdx.x + nu);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// C: direct gathers.  Row-major u16 tables [7][256][256] (512-byte rows), `active` vectors per block and node, half a wave per vector.
__global__ __launch_bounds__(NT) void gather_direct(const u32x4 *__restrict__ Trow, const u32x4 *__restrict__ Uq, int nodes, int active, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    unsigned acc = 0;
    for (int nu = 0; nu < nodes; ++nu) {
        const u32x4 *T = Trow + (size_t)(nu & 7) * 7 * 256 * 32;
        for (int v0 = wave * 2; v0 < active; v0 += 32) {
            const int v = v0 + half;
            unsigned seed = (unsigned)(blockIdx.x * 4099 + v * 131 + nu * 7);
            // the unary piece: 512 B per vector from a 2 GiB stream (HBM)
            seed = seed * 1664525u + 1013904223u;
            u32x4 s = Uq[((size_t)(seed >> 6) % (size_t)(1u << 22)) * 32 + l32];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                seed = seed * 1664525u + 1013904223u;
                const u32x4 r = T[((size_t)k * 256 + ((seed >> 10) & 255u)) * 32 + l32];
                s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
            }
            unsigned mn = min(min(s.x, s.y), min(s.z, s.w));
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 32));
            acc += mn;
        }
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char **argv) {
    const int nodes = argc > 1 ? atoi(argv[1]) : 32;
    u32x4 *Tq, *Uq; unsigned *sink;
    const size_t tq_bytes = (size_t)16 << 20;                           // eight nodes x eight slices of 112 KB (7.3 MB) + slack for the padded variant
    CK(hipMalloc(&Tq, tq_bytes)); CK(hipMalloc(&Uq, (size_t)2 << 30)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(Tq, 1, tq_bytes)); CK(hipMemset(Uq, 1, (size_t)2 << 30));
    const int LDS = 160 * 1024 - 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_vgpr<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_vgpr<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_dma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&stage_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    printf("%d node updates per launch, 256 blocks x 1024 threads; microseconds per node update of a block\n", nodes);
    for (int work : {0, 4, 16, 64}) {
        const float a = timeit([&] { hipLaunchKernelGGL(stage_vgpr<0>, dim3(256), dim3(NT), LDS, 0, Tq, nodes, work, sink); }, 3);
        const float a2 = timeit([&] { hipLaunchKernelGGL(stage_vgpr<1>, dim3(256), dim3(NT), LDS, 0, Tq, nodes, work, sink); }, 3);
        printf("work %3d: A with the blocks of an XCD starting at different slices %7.2f us\n", work, a2 * 1e3 / nodes);
        const float b = timeit([&] { hipLaunchKernelGGL(stage_dma<1>, dim3(256), dim3(NT), LDS, 0, Tq, nodes, work, sink); }, 3);
        const float c = timeit([&] { hipLaunchKernelGGL(stage_dma<0>, dim3(256), dim3(NT), LDS, 0, Tq, nodes, work, sink); }, 3);
        printf("work %3d LDS reads per thread and (half-)slice:  A vgpr 8 x 112 KB %7.2f us   B dma 16 x 56 KB (14-lane lines) %7.2f us   B' dma 16 x 64 KB (whole waves) %7.2f us\n",
               work, a * 1e3 / nodes, b * 1e3 / nodes, c * 1e3 / nodes);
    }
    for (int active : {64, 128, 256, 400, 512, 768, 1024, 2048, 3968}) {
        const float g = timeit([&] { hipLaunchKernelGGL(gather_direct, dim3(256), dim3(NT), 0, 0, Tq, Uq, nodes, active, sink); }, 3);
        printf("C direct gather, %4d active vectors per block: %7.2f us per node update  (%.1f ns per vector)\n", active, g * 1e3 / nodes, g * 1e6 / nodes / active);
    }
    return 0;
}
