// tools/ubench_write.hip -- HBM write rates for the store patterns of the unary GEMM's epilogue (8 GB of f32 + 4 GB of u16 per 10^6 vectors).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_write.hip -o tools/bin/ubench_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// PAT 0: 16 B per lane, wave-contiguous (1 KiB per instruction).  PAT 1: 4 B per lane; a wave instruction writes four 64-byte pieces (16 lanes each)
// that lie `piece_stride` bytes apart, successive instructions of a thread advance by 64 B (the next row of the same slice): the GEMM epilogue's pattern.
// PAT 2: 16 B per lane, four lanes = one 64-byte piece, pieces as in PAT 1 (what a 4 x 4 lane transpose would give).  NT: nontemporal stores.
template <int PAT, int NT>
__global__ __launch_bounds__(256) void wk(float *out, int64_t nbytes, int64_t piece_stride) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    const int64_t wave = tid >> 6, nwaves = nth >> 6;
    if (PAT == 0) {
        for (int64_t o = tid * 16; o < nbytes; o += nth * 16) {
            f32x4 v = (f32x4){(float)o, 1.f, 2.f, 3.f};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(out) + o));
            else *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(out) + o) = v;
        }
    } else if (PAT == 1) {
        // region = 4 pieces-columns x R rows x 64 B; a wave owns 16 consecutive rows of its 4 piece columns per step
        const int64_t rows = piece_stride / 64;                     // rows per piece column (= vectors)
        const int64_t ncolgrp = nbytes / (4 * piece_stride);        // groups of 4 piece columns
        const int64_t steps = rows / 16;
        for (int64_t w = wave; w < ncolgrp * steps; w += nwaves) {
            const int64_t cg = w / steps, st = w % steps;
            char *base = reinterpret_cast<char *>(out) + (cg * 4 + (lane >> 4)) * piece_stride + st * 16 * 64 + (lane & 15) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (NT) __builtin_nontemporal_store((float)r, reinterpret_cast<float *>(base + r * 64));
                else *reinterpret_cast<float *>(base + r * 64) = (float)r;
            }
        }
    } else {
        const int64_t rows = piece_stride / 64;
        const int64_t ncolgrp = nbytes / (4 * piece_stride);
        const int64_t steps = rows / 16;
        for (int64_t w = wave; w < ncolgrp * steps; w += nwaves) {
            const int64_t cg = w / steps, st = w % steps;
            // 16 lanes cover 4 rows x 64 B of one piece column; 4 instructions cover the 16 rows
            char *base = reinterpret_cast<char *>(out) + (cg * 4 + (lane >> 4)) * piece_stride + st * 16 * 64 + ((lane & 15) >> 2) * 64 + (lane & 3) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v = (f32x4){(float)r, 1.f, 2.f, 3.f};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(base + r * 256));
                else *reinterpret_cast<f32x4 *>(base + r * 256) = v;
            }
        }
    }
}
template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const int64_t nbytes = 8ll << 30, stride = 1000000ll * 64;      // 8 GiB region; piece columns of 10^6 rows x 64 B (8192 piece columns fit: 2048 groups of 4)
    float *out; CK(hipMalloc(&out, nbytes));
    const int64_t used = (nbytes / (4 * stride)) * 4 * stride;
#define R(P, N, name) { float ms = timeit([&] { wk<P, N><<<2048, 256>>>(out, P == 0 ? used : nbytes, stride); }, 5); printf("%-60s %7.3f ms  %6.2f TB/s\n", name, ms, used / ms / 1e9); }
    R(0, 0, "16 B per lane, contiguous")
    R(0, 1, "16 B per lane, contiguous, nontemporal")
    R(1, 0, "4 B per lane, 4 x 64-B pieces per instruction (epilogue)")
    R(1, 1, "4 B per lane, 4 x 64-B pieces per instruction, nontemporal")
    R(2, 0, "16 B per lane, 4 x 256-B runs per instruction")
    R(2, 1, "16 B per lane, 4 x 256-B runs per instruction, nontemporal")
    return 0;
}
