// tools/ubench_sparse.hip -- what does HBM deliver, and what does FETCH_SIZE tally, when a kernel reads ISOLATED pieces of 128-byte lines?
// (the walk kernel's sparse sweeps read one 64-byte (m <= 8) or 32-byte (m > 8) piece of a 128-byte line whose other
// half belongs to an inactive neighbour; the "traffic / algorithmic" ratios of profiles/r05_traffic_by_sweep.txt applied the guide's x 2 -- calibrated
// on wide coalesced reads only -- to that pattern.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sparse.hip -o tools/bin/ubench_sparse
//   tools/bin/ubench_sparse                                   timings: useful GB/s and line GB/s per pattern
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/bin/ubench_sparse 1     one launch per pattern (kernel names carry the pattern): the counter per known byte count
// The buffer is 2 GiB (8 x the 256 MiB Infinity Cache); every pattern touches each of its lines exactly once per launch, pieces are 16 bytes per lane
// (global_load_dwordx4, nontemporal as the walk's level stream), lanes of a piece are adjacent lanes of one wave (4 lanes = 64 B, 2 lanes = 32 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PIECE bytes (16 .. 128) read at the head of every STRIDE-byte window; the windows a wave instruction covers are consecutive (64 * 16 / PIECE of them).
// PERM: the window index is bit-reversed inside blocks of 2^20 windows (isolated pieces in random-looking lines instead of a regular stride).
template <int PIECE, int STRIDE, int PERM>
__global__ __launch_bounds__(256) void rd(const char *__restrict__ buf, int64_t nwin, unsigned *__restrict__ sink) {
    constexpr int LPP = PIECE / 16;                                   // lanes per piece
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    u32x4 acc = (u32x4){0u, 0u, 0u, 0u};
    for (int64_t t = tid; t < nwin * LPP; t += nth) {
        int64_t w = t / LPP;
        const int q = (int)(t % LPP);
        if (PERM) {
            const unsigned lo = (unsigned)(w & 0xfffff);
            w = (w & ~(int64_t)0xfffff) | (int64_t)(__brev(lo) >> 12);
        }
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(buf + w * STRIDE + q * 16));
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = acc.x;      // never true on the zero-filled / patterned buffer: keeps the loads
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

template <int PIECE, int STRIDE, int PERM>
static void leg(const char *name, const char *buf, int64_t bytes, unsigned *sink, int reps) {
    const int64_t nwin = bytes / STRIDE;
    const float ms = timeit([&] { hipLaunchKernelGGL((rd<PIECE, STRIDE, PERM>), dim3(256 * 8), dim3(256), 0, 0, buf, nwin, sink); }, reps);
    const double useful = (double)nwin * PIECE, lines = (double)nwin * (STRIDE < 128 ? STRIDE : 128);
    printf("%-44s useful %8.1f MB  lines touched %8.1f MB  %7.3f ms  useful %7.1f GB/s  whole-line %7.1f GB/s\n", name, useful / 1e6, lines / 1e6, ms,
           useful / ms / 1e6, lines / ms / 1e6);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int64_t bytes = 2ll << 30;
    char *buf; unsigned *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    printf("buffer %lld MiB, %d timed launches per pattern (kernel template arguments = <piece bytes, stride bytes, permuted>)\n", (long long)(bytes >> 20), reps);
    leg<128, 128, 0>("dense: every byte (8 lanes x 16 B per line)", buf, bytes, sink, reps);
    leg<64, 128, 0>("64-byte piece of every 128-byte line", buf, bytes, sink, reps);
    leg<32, 128, 0>("32-byte piece of every 128-byte line", buf, bytes, sink, reps);
    leg<16, 128, 0>("16-byte piece of every 128-byte line", buf, bytes, sink, reps);
    leg<64, 256, 0>("64-byte piece of every OTHER line", buf, bytes, sink, reps);
    leg<32, 256, 0>("32-byte piece of every OTHER line", buf, bytes, sink, reps);
    leg<64, 512, 0>("64-byte piece of every 4th line", buf, bytes, sink, reps);
    leg<64, 1024, 1>("64-byte piece, scattered (1 line in 8)", buf, bytes, sink, reps);
    leg<32, 1024, 1>("32-byte piece, scattered (1 line in 8)", buf, bytes, sink, reps);
    leg<128, 1024, 1>("whole line, scattered (1 line in 8)", buf, bytes, sink, reps);
    leg<64, 64, 0>("dense in 64-byte pieces (control)", buf, bytes, sink, reps);
    return 0;
}
