// tools/ubench_cost.hip -- what bounds the cost pass, and what would an energy-identity filter cost instead?
//   exact legs : gather m codeword rows + the x row per evaluated vector (quarter wave / half wave per vector, 4 or 8 vectors in flight per wave)
//   filter legs: thread per vector, SUM_j U_j[b_j] + SUM_{j<k} T_jk[b_j][b_k] from the resident f32 unaries (slice-major, 64-byte pieces in an
//                m n 1 KiB buffer) and the pair tables (row-major 16 MiB / compact upper triangle 7 MiB)
// Timing only (no parity): random codes, a random fraction `f` of the vectors evaluated.  hipcc --offload-arch=gfx950 -O3 tools/ubench_cost.hip -o tools/bin/ubench_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int M = 8, H = 256;

__device__ inline float rowsum16(float v) {      // sum over the 16 lanes of a DPP row (order irrelevant here)
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

// LPV lanes per vector (16 or 32), 16-byte loads; NV = 64 / LPV vectors per round; UNR rounds in flight
template <int LPV, int UNR, int MODE>      // MODE 0 full, 1 no X, 2 no K
__global__ __launch_bounds__(256) void cost_exact(const float *__restrict__ X, const float *__restrict__ K, const uint8_t *__restrict__ rec,
                                                  const uint8_t *__restrict__ mask, float *__restrict__ out, int64_t n, int d) {
    constexpr int NV = 64 / LPV;
    const int lane = threadIdx.x & 63, sub = lane / LPV, lp = lane % LPV;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t base = w * 64; base < n; base += nwaves * 64) {
        const int64_t il = base + lane;
        const bool live = il < n;
        const uint64_t rn = live ? *reinterpret_cast<const uint64_t *>(rec + il * 8) : 0ull;
        uint64_t todo = __ballot(live && mask[live ? il : 0]);
        while (todo) {
            uint64_t r[UNR];
            int64_t ii[UNR];
            bool hv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                int sidx[NV];
                bool h[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    h[v] = todo != 0;
                    sidx[v] = h[v] ? __builtin_ctzll(todo) : 0;
                    if (h[v]) todo &= todo - 1;
                }
                int my = sidx[0];
                bool mh = h[0];
#pragma unroll
                for (int v = 1; v < NV; ++v) if (sub == v) { my = sidx[v]; mh = h[v]; }
                r[u] = __shfl(rn, my, 64);
                ii[u] = base + my;
                hv[u] = mh;
            }
            f32x4 p[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) p[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < d; c0 += 4 * LPV) {
                const int t = c0 + 4 * lp;
                f32x4 xv[UNR], kv[UNR][M];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    xv[u] = (MODE == 1) ? (f32x4){1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4 *>(X + ii[u] * (int64_t)d + t);
#pragma unroll
                    for (int k = 0; k < M; ++k)
                        kv[u][k] = (MODE == 2) ? (f32x4){(float)((r[u] >> (8 * k)) & 255), 0.f, 0.f, 0.f}
                                               : *reinterpret_cast<const f32x4 *>(K + ((int64_t)(k * H) + ((r[u] >> (8 * k)) & 0xffu)) * d + t);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    f32x4 cb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < M; ++k) cb = cb + kv[u][k];
                    const f32x4 rr = cb - xv[u];
                    p[u] = p[u] + rr * rr;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float v = (p[u].x + p[u].y) + (p[u].z + p[u].w);
                v = rowsum16(v);
                if (LPV == 32) v += __shfl_xor(v, 16, 64);
                if (hv[u] && lp == 0) out[ii[u]] = v;
            }
        }
    }
}

// thread per vector.  U slice-major f32: U[(j*16 + a/16) * n*16 + i*16 + a%16].  T row-major [j][k][b][a] (TRI = 0) or compact upper triangle
// [pair(j<k)][b_j][b_k] (TRI = 1).  NU = number of U reads (8 = all codes; fewer = "only the changed codes" + one 32-byte cached record)
template <int TRI, int NU, int NT>
__global__ __launch_bounds__(256) void cost_filter(const float *__restrict__ U, const float *__restrict__ T, const uint8_t *__restrict__ rec,
                                                   const uint8_t *__restrict__ mask, const float *__restrict__ ucache, float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !mask[i]) return;
    const uint64_t r = *reinterpret_cast<const uint64_t *>(rec + i * 8);
    int b[M];
#pragma unroll
    for (int k = 0; k < M; ++k) b[k] = (int)((r >> (8 * k)) & 255);
    float uu[M], tt[28];
#pragma unroll
    for (int j = 0; j < M; ++j) uu[j] = (j < NU) ? U[((int64_t)(j * 16 + (b[j] >> 4)) * n + i) * 16 + (b[j] & 15)] : 0.f;
    f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if (NU < M) { c0 = *reinterpret_cast<const f32x4 *>(ucache + i * 8); c1 = *reinterpret_cast<const f32x4 *>(ucache + i * 8 + 4); }
    int q = 0;
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int k = j + 1; k < M; ++k, ++q)
            tt[q] = (q < NT) ? (TRI ? T[((int64_t)q * H + b[j]) * H + b[k]] : T[(((int64_t)j * M + k) * H + b[k]) * H + b[j]]) : 0.f;
    double f = (double)c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
#pragma unroll
    for (int j = 0; j < M; ++j) f += (double)uu[j];
#pragma unroll
    for (int e = 0; e < 28; ++e) f += (double)tt[e];
    out[i] = (float)f;
}

template <class F>
static float timeit(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int d = argc > 2 ? atoi(argv[2]) : 128;
    float *X, *K, *U, *T, *out, *uc;
    uint8_t *rec, *mask;
    CK(hipMalloc(&X, sizeof(float) * n * d));
    CK(hipMalloc(&K, sizeof(float) * M * H * d));
    CK(hipMalloc(&U, sizeof(float) * n * M * H));
    CK(hipMalloc(&T, sizeof(float) * M * M * H * H));
    CK(hipMalloc(&out, sizeof(float) * n));
    CK(hipMalloc(&uc, sizeof(float) * n * 8));
    CK(hipMalloc(&rec, n * 8));
    CK(hipMalloc(&mask, n));
    CK(hipMemset(X, 0, sizeof(float) * n * d)); CK(hipMemset(K, 0, sizeof(float) * M * H * d)); CK(hipMemset(U, 0, sizeof(float) * n * M * H));
    CK(hipMemset(T, 0, sizeof(float) * M * M * H * H)); CK(hipMemset(uc, 0, sizeof(float) * n * 8));
    std::vector<uint8_t> hr(n * 8), hm(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &v : hr) v = (uint8_t)(rnd() >> 33);
    CK(hipMemcpy(rec, hr.data(), n * 8, hipMemcpyHostToDevice));
    const unsigned grid_w = 2048, grid_t = (unsigned)((n + 255) / 256);
    for (double f : {1.0, 0.4, 0.1}) {
        for (auto &v : hm) v = ((rnd() >> 11) * (1.0 / 9007199254740992.0)) < f;
        CK(hipMemcpy(mask, hm.data(), n, hipMemcpyHostToDevice));
        const double ev = f * n;
        printf("== n=%lld d=%d evaluated fraction %.2f\n", (long long)n, d, f);
#define EX(LPV, UNR, MODE, name) { float us = timeit([&] { cost_exact<LPV, UNR, MODE><<<grid_w, 256>>>(X, K, rec, mask, out, n, d); }, 10); \
        printf("%-34s %8.1f us  %6.2f ns/vector  %6.2f TB/s (rows+x)\n", name, us, us * 1e3 / ev, ev * (M + 1) * d * 4 / us / 1e6); }
        EX(16, 1, 0, "exact quarter-wave x4 (cost4)")
        EX(16, 2, 0, "exact quarter-wave x8")
        EX(32, 1, 0, "exact half-wave x2")
        EX(32, 2, 0, "exact half-wave x4")
        EX(32, 4, 0, "exact half-wave x8")
        EX(16, 1, 1, "exact quarter-wave x4, no X")
        EX(16, 1, 2, "exact quarter-wave x4, no K")
        EX(16, 2, 1, "exact quarter-wave x8, no X")
#define FI(TRI, NU, NT, name) { float us = timeit([&] { cost_filter<TRI, NU, NT><<<grid_t, 256>>>(U, T, rec, mask, uc, out, n); }, 10); \
        printf("%-34s %8.1f us  %6.2f ns/vector  %6.1f G reads/s\n", name, us, us * 1e3 / ev, ev * (NU + NT + (NU < M ? 2 : 0)) / us / 1e3); }
        FI(0, 8, 28, "filter 8U + 28T row-major")
        FI(1, 8, 28, "filter 8U + 28T triangle")
        FI(1, 3, 28, "filter 3U + cache + 28T tri")
        FI(1, 0, 28, "filter 0U + cache + 28T tri")
        FI(1, 8, 0, "filter 8U only")
        FI(1, 0, 0, "filter cache only")
    }
    return 0;
}
