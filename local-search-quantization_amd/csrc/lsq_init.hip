// lsq_init.hip -- the two data-parallel kernels of the INITIALISERS, gfx950:
//
//   viterbi_kernel       ChainQ's exact chain encoder: src/encodings/encode_chain.jl:2-89 (encode_viterbi!) per vector
//                            acc_0[k]   = U_0[k]
//                            cost[k]    = acc_i[k] + bb_i[k][j]            bb_i = 2 C_i' C_{i+1}   (:103-106; plain f32 add, :52-54)
//                            mincost_i[j], minidx_i[j] = first minimum over k (strict '<' scan from k = 0, :58-66)
//                            acc_{i+1}[j] = U_{i+1}[j] + mincost_i[j]      (:41-45, :70-72)
//                            b_{m-1} = first argmin acc_{m-1};  b_i = minidx_i[b_{i+1}]   (:74-80)
//   unary_argmin_kernel  the nearest-codeword assignment of PQ / OPQ and of their k-means (src/pq/PQ.jl:12-41, src/opq/kmeans.jl:6-75,
//                        src/opq/OPQ.jl:60-66,88-91): b_j = first argmin_a ( ||c_ja||^2 - 2 <x, c_ja> ) per codebook j -- a codebook that is zero
//                        outside its sub-space gives exactly the sub-space distance minus ||x_sub||^2, which no candidate sees.
//
// Both read the f32 unaries the path's own unary GEMM wrote (row-major planes U[(j n + i) 256 + a]: the same bits as lsq_get_unaries) and, for the
// chain, the pair tables of prepare_tables (T[((j m + k) 256 + b) 256 + a] = 2 <c_kb, c_ja>: row b of block (j = i + 1, k = i) is bb_i[b][:]).
// Results are integer codes: bit-exact against oracle/init_oracle.py, which restates the same arithmetic on the oracle's unaries and tables.
//
// Viterbi, the shape of the work: (m - 1) min-plus steps of a 256 x 256 table per vector = 65 536 add + compare + select per step, all VALU
// (4 instructions per (source, target) pair; nothing to reuse across vectors but the table).  A 1024-thread block takes NV vectors through the chain
// together: the step's table passes through LDS in chunks of 64 source rows (64 KB, staged once per chunk for all NV vectors: 1.8 us against ~30 us of
// arithmetic), a wave owns NV / 16 vectors, a lane four target codes (one ds_read_b128 per source row, shared by the wave's vectors); the running
// values acc[k] sit in LDS so that a source's value reaches all 64 lanes as a broadcast read; the back pointers (m - 1) x 256 bytes per vector stay in LDS
// until the trace.  Rate: ~4096 wave-instructions per vector and step -> ~50 ns per vector at m = 8 chip-wide (10^5 training vectors: ~5 ms).
#include "lsq_internal.h"
#include "lsq_wave.h"

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int VIT_NT = 1024, VIT_CH = 64;                     // threads per block; source rows per staged chunk

template <int NV>
constexpr int vit_lds_bytes(int m) { return VIT_CH * LSQ_H * 4 + NV * LSQ_H * 4 + NV * (m - 1) * LSQ_H; }

// lexicographic (value, index) minimum over the wave == the strict-'<' scan from index 0 (finite inputs: the reference's scans say nothing useful about NaN)
__device__ inline void wave_first_min(float &v, int &a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oa = __shfl_xor(a, o, 64);
        if (ov < v || (ov == v && oa < a)) { v = ov; a = oa; }
    }
}

// first minimum of the lane's four consecutive candidates 4 lane .. 4 lane + 3 (ascending scan, strict '<')
__device__ inline void lane_first_min(const f32x4 &s, int lane, float &v, int &a) {
    v = s.x; a = 4 * lane;
    if (s.y < v) { v = s.y; a = 4 * lane + 1; }
    if (s.z < v) { v = s.z; a = 4 * lane + 2; }
    if (s.w < v) { v = s.w; a = 4 * lane + 3; }
}

template <int NV>
__global__ __launch_bounds__(VIT_NT) void viterbi_kernel(const float *__restrict__ U, const float *__restrict__ T, int64_t n, int m,
                                                         uint8_t *__restrict__ codes) {
    constexpr int VPW = NV / 16;                                // vectors per wave
    extern __shared__ f32x4 vit_lds[];
    f32x4 *tabS = vit_lds;                                      // [VIT_CH][64] f32x4: rows of the step's table
    float *accS = reinterpret_cast<float *>(vit_lds + VIT_CH * 64);      // [NV][256]
    uint8_t *backS = reinterpret_cast<uint8_t *>(accS + NV * LSQ_H);     // [NV][m - 1][256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nbatch = (n + NV - 1) / NV;
    for (int64_t batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        int64_t vi[VPW];
        bool live[VPW];
        f32x4 mv[VPW];                                          // running minima of the lane's four targets
        u32x4_t mi[VPW];                                        // ... and the sources that attain them
#pragma unroll
        for (int e = 0; e < VPW; ++e) {
            vi[e] = batch * NV + wave * VPW + e;
            live[e] = vi[e] < n;
            const int64_t iv = live[e] ? vi[e] : 0;
            const f32x4 u0 = *reinterpret_cast<const f32x4 *>(U + iv * LSQ_H + 4 * lane);       // plane 0
            *reinterpret_cast<f32x4 *>(accS + (wave * VPW + e) * LSQ_H + 4 * lane) = u0;
        }
        for (int i = 0; i + 1 < m; ++i) {
            const float *Ti = T + ((int64_t)(i + 1) * m + i) * LSQ_H * LSQ_H;                  // rows = source codes of codebook i, columns = target codes of i + 1
            for (int c = 0; c < LSQ_H / VIT_CH; ++c) {
                __syncthreads();                                // the previous chunk has been read by every wave (and accS of this step is written)
#pragma unroll
                for (int r = 0; r < VIT_CH * 64 / VIT_NT; ++r)
                    tabS[threadIdx.x + r * VIT_NT] = reinterpret_cast<const f32x4 *>(Ti + (int64_t)c * VIT_CH * LSQ_H)[threadIdx.x + r * VIT_NT];
                __syncthreads();
                int k0 = 0;
                if (c == 0) {                                   // the scan starts AT source 0 (encode_chain.jl:58-59), whatever its value
                    const f32x4 row = tabS[lane];
#pragma unroll
                    for (int e = 0; e < VPW; ++e) {
                        const float a0 = accS[(wave * VPW + e) * LSQ_H];
                        mv[e] = (f32x4){a0 + row.x, a0 + row.y, a0 + row.z, a0 + row.w};
                        mi[e] = (u32x4_t){0u, 0u, 0u, 0u};
                    }
                    k0 = 1;
                }
#pragma unroll 4
                for (int kk = k0; kk < VIT_CH; ++kk) {
                    const f32x4 row = tabS[kk * 64 + lane];
                    const unsigned k = (unsigned)(c * VIT_CH + kk);
#pragma unroll
                    for (int e = 0; e < VPW; ++e) {
                        const float ak = accS[(wave * VPW + e) * LSQ_H + c * VIT_CH + kk];     // broadcast read
                        const float c0 = ak + row.x, c1 = ak + row.y, c2 = ak + row.z, c3 = ak + row.w;
                        if (c0 < mv[e].x) { mv[e].x = c0; mi[e].x = k; }
                        if (c1 < mv[e].y) { mv[e].y = c1; mi[e].y = k; }
                        if (c2 < mv[e].z) { mv[e].z = c2; mi[e].z = k; }
                        if (c3 < mv[e].w) { mv[e].w = c3; mi[e].w = k; }
                    }
                }
            }
            __syncthreads();                                    // every wave is done with accS of step i (its own rows only -- but the barrier also orders the chunk loop)
#pragma unroll
            for (int e = 0; e < VPW; ++e) {
                const int v = wave * VPW + e;
                reinterpret_cast<uint32_t *>(backS + ((size_t)v * (m - 1) + i) * LSQ_H)[lane] = mi[e].x | (mi[e].y << 8) | (mi[e].z << 16) | (mi[e].w << 24);
                const int64_t iv = live[e] ? vi[e] : 0;
                const f32x4 un = *reinterpret_cast<const f32x4 *>(U + ((int64_t)(i + 1) * n + iv) * LSQ_H + 4 * lane);
                *reinterpret_cast<f32x4 *>(accS + v * LSQ_H + 4 * lane) = (f32x4){un.x + mv[e].x, un.y + mv[e].y, un.z + mv[e].z, un.w + mv[e].w};
            }
        }
        __syncthreads();
        // the last codebook's first argmin, then the trace through the back pointers
#pragma unroll
        for (int e = 0; e < VPW; ++e) {
            const int v = wave * VPW + e;
            const f32x4 fin = *reinterpret_cast<const f32x4 *>(accS + v * LSQ_H + 4 * lane);
            float bv; int ba;
            lane_first_min(fin, lane, bv, ba);
            wave_first_min(bv, ba);
            if (lane == 0 && live[e]) {
                int code = ba;
                codes[vi[e] * m + (m - 1)] = (uint8_t)code;
                for (int i = m - 2; i >= 0; --i) {
                    code = backS[((size_t)v * (m - 1) + i) * LSQ_H + code];
                    codes[vi[e] * m + i] = (uint8_t)code;
                }
            }
        }
        __syncthreads();                                        // backS / accS are reused by the next batch
    }
}

// one wave per (vector, codebook) row of 256 unaries: first argmin and its value
__global__ __launch_bounds__(256) void unary_argmin_kernel(const float *__restrict__ U, int64_t n, int m, uint8_t *__restrict__ codes,
                                                           float *__restrict__ minval) {
    const int lane = threadIdx.x & 63;
    const int64_t nrows = n * m, stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += stride) {
        const int64_t j = row / n, i = row - j * n;             // plane-major rows: consecutive waves read consecutive KiB
        const f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(U + row * LSQ_H + 4 * lane));
        float bv; int ba;
        lane_first_min(s, lane, bv, ba);
        wave_first_min(bv, ba);
        if (lane == 0) {
            codes[i * m + j] = (uint8_t)ba;
            if (minval) minval[i * m + j] = bv;
        }
    }
}

}  // namespace

int lsq_launch_viterbi(hipStream_t s, const float *U, const float *T, int64_t n, int m, uint8_t *codes) {
    if (n <= 0) return LSQ_OK;
    if (m < 2 || m > LSQ_MAX_M) { lsq_set_error("lsq_launch_viterbi: a chain needs 2 <= m <= 16 codebooks (got %d)", m); return LSQ_EINVAL; }
    // two vectors per wave while their back pointers fit next to the table chunk (m <= 8); one above
    if (m <= 8) {
        constexpr int NV = 32;
        const int lds = vit_lds_bytes<NV>(m);
        static LdsOptIn optin;
        LSQ_TRY(optin_lds(optin, &viterbi_kernel<NV>, vit_lds_bytes<NV>(8)));
        const int64_t nb = (n + NV - 1) / NV;
        hipLaunchKernelGGL(viterbi_kernel<NV>, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(VIT_NT), lds, s, U, T, n, m, codes);
    } else {
        constexpr int NV = 16;
        const int lds = vit_lds_bytes<NV>(m);
        static LdsOptIn optin;
        LSQ_TRY(optin_lds(optin, &viterbi_kernel<NV>, vit_lds_bytes<NV>(LSQ_MAX_M)));
        const int64_t nb = (n + NV - 1) / NV;
        hipLaunchKernelGGL(viterbi_kernel<NV>, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(VIT_NT), lds, s, U, T, n, m, codes);
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_unary_argmin(hipStream_t s, const float *U, int64_t n, int m, uint8_t *codes, float *minval) {
    if (n <= 0) return LSQ_OK;
    const int64_t want = (n * m + 3) / 4;
    hipLaunchKernelGGL(unary_argmin_kernel, dim3((unsigned)(want < 256 * 8 ? want : 256 * 8)), dim3(256), 0, s, U, n, m, codes, minval);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
