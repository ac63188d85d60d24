// tools/ubench_gemm.hip -- variants of the chain-exact fp32 MFMA contraction of csrc/lsq_gemm.hip on the cfg2 unary shape
// (M x 128) x (2048 x 128)^T -> f32 slice-major planes + u16 level planes, timed with HIP events and checked bit for bit against a scalar fmaf chain.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_gemm.hip -o tools/bin/ubench_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int H = 256, BM = 128, BN = 128, SL = 16, SLQ = 32;

struct Args {
    const float *A, *B, *sci;
    float *D;
    uint16_t *Dq;
    int64_t M;
    int N, Kd;
    float lo, inv;
    int64_t row_tiles;
    int col_tiles;
};

__device__ inline int64_t d_off(int64_t M, int64_t r, int c) {      // slice-major f32 planes
    const int a = c % H;
    return (int64_t)(c / H) * (M * H) + (int64_t)(a / SL) * (M * SL) + (a % SL) + r * SL;
}
__device__ inline int64_t q_off(int64_t M, int64_t r, int c) {
    const int a = c % H;
    return (int64_t)(c / H) * (M * H) + (int64_t)(a / SLQ) * (M * SLQ) + (a % SLQ) + r * SLQ;
}

// reference: scalar chain
__global__ void ref_kernel(Args g, int64_t rows) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * g.N) return;
    const int64_t r = e / g.N;
    const int c = (int)(e % g.N);
    float acc = 0.f;
    for (int t = 0; t < g.Kd; ++t) acc = fmaf(g.A[r * g.Kd + t], -2.0f * g.B[(int64_t)c * g.Kd + t], acc);
    const float v = acc + g.sci[c];
    g.D[d_off(g.M, r, c)] = v;
    const float qf = rintf((v - g.lo) * g.inv);
    g.Dq[q_off(g.M, r, c)] = (uint16_t)(unsigned)__builtin_amdgcn_fmed3f(qf, 0.f, 65535.f);
}

// ---- epilogue shared by the variants: acc[2][2] of a wave's 64 x 64 sub-tile -> f32 stores + u16 levels through an LDS tile ----------------
template <int CTRL>
__device__ inline float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
// 4 x 4 transpose of r[0..3] across the four lanes of a quad: afterwards lane i of the quad holds, in r[0..3], what lanes 0..3 held in r[i]
__device__ inline void quad_transpose(float (&r)[4], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
    {   // exchange across lane ^ 1 (quad_perm [1,0,3,2] = 0xB1).  The DPP reads happen with ALL lanes active (a select inside a branch would read inactive lanes)
        const float x0 = dppf<0xB1>(r[0]), x1 = dppf<0xB1>(r[1]), x2 = dppf<0xB1>(r[2]), x3 = dppf<0xB1>(r[3]);
        const float a = o1 ? x1 : r[0], b = o1 ? r[1] : x0, c = o1 ? x3 : r[2], d = o1 ? r[3] : x2;
        r[0] = a; r[1] = b; r[2] = c; r[3] = d;
    }
    {   // exchange across lane ^ 2 (quad_perm [2,3,0,1] = 0x4E)
        const float x0 = dppf<0x4E>(r[0]), x1 = dppf<0x4E>(r[1]), x2 = dppf<0x4E>(r[2]), x3 = dppf<0x4E>(r[3]);
        const float a = o2 ? x2 : r[0], c = o2 ? r[2] : x0, b = o2 ? x3 : r[1], d = o2 ? r[3] : x1;
        r[0] = a; r[1] = b; r[2] = c; r[3] = d;
    }
}

template <int QLD, int EPF = 3>      // EPF bit 0: f32 stores, bit 1: u16 levels, bit 2: f32 stores as 16-byte stores after a quad transpose, bit 3: the library's level arithmetic (row shift from LDS, column shift, range flag)
__device__ inline void epilogue(const Args &g, const f32x16 (&acc)[2][2], int64_t row0, int col0, int wy, int wx, int l31, int lhi, uint16_t *qtile, int tid, const float *sgs = nullptr, unsigned *flagw = nullptr) {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int c = col0 + wx * 64 + tj * 32 + l31;
        const float add = g.sci[c];
        const int64_t lrow = row0 + wy * 64 + 4 * lhi;
        float *__restrict__ Dl = g.D + d_off(g.M, lrow, c);
        uint16_t *__restrict__ ql = qtile + (wy * 64 + 4 * lhi) * QLD + wx * 64 + tj * 32 + l31;
        if (EPF & 4) {
            // lane (column c, rows ro .. ro + 3)  ->  lane (row ro + (lane & 3), columns 4 (l31 >> 2) .. + 3): one 16-byte store per group of four rows
            const int cq = col0 + wx * 64 + tj * 32 + (l31 & ~3);
            float *__restrict__ Dq4 = g.D + d_off(g.M, lrow + (l31 & 3), cq);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float r4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r4[e] = acc[ti][tj][4 * gq + e] + add;
                    quad_transpose(r4, l31);
                    *reinterpret_cast<f32x4 *>(Dq4 + (int64_t)(ti * 32 + 8 * gq) * SL) = (f32x4){r4[0], r4[1], r4[2], r4[3]};
                }
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ti * 32 + (r & 3) + 8 * (r >> 2);
                const float v = acc[ti][tj][r] + add;
                if (EPF & 1) Dl[(int64_t)ro * SL] = v;
                if ((EPF & 2) && (EPF & 8)) {
                    const float sg = sgs[wy * 64 + 4 * lhi + ro];
                    const float qf = rintf(((v + add * 1e-3f) - sg) * g.inv);
                    const float qc = __builtin_amdgcn_fmed3f(qf, 0.f, 65535.f);
                    if (!(qf == qc)) atomicOr(flagw + ((row0 + ro) >> 1), 1u);
                    ql[ro * QLD] = (uint16_t)(unsigned)qc;
                } else if (EPF & 2) {
                    const float qf = rintf((v - g.lo) * g.inv);
                    ql[ro * QLD] = (uint16_t)(unsigned)__builtin_amdgcn_fmed3f(qf, 0.f, 65535.f);
                } else if (!(EPF & 1) && v == 12345.678f) ql[0] = 1;
            }
    }
    if (!(EPF & 2)) return;
    __syncthreads();
    const int plane = col0 / H, a0 = col0 % H;
#pragma unroll
    for (int i = 0; i < BM * BN / 8 / 256; ++i) {
        const int e = tid + i * 256, rl = e / (BN / 8), ch = e % (BN / 8);
        const int a = a0 + 8 * ch;
        *reinterpret_cast<uint4 *>(g.Dq + (int64_t)plane * (g.M * (int64_t)H) + ((int64_t)(a / SLQ) * g.M + (row0 + rl)) * SLQ + (a % SLQ)) =
            *reinterpret_cast<const uint4 *>(qtile + rl * QLD + 8 * ch);
    }
}

// ---- V0: the library's structure (BK = 16, LD = 18, register double-buffered, 4 blocks per CU) ----------------------------------------------
template <int BK, int UNROLL_FIXED, int NOEPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void v0_kernel(Args g) {
    constexpr int LD = BK + 2, NE = BM * BK / 4 / 256, Q4 = BK / 4;
    __shared__ float smem[2 * BM * LD + 2 * BN * LD];
    float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(smem);
    float (*Bs)[BN * LD] = reinterpret_cast<float (*)[BN * LD]>(smem + 2 * BM * LD);
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t s = b >> 3;
    const int64_t rt = (s / g.col_tiles) * 8 + xcd;
    const int ct = (int)(s % g.col_tiles);
    if (rt >= g.row_tiles) return;
    const int64_t row0 = rt * BM;
    const int col0 = ct * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[NE], rb[NE];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;
            ra[i] = *reinterpret_cast<const float4 *>(g.A + (row0 + r) * g.Kd + k0 + 4 * q);
            rb[i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)(col0 + r) * g.Kd + k0 + 4 * q);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;
            float *oa = As[buf] + r * LD + 4 * q, *ob = Bs[buf] + r * LD + 4 * q;
            oa[0] = ra[i].x; oa[1] = ra[i].y; oa[2] = ra[i].z; oa[3] = ra[i].w;
            ob[0] = rb[i].x * -2.f; ob[1] = rb[i].y * -2.f; ob[2] = rb[i].z * -2.f; ob[3] = rb[i].w * -2.f;
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < g.Kd; k0 += BK) {
        const bool more = k0 + BK < g.Kd;
        if (more) gload(k0 + BK);
        const float *ap = As[cur] + (wy * 64 + l31) * LD + lhi;
        const float *bp = Bs[cur] + (wx * 64 + l31) * LD + lhi;
        if (UNROLL_FIXED) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = ap[kk], a1 = ap[32 * LD + kk], b0 = bp[kk], b1 = bp[32 * LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else {
            const int kend = (g.Kd - k0 < BK) ? ((g.Kd - k0 + 1) & ~1) : BK;
            for (int kk = 0; kk < kend; kk += 2) {
                const float a0 = ap[kk], a1 = ap[32 * LD + kk], b0 = bp[kk], b1 = bp[32 * LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (NOEPI == 1) {      // K loop only: one store per thread keeps the MFMAs alive
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) g.D[tid] = sacc;
        return;
    }
    if (NOEPI == 7 || NOEPI == 8) {
        __shared__ float sgs[BM];
        if (tid < BM) sgs[tid] = g.lo + 1e-3f * (float)((row0 + tid) & 15);
        __syncthreads();
        if (NOEPI == 7) epilogue<BN + 8, 11>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid, sgs, reinterpret_cast<unsigned *>(g.Dq));
        else epilogue<BN + 8, 14>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid, sgs, reinterpret_cast<unsigned *>(g.Dq));
        return;
    }
    if (NOEPI == 5) epilogue<BN + 8, 6>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
    else if (NOEPI == 6) epilogue<BN + 8, 4>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
    else if (NOEPI == 3) epilogue<BN + 8, 1>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
    else if (NOEPI == 4) epilogue<BN + 8, 2>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
    else epilogue<BN + 8>(g, acc, NOEPI == 2 ? (row0 & 1023) & ~127ll : row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
}

// ---- V1: persistent blocks; operands k-PERMUTED in LDS so that one ds_read_b128 feeds four k-steps of a lane; whole K = 128 panel of B kept?  no:
// K chunks of 32, the A / B chunk of the NEXT step (possibly of the next tile) prefetched into registers during the MFMAs -------------------------------
// LDS row layout for a chunk of BK = 32 k's: [k0 k2 k4 ... k30 | k1 k3 ... k31], row stride LD = 32 + 4 floats: lane (row, lhi) reads 16 B pieces at
// row * LD + lhi * 16 + 4 * g  (g = 0..3) = its operand for k-steps 4 g .. 4 g + 3.
template <int BPC, int NOEPI, int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BPC, BPC))) void v1_kernel(Args g) {
    constexpr int LD = BK + 4, NE = BM * BK / 4 / 256, Q4 = BK / 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LD + 2 * BN * LD];
    float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(smem);
    float (*Bs)[BN * LD] = reinterpret_cast<float (*)[BN * LD]>(smem + 2 * BM * LD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int64_t ntiles = g.row_tiles * g.col_tiles;
    const int nchunk = g.Kd / BK;
    float4 ra[NE], rb[NE];
    // tile sequence of this block: tile index t -> (rt, ct) with the col tiles of one row panel adjacent (same XCD: blockIdx % 8 fixed per block)
    auto tile_of = [&](int64_t t, int64_t &row0, int &col0) {
        const int64_t rt = t / g.col_tiles;
        row0 = rt * BM;
        col0 = (int)(t % g.col_tiles) * BN;
    };
    auto gload = [&](int64_t row0, int col0, int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;
            ra[i] = *reinterpret_cast<const float4 *>(g.A + (row0 + r) * g.Kd + k0 + 4 * q);
            rb[i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)(col0 + r) * g.Kd + k0 + 4 * q);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;      // k = 4 q .. 4 q + 3: even ones to the first half at (4 q) / 2, odd ones to the second
            float *oa = As[buf] + r * LD + 2 * q, *ob = Bs[buf] + r * LD + 2 * q;
            *reinterpret_cast<float2 *>(oa) = make_float2(ra[i].x, ra[i].z);
            *reinterpret_cast<float2 *>(oa + BK / 2) = make_float2(ra[i].y, ra[i].w);
            *reinterpret_cast<float2 *>(ob) = make_float2(rb[i].x * -2.f, rb[i].z * -2.f);
            *reinterpret_cast<float2 *>(ob + BK / 2) = make_float2(rb[i].y * -2.f, rb[i].w * -2.f);
        }
    };
    // persistent loop: block b of XCD x = b % 8 takes row panels x, x + 8, ... : tiles (rt, all ct); blocks of one XCD share them round robin
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int64_t panels_x = (g.row_tiles - xcd + 7) / 8;                 // row panels of this XCD
    const int64_t tiles_x = panels_x * g.col_tiles;
    int64_t row0 = 0; int col0 = 0;
    int64_t t = bx;
    if (t >= tiles_x) return;
    auto tile_x = [&](int64_t tx, int64_t &r0, int &c0) { r0 = ((tx / g.col_tiles) * 8 + xcd) * BM; c0 = (int)(tx % g.col_tiles) * BN; };
    tile_x(t, row0, col0);
    gload(row0, col0, 0);
    int cur = 0;
    (void)ntiles; (void)tile_of;
    while (true) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        sstore(cur);
        __syncthreads();
        const int64_t tn = t + nbx;
        int64_t nrow0 = row0; int ncol0 = col0;
        const bool have_next = tn < tiles_x;
        if (have_next) tile_x(tn, nrow0, ncol0);
        for (int c = 0; c < nchunk; ++c) {
            const bool more = c + 1 < nchunk;
            if (more) gload(row0, col0, (c + 1) * BK);
            else if (have_next) gload(nrow0, ncol0, 0);              // the next tile's first chunk rides under this chunk's MFMAs and the epilogue
            const f32x4 *ap = reinterpret_cast<const f32x4 *>(As[cur] + (wy * 64 + l31) * LD + lhi * (BK / 2));
            const f32x4 *bp = reinterpret_cast<const f32x4 *>(Bs[cur] + (wx * 64 + l31) * LD + lhi * (BK / 2));
#pragma unroll
            for (int gq = 0; gq < BK / 8; ++gq) {
                const f32x4 a0 = ap[gq], a1 = ap[gq + 32 * LD / 4], b0 = bp[gq], b1 = bp[gq + 32 * LD / 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
                }
            }
            if (more) { sstore(cur ^ 1); __syncthreads(); cur ^= 1; }
        }
        if (!NOEPI) {
            __syncthreads();                                          // everyone is done reading the panels: the level tile may overwrite them
            epilogue<BN + 8>(g, acc, row0, col0, wy, wx, l31, lhi, reinterpret_cast<uint16_t *>(smem), tid);
            __syncthreads();
        } else {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 12345.678f) g.D[tid] = sacc;
            __syncthreads();
        }
        if (!have_next) break;
        t = tn; row0 = nrow0; col0 = ncol0;
        cur = 0;
    }
}


// ---- V2: persistent, two blocks per CU, TWO accumulator sets: the epilogue of tile t (adds, f32 stores, levels into an LDS tile of its own) is
// spread over the 64 k-steps of tile t + 1 -- one accumulator value per k-step, between the MFMAs.  Kd = 128 only (8 chunks of 16, fully unrolled).
template <int EPI>      // 1 = full epilogue, 0 = K loop only
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void v2_kernel(Args g) {
    constexpr int BK = 16, LD = BK + 2, NE = BM * BK / 4 / 256, Q4 = BK / 4, QLD = BN + 8;
    __shared__ float smem[2 * BM * LD + 2 * BN * LD];
    __shared__ __attribute__((aligned(16))) uint16_t qtile[BM * QLD];
    float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(smem);
    float (*Bs)[BN * LD] = reinterpret_cast<float (*)[BN * LD]>(smem + 2 * BM * LD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int64_t panels_x = (g.row_tiles - xcd + 7) / 8;
    const int64_t tiles_x = panels_x * g.col_tiles;
    auto tile_x = [&](int64_t tx, int64_t &r0, int &c0) { r0 = ((tx / g.col_tiles) * 8 + xcd) * BM; c0 = (int)(tx % g.col_tiles) * BN; };
    int64_t t = bx;
    if (t >= tiles_x) return;
    int64_t row0; int col0;
    tile_x(t, row0, col0);
    float4 ra[NE], rb[NE];
    auto gload = [&](int64_t r0, int c0, int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;
            ra[i] = *reinterpret_cast<const float4 *>(g.A + (r0 + r) * g.Kd + k0 + 4 * q);
            rb[i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)(c0 + r) * g.Kd + k0 + 4 * q);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * 256, r = e / Q4, q = e % Q4;
            float *oa = As[buf] + r * LD + 4 * q, *ob = Bs[buf] + r * LD + 4 * q;
            oa[0] = ra[i].x; oa[1] = ra[i].y; oa[2] = ra[i].z; oa[3] = ra[i].w;
            ob[0] = rb[i].x * -2.f; ob[1] = rb[i].y * -2.f; ob[2] = rb[i].z * -2.f; ob[3] = rb[i].w * -2.f;
        }
    };
    f32x16 acc[2][2], old[2][2];
    // what the deferred epilogue needs from the PREVIOUS tile
    int64_t Dl[2] = {0, 0};                                               // element offsets into g.D (a pointer initialised with nullptr would become a FLAT address)
    float addp[2] = {0.f, 0.f};
    int64_t prow0 = 0; int pcol0 = 0;
    const int qbase = (wy * 64 + 4 * lhi) * QLD + wx * 64 + l31;
    gload(row0, col0, 0);
    sstore(0);
    __syncthreads();
    bool have_next = false;
    int64_t nrow0 = 0; int ncol0 = 0;
    auto tile_body = [&](auto prev_tag) {
        constexpr bool have_prev = decltype(prev_tag)::value;
        const int64_t tn = t + nbx;
        have_next = tn < tiles_x;
        nrow0 = row0; ncol0 = col0;
        if (have_next) tile_x(tn, nrow0, ncol0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cur = c & 1;
            if (c < 7) gload(row0, col0, (c + 1) * BK);
            else if (have_next) gload(nrow0, ncol0, 0);
            const float *ap = As[cur] + (wy * 64 + l31) * LD + lhi;
            const float *bp = Bs[cur] + (wx * 64 + l31) * LD + lhi;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = ap[kk], a1 = ap[32 * LD + kk], b0 = bp[kk], b1 = bp[32 * LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                if (EPI && have_prev) {                                   // one value of the previous tile per k-step
                    const int i = c * 8 + kk / 2;                         // compile time: 0..63
                    const int tj = i >> 5, ti = (i >> 4) & 1, r = i & 15;
                    const int ro = ti * 32 + (r & 3) + 8 * (r >> 2);
                    const float v = old[ti][tj][r] + addp[tj];
                    g.D[Dl[tj] + (int64_t)ro * SL] = v;
                    const float qf = rintf((v - g.lo) * g.inv);
                    qtile[qbase + tj * 32 + ro * QLD] = (uint16_t)(unsigned)__builtin_amdgcn_fmed3f(qf, 0.f, 65535.f);
                }
            }
            if (c < 7 || have_next) sstore(cur ^ 1);
            __syncthreads();
        }
        if (EPI && have_prev) {                                           // the level tile of the previous tile is complete (barriers above): 8 x 16-byte stores per thread
            const int plane = pcol0 / H, a0 = pcol0 % H;
#pragma unroll
            for (int i = 0; i < BM * BN / 8 / 256; ++i) {
                const int e = tid + i * 256, rl = e / (BN / 8), ch = e % (BN / 8);
                const int a = a0 + 8 * ch;
                *reinterpret_cast<uint4 *>(g.Dq + (int64_t)plane * (g.M * (int64_t)H) + ((int64_t)(a / SLQ) * g.M + (prow0 + rl)) * SLQ + (a % SLQ)) =
                    *reinterpret_cast<const uint4 *>(qtile + rl * QLD + 8 * ch);
            }
            __syncthreads();                                              // before the next tile's k-steps write levels again
        }
        // this tile becomes the previous one
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) old[i][j] = acc[i][j];
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int cc = col0 + wx * 64 + tj * 32 + l31;
            addp[tj] = g.sci[cc];
            Dl[tj] = d_off(g.M, row0 + wy * 64 + 4 * lhi, cc);
        }
        prow0 = row0; pcol0 = col0;
        t = tn; row0 = nrow0; col0 = ncol0;
    };
    tile_body(std::false_type{});
    while (have_next) tile_body(std::true_type{});
    if (EPI) {                                                            // the last tile: plain epilogue
        epilogue<QLD>(g, old, prow0, pcol0, wy, wx, l31, lhi, qtile, tid);
    } else {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += old[i][j][r];
        if (sacc == 12345.678f) g.D[tid] = sacc;
    }
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
    return ms / reps;
}

int main(int argc, char **argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 1000064;      // multiple of 128
    const int N = 2048, Kd = argc > 2 ? atoi(argv[2]) : 128;
    const int64_t Mr = (M / 128) * 128;
    Args g{};
    float *A, *B, *sci, *D, *Dref;
    uint16_t *Dq, *Dqref;
    CK(hipMalloc(&A, sizeof(float) * Mr * Kd)); CK(hipMalloc(&B, sizeof(float) * N * Kd)); CK(hipMalloc(&sci, sizeof(float) * N));
    CK(hipMalloc(&D, sizeof(float) * Mr * N)); CK(hipMalloc(&Dq, sizeof(uint16_t) * Mr * N));
    const int64_t rows_ref = 1024;
    std::vector<float> hA(Mr * Kd), hB((size_t)N * Kd), hs(N);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &v : hA) v = (float)(rnd() >> 56);
    for (auto &v : hB) v = (float)(rnd() >> 56) / 8.f;
    for (auto &v : hs) v = (float)(rnd() >> 44);
    CK(hipMemcpy(A, hA.data(), sizeof(float) * Mr * Kd, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), sizeof(float) * N * Kd, hipMemcpyHostToDevice));
    CK(hipMemcpy(sci, hs.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    g.A = A; g.B = B; g.sci = sci; g.D = D; g.Dq = Dq; g.M = Mr; g.N = N; g.Kd = Kd; g.lo = -1.3e6f; g.inv = 65535.f / 2.8e6f;      // every value inside the level range: the flag branch of the library-like leg is never taken
    g.row_tiles = Mr / BM; g.col_tiles = N / BN;
    // reference for the first rows_ref rows of a SEPARATE output with the same M (offsets depend on M)
    CK(hipMalloc(&Dref, sizeof(float) * Mr * N)); CK(hipMalloc(&Dqref, sizeof(uint16_t) * Mr * N));
    Args gr = g; gr.D = Dref; gr.Dq = Dqref;
    ref_kernel<<<(unsigned)((rows_ref * N + 255) / 256), 256>>>(gr, rows_ref);
    CK(hipDeviceSynchronize());
    std::vector<float> hd(rows_ref * SL), hr(rows_ref * SL);
    auto check = [&](const char *name) {
        // compare plane 3, slice 5 (a contiguous run of rows) and the u16 plane 2, slice 1 for the first rows_ref rows
        int64_t bad = 0;
        const int64_t off = 3ll * (Mr * H) + 5ll * (Mr * SL);
        CK(hipMemcpy(hd.data(), D + off, sizeof(float) * rows_ref * SL, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), Dref + off, sizeof(float) * rows_ref * SL, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < hd.size(); ++e) bad += memcmp(&hd[e], &hr[e], 4) != 0;
        std::vector<uint16_t> qd(rows_ref * SLQ), qr(rows_ref * SLQ);
        const int64_t qo = 2ll * (Mr * H) + 1ll * (Mr * SLQ);
        CK(hipMemcpy(qd.data(), Dq + qo, 2 * rows_ref * SLQ, hipMemcpyDeviceToHost));
        CK(hipMemcpy(qr.data(), Dqref + qo, 2 * rows_ref * SLQ, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < qd.size(); ++e) bad += qd[e] != qr[e];
        printf("   check %-28s %s (%lld mismatches)\n", name, bad ? "FAIL" : "ok", (long long)bad);
    };
    const double flop = 2.0 * Mr * N * Kd;
    const unsigned grid0 = (unsigned)(((g.row_tiles + 7) / 8) * 8 * g.col_tiles);
#define RUN(name, ...) { CK(hipMemset(D, 0xff, sizeof(float) * Mr * N)); float ms = timeit([&] { __VA_ARGS__; }, 5); \
        printf("%-40s %7.3f ms  %6.1f TF/s  %.2f of 157\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3); CK(hipGetLastError()); }
    RUN("v0 BK16 (library structure)", (v0_kernel<16, 0, 0><<<grid0, 256>>>(g))); check("v0");
    RUN("v0 BK16 fixed-trip unrolled", (v0_kernel<16, 1, 0><<<grid0, 256>>>(g))); check("v0u");
    RUN("v0 BK16 K loop only", (v0_kernel<16, 0, 1><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, stores into 1024 rows (L2)", (v0_kernel<16, 1, 2><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, f32 stores only", (v0_kernel<16, 1, 3><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, u16 levels only", (v0_kernel<16, 1, 4><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, f32 as 16-B stores after quad transpose + u16", (v0_kernel<16, 1, 5><<<grid0, 256>>>(g))); check("v0 transposed stores");
    RUN("v0 BK16 unrolled, f32 as 16-B stores only", (v0_kernel<16, 1, 6><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, library-like level arithmetic, dword stores", (v0_kernel<16, 1, 7><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, library-like level arithmetic, wide stores", (v0_kernel<16, 1, 8><<<grid0, 256>>>(g)));
    RUN("v0 BK16 unrolled, K loop only", (v0_kernel<16, 1, 1><<<grid0, 256>>>(g)));
    RUN("v0 BK32 unrolled", (v0_kernel<32, 1, 0><<<grid0, 256>>>(g))); check("v0 bk32");
#define V1(BPC, BKK) { char nm[80]; snprintf(nm, sizeof nm, "v1 persistent kperm BK%d, %d blocks/CU", BKK, BPC); \
        RUN(nm, (v1_kernel<BPC, 0, BKK><<<256u * BPC, 256>>>(g))); check(nm); RUN("   ... K loop only", (v1_kernel<BPC, 1, BKK><<<256u * BPC, 256>>>(g))); }
    V1(3, 16)
    RUN("v2 persistent, epilogue under the next tile", (v2_kernel<1><<<512, 256>>>(g))); check("v2");
    RUN("   ... K loop only", (v2_kernel<0><<<512, 256>>>(g)));
    return 0;
}
