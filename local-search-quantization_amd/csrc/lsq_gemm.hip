// lsq_gemm.hip -- chain-exact fp32 contraction on the gfx950 matrix cores.
//
// Serves the two table builds of the hot path:
//   get_unaries  (reference src/utils.jl:94-122):  U_j = (-2 C_j') X + ||c||^2
//   get_binaries (reference src/utils.jl:125-144): Bin = 2 C_i' C_j   (all ordered pairs)
//
// Numerics contract (oracle/lsq_oracle.c [build-defined 1]): every output element is the
// k-ASCENDING fmaf chain from +0 of its d products, then (optionally) one rounded add.
// v_mfma_f32_32x32x2_f32 computes exactly that chain (two k per instruction, k0 then k1, one
// rounding per product, no wider accumulation), so stepping k in ascending order through one
// accumulator reproduces the oracle bit for bit.  fp32 MFMA runs at the fp32 vector rate
// (157 TF peak) -- this is the right unit for an exact-f32 contraction; nothing is reshaped into
// a lower precision.
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128 rows x 128 cols, K chunks of BK staged in LDS
// (row stride BK+1 floats -> the per-k column reads are bank-conflict-free), double-buffered: the next chunk is loaded
// into registers while the MFMAs of the current one run, one barrier per chunk.  BK = 16 by default, and the kernel is
// compiled for FOUR resident blocks per CU (amdgpu_waves_per_eu(4,4): 120 VGPRs, no spills; three blocks at the compiler's
// own choice of 77 + 64 registers were 6 % slower -- a K = 128 tile is short, the prologue / epilogue of one block has to
// hide under the MFMAs of the others).  Measured on one box at 10^6 x 128: 5.57 ms (BK 8/16/32/64 at three blocks:
// 5.9/5.7/6.4/10.6 ms; single-buffered BK = 8: 6.7 ms; d = 960, 250 K vectors: 9.4 -> 8.0 ms).  Each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 VGPRs).  The MFMA M dimension carries the
// ROWS of A (vectors) and N the candidates, so for a fixed accumulator register a wave stores
// two 128-byte runs of consecutive candidates -- full-line writes of the 8 KB/vector unary rows.
// Block -> tile mapping is XCD-aware: the col tiles that share one 128-row A panel run on the
// same XCD (block b sits on XCD b % 8), so the panel is fetched from HBM once and re-served by
// that XCD's L2.
#include <stdlib.h>

#include "lsq_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128;
// LDS row stride = BK + LSQ_GEMM_PAD floats.  A wave's MFMA operand read touches rows l31 = 0..31 at k = kk + lhi (lhi = 0, 1): bank =
// (row * LD + lhi) mod 64.  LD = 17 maps rows 0..31 to 32 distinct banks but the lhi = 1 half lands on banks of the lhi = 0 half (17 r' + 1 ==
// 17 r mod 64 has solutions): the 25 % SQ_LDS_BANK_CONFLICT of profiles/r01h.  LD = 18: 18 r mod 64 = 2 (9 r mod 32) covers the even banks once,
// + lhi the odd ones: conflict-free for the reads (and 8-byte aligned rows for the staging stores).
#ifndef LSQ_GEMM_PAD
#define LSQ_GEMM_PAD 2
#endif

// One K chunk of a 128-row panel: global -> registers (tile_load), registers -> LDS (tile_store).  Split in two so
// that the loads of chunk c+1 are in flight while the MFMAs of chunk c run (register double buffering).
template <bool VEC4, int BK>
struct TileRegs { float4 v[BM * BK / 4 / 256]; };

template <bool VEC4, int BK>
__device__ inline void tile_load(const float *__restrict__ src, int64_t rows_total, int64_t row0, int Kd, int k0,
                                 TileRegs<VEC4, BK> &t, int tid, int64_t ld) {
    constexpr int Q4 = BK / 4, NE = BM * BK / 4 / 256;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int r = e / Q4, q = e % Q4;
        const int64_t gr = row0 + r;
        const int kk = k0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < rows_total) {
            const float *p = src + gr * ld + kk;
            if (VEC4 && kk + 3 < Kd) {
                v = *reinterpret_cast<const float4 *>(p);
            } else {
                if (kk + 0 < Kd) v.x = p[0];
                if (kk + 1 < Kd) v.y = p[1];
                if (kk + 2 < Kd) v.z = p[2];
                if (kk + 3 < Kd) v.w = p[3];
            }
        }
        t.v[i] = v;
    }
}

template <bool VEC4, int BK>
__device__ inline void tile_store(const TileRegs<VEC4, BK> &t, float scale, float *__restrict__ dst, int tid) {
    constexpr int LD = BK + LSQ_GEMM_PAD, Q4 = BK / 4, NE = BM * BK / 4 / 256;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + i * 256;
        const int r = e / Q4, q = e % Q4;
        float *o = dst + r * LD + 4 * q;
        o[0] = t.v[i].x * scale; o[1] = t.v[i].y * scale; o[2] = t.v[i].z * scale; o[3] = t.v[i].w * scale;   // scale is +-2 or 1: exact
    }
}

// Q16 = 1: the epilogue ALSO emits every value as a 16-bit fixed-point level q = rint((v - loU_j) * invD_j) of its column plane j = c / h
// (lsq_q16_params) into the slice-major u16 planes Dq -- the filter input of icm_walkq_kernel.  The range [loU, loU + 65535 D) comes from a
// SAMPLE of the rows (below), so a value may fall outside it: its level is clamped (meaningless), bit j of qflag[row] is raised and
// icm_walkq_kernel sends that vector's node j through the f32 path -- exactness never depends on the sample.
// Q16 = 2: range-only pass over a sample of the rows (every rts-th 128-row panel, contiguous reads): nothing is stored, the minimum / maximum of every column plane are
// accumulated in qrange[2 j], qrange[2 j + 1] as order-preserving uint keys.
template <bool VEC4, int BK, int Q16 = 0, bool FULLK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void chain_gemm_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                         const float *__restrict__ addv, float alpha, int64_t M, int N,
                                                         int Kd, int h, int64_t plane_stride, int64_t row_stride,
                                                         float *__restrict__ D, int64_t row_tiles, int col_tiles, int slice,
                                                         int64_t Mtot, int64_t rbase, uint16_t *__restrict__ Dq, int slice_q,
                                                         lsq_q16_params *__restrict__ qp, int64_t lda, unsigned short *__restrict__ qflag,
                                                         unsigned *__restrict__ qrange, int rts, const float *__restrict__ sigma,
                                                         const float *__restrict__ colshift, int stagger) {
    constexpr int LD = BK + LSQ_GEMM_PAD;
    __shared__ float smem[2 * BM * LD + 2 * BN * LD];      // A and B panels, double-buffered; reused by the u16 epilogue as a 128 x 128 level tile
    float (*As)[BM * LD] = reinterpret_cast<float (*)[BM * LD]>(smem);
    float (*Bs)[BN * LD] = reinterpret_cast<float (*)[BN * LD]>(smem + 2 * BM * LD);

    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t s = b >> 3;
    const int64_t rt = (s / col_tiles) * 8 + xcd;
    const int ct = (int)(s % col_tiles);
    if (rt >= row_tiles) return;
    // Phase stagger: all resident blocks start together and do identical work, so K loops (MFMA) and epilogues (12 KB of stores per thread-block row) of the
    // four blocks of a CU would stay in step -- the matrix cores idle while every block stores, HBM idles while every block multiplies.  The first
    // generation of blocks is delayed by a pseudo-random quarter of a tile time; the offsets persist because every tile takes the same time.
    if (stagger > 0 && b < 1024) {
        const unsigned slot = ((unsigned)b * 2654435761u) >> 30;
        for (unsigned w = 0; w < slot * (unsigned)stagger; ++w) __builtin_amdgcn_s_sleep(127);
    }
    const int64_t row0 = rt * BM * rts;                  // rts > 1 (range-only pass): every rts-th 128-row panel
    const int col0 = ct * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // per-(row, plane) shift of the value a LEVEL is taken of (argmin-neutral: lsq_icmq.hip); the f32 output stays unshifted.  The block's 128 shifts
    // (one column plane per block) wait in LDS: a global load in the epilogue would queue behind the epilogue's own stores (one vmcnt for both).
    // Q16 = 1 keeps them as lo_row = loU - sigma (one rounding, inside the bound's f32 term): the level is rint((v - lo_row) * invD), no extra add.
    __shared__ __attribute__((aligned(16))) float sgs[BM];
    if (Q16 != 0 && tid < BM) {
        const int64_t row = row0 + tid;
        const float sg = (sigma != nullptr && row < M) ? sigma[(rbase + row) * (N / h) + col0 / h] : 0.0f;
        sgs[tid] = (Q16 == 1) ? ((qp->ok != 0 ? qp->node[col0 / h].loU : 0.0f) - sg) : sg;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    TileRegs<VEC4, BK> ra, rb;
    tile_load<VEC4, BK>(A, M, row0, Kd, 0, ra, tid, lda);
    tile_load<VEC4, BK>(Bm, N, col0, Kd, 0, rb, tid, (int64_t)Kd);
    tile_store<VEC4, BK>(ra, 1.0f, As[0], tid);
    tile_store<VEC4, BK>(rb, alpha, Bs[0], tid);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < Kd; k0 += BK) {
        const bool more = k0 + BK < Kd;
        if (more) {                                              // next chunk travels HBM/L2 -> registers under this chunk's MFMAs
            tile_load<VEC4, BK>(A, M, row0, Kd, k0 + BK, ra, tid, lda);
            tile_load<VEC4, BK>(Bm, N, col0, Kd, k0 + BK, rb, tid, (int64_t)Kd);
        }
        // FULLK (Kd a multiple of BK: the launcher knows): fixed trip count, unrolled -- the operand reads of the later k-steps are issued under the MFMAs
        // of the earlier ones (the variable-trip loop waits for its four ds_reads in every step: tools/ubench_gemm.hip, 5.25 -> 5.03 ms per 10^6 x 128)
        const int kend = FULLK ? BK : ((Kd - k0 < BK) ? ((Kd - k0 + 1) & ~1) : BK);   // odd tail: one zero product appended
        const float *ap = As[cur] + (wy * 64 + l31) * LD + lhi;
        const float *bp = Bs[cur] + (wx * 64 + l31) * LD + lhi;
        auto kstep = [&](int kk) {                              // ascending k through one accumulator: the oracle's fmaf chain
            const float a0 = ap[kk], a1 = ap[32 * LD + kk];
            const float b0 = bp[kk], b1 = bp[32 * LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        };
        if constexpr (FULLK) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) kstep(kk);
        } else {
            for (int kk = 0; kk < kend; kk += 2) kstep(kk);
        }
        if (more) {                                              // the other buffer was last read one iteration ago (barrier below)
            tile_store<VEC4, BK>(ra, 1.0f, As[cur ^ 1], tid);
            tile_store<VEC4, BK>(rb, alpha, Bs[cur ^ 1], tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    float bmin = __builtin_inff(), bmax = -__builtin_inff();      // range-only pass: the block tile lies in ONE column plane (h is a multiple of 128)
    bool bnan = false;
    constexpr int QLD = BN + 8;                                  // u16 level tile: 128 rows x (128 + 8) columns = 34 816 B <= the panels' 36 864 B
    static_assert(Q16 != 1 || BM * QLD * 2 <= (int)sizeof(float) * (2 * BM * LD + 2 * BN * LD), "level tile must fit the panel storage");
    uint16_t *qtile = reinterpret_cast<uint16_t *>(smem);        // free: the K loop ended with a barrier
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int c = col0 + wx * 64 + tj * 32 + l31;
        if (c >= N) continue;
        const float add = addv ? addv[c] : 0.0f;
        const float gc = (Q16 != 0 && colshift) ? colshift[c] : 0.0f;      // per-candidate shift of the LEVEL input (double-centred tables: lsq_icmq.hip)
        // row-major planes: off = (c/h)*plane + (c%h) + r*row_stride
        // slice-major planes (slice = SL > 0): off = (c/h)*plane + ((c%h)/SL)*(Mtot*SL) + (c%h)%SL + r*SL
        // (r counts from rbase: the launch may cover rows [rbase, rbase + M) of a Mtot-row output)
        const int a = c % h;
        const int64_t coff = slice ? (int64_t)(c / h) * plane_stride + (int64_t)(a / slice) * (Mtot * slice) + (a % slice)
                                   : (int64_t)(c / h) * plane_stride + a;
        const int64_t rstride = slice ? (int64_t)slice : row_stride;
        float qinv = 0.f, qhi = 0.f;
        int noor = 0;
        const bool q16 = Q16 == 1 && qp->ok != 0;          // unusable bounds (non-finite data): the f32 walk handles the chunk, nothing to emit
        int npair = 0;                                       // (row, plane) pairs this lane flagged FIRST
        if (q16) {
            qinv = qp->node[c / h].invD;
            qhi = qp->node[c / h].hiq;
        }
        float vmin = __builtin_inff(), vmax = -__builtin_inff();
        // Lean epilogue (round 3: with 64 accumulators per thread the old per-element address arithmetic -- 64-bit row offsets, bound checks, tile
        // indices -- cost ~25 VALU instructions per value, 1.3 ms of issue time per 10^6 vectors): everything lane-dependent is folded into one
        // pointer / one LDS index per (thread, tj); what is left per element is a wave-uniform offset (scalar unit) or a compile-time constant.
        const bool full = row0 + BM <= M;                                   // block-uniform: no per-element row check
        const int64_t lrow = row0 + wy * 64 + 4 * lhi;                      // the lane's first row; element (ti, r) sits (ti*32 + (r&3) + 8*(r>>2)) rows further
        float *__restrict__ Dl = (Q16 == 2) ? nullptr : D + coff + (rbase + lrow) * rstride;
        uint16_t *__restrict__ ql = qtile + (wy * 64 + 4 * lhi) * QLD + wx * 64 + tj * 32 + l31;
        const float *__restrict__ sgl = sgs + wy * 64 + 4 * lhi;
        f32x4 sg4 = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ti * 32 + (r & 3) + 8 * (r >> 2);            // compile-time
                if ((r & 3) == 0 && Q16 != 0) sg4 = *reinterpret_cast<const f32x4 *>(sgl + ro);      // the shifts of rows ro .. ro + 3: one 16-byte LDS read
                const float sg = (r & 3) == 0 ? sg4.x : ((r & 3) == 1 ? sg4.y : ((r & 3) == 2 ? sg4.z : sg4.w));
                if (full || lrow + ro < M) {
                    float v = acc[ti][tj][r];
                    if (addv) v = v + add;           // one rounded add (utils.jl:112-118)
                    if (Q16 == 2) {
                        const float w = (v + gc) + sg;
                        vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
                        continue;
                    }
                    Dl[(int64_t)ro * rstride] = v;
                    if (q16) {
                        const float qf = rintf(((v + gc) - sg) * qinv);
                        const float qc = __builtin_amdgcn_fmed3f(qf, 0.0f, qhi);       // clamp to the level range (NaN -> 0)
                        if (!(qf == qc)) {                                               // outside [0, hiq] or NaN: flag the (vector, node) pair
                            const int64_t row = lrow + ro;
                            const unsigned bit = (1u << (c / h)) << (16 * (int)((rbase + row) & 1));
                            const unsigned old = atomicOr(reinterpret_cast<unsigned *>(qflag + ((rbase + row) & ~(int64_t)1)), bit);
                            npair += (old & bit) ? 0 : 1;
                            ++noor;
                        }
                        // levels go through an LDS tile so that they leave the chip as 16-byte stores (8 candidates of a row), not 2-byte ones
                        ql[ro * QLD] = (uint16_t)(unsigned)qc;
                    }
                }
            }
        }
        if (q16 && noor) atomicAdd(&qp->oor, noor);
        if (q16 && npair) atomicAdd(&qp->nflag, npair);
        if (Q16 == 2) { bmin = fminf(bmin, vmin); bmax = fmaxf(bmax, vmax); bnan = bnan || !(vmin == vmin && vmax == vmax); }
    }
    if (Q16 == 1) {
        if (qp->ok != 0) {                                 // block-uniform
            __syncthreads();
            const int plane = col0 / h, a0 = col0 % h;
#pragma unroll
            for (int i = 0; i < BM * BN / 8 / 256; ++i) {
                const int e = tid + i * 256, rl = e / (BN / 8), ch = e % (BN / 8);
                const int64_t row = row0 + rl;
                const int a = a0 + 8 * ch;
                if (row < M && col0 + 8 * ch < N)
                    *reinterpret_cast<uint4 *>(Dq + (int64_t)plane * (Mtot * (int64_t)h) + ((int64_t)(a / slice_q) * Mtot + (rbase + row)) * slice_q + (a % slice_q)) =
                        *reinterpret_cast<const uint4 *>(qtile + rl * QLD + 8 * ch);
            }
        }
    }
    if (Q16 == 2) {                                        // one pair of atomics per block (not per wave: they all hit the plane's two words)
        __shared__ float rmin[4], rmax[4];
        __shared__ int rnan[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { bmin = fminf(bmin, __shfl_xor(bmin, off, 64)); bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64)); }
        const bool wnan = __ballot(bnan) != 0ull;
        if (lane == 0) { rmin[wave] = bmin; rmax[wave] = bmax; rnan[wave] = wnan ? 1 : 0; }
        __syncthreads();
        if (tid == 0) {
            const float lo4 = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3])), hi4 = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
            const int plane = col0 / h;
            if (rnan[0] | rnan[1] | rnan[2] | rnan[3]) atomicOr(&qrange[2 * LSQ_MAX_M], 1u);      // a non-finite value in the sample
            else if (lo4 <= hi4) {
                const unsigned bl = __float_as_uint(lo4), bh = __float_as_uint(hi4);
                atomicMin(&qrange[2 * plane], bl ^ ((unsigned)((int)bl >> 31) | 0x80000000u));
                atomicMax(&qrange[2 * plane + 1], bh ^ ((unsigned)((int)bh >> 31) | 0x80000000u));
            }
        }
    }
}

__global__ __launch_bounds__(64) void sqnorms_kernel(const float *__restrict__ Kb, int rows, int d, float *__restrict__ sci) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = Kb + (int64_t)r * d;
    float acc = 0.0f;
    // the chain itself is canonical (k ascending, one fmaf per term: oracle parity); only its operands are fetched ahead, 16 bytes at a time
    if ((d & 3) == 0 && (((uintptr_t)Kb) & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *p4 = reinterpret_cast<const f4 *>(p);
        int t = 0;
        for (; t + 8 <= d / 4; t += 8) {
            f4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p4[t + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc = fmaf(v[e].x, v[e].x, acc); acc = fmaf(v[e].y, v[e].y, acc); acc = fmaf(v[e].z, v[e].z, acc); acc = fmaf(v[e].w, v[e].w, acc); }
        }
        for (; t < d / 4; ++t) { const f4 v = p4[t]; acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc); }
    } else {
        for (int t = 0; t < d; ++t) acc = fmaf(p[t], p[t], acc);
    }
    sci[r] = acc;
}

}  // namespace

int lsq_launch_chain_gemm(hipStream_t s, const float *A, const float *Bm, const float *addv, float alpha, int64_t M,
                          int N, int Kd, int h, int64_t plane_stride, int64_t row_stride, float *D, int slice, int64_t Mtot, int64_t rbase,
                          uint16_t *Dq, int slice_q, lsq_q16_params *qp, int64_t lda, unsigned short *qflag, unsigned *qrange, int rts, const float *sigma, const float *colshift) {
    if (lda <= 0) lda = Kd;
    if (rts < 1) rts = 1;
    if (M <= 0 || N <= 0) return LSQ_OK;
    const int64_t row_tiles = (M + (int64_t)BM * rts - 1) / ((int64_t)BM * rts);
    const int col_tiles = (N + BN - 1) / BN;
    const int64_t blocks = ((row_tiles + 7) / 8) * 8 * col_tiles;
    if (blocks > 0x7fffffffLL) { lsq_set_error("chain_gemm: grid too large"); return LSQ_EINVAL; }
    const bool vec4 = (Kd % 4 == 0) && (((uintptr_t)A | (uintptr_t)Bm) % 16 == 0);
    const int bk = LSQ_KNOB("LSQ_GEMM_BK", 16);
    const bool fullk = Kd % 16 == 0;
    const int stagger = (M >= 65536) ? LSQ_KNOB("LSQ_GEMM_STAGGER", 0) : 0;      // units of s_sleep(127) = 8128 clocks per quarter
    // K chunks of 8 or 16 only: both fit four resident blocks per CU (the kernel is compiled for 4 waves per SIMD)
    if (qrange) {          // range-only pass
        if (vec4 && lda % 4 == 0)
            if (fullk) hipLaunchKernelGGL((chain_gemm_kernel<true, 16, 2, true>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, qrange, rts, sigma, colshift, 0);
            else hipLaunchKernelGGL((chain_gemm_kernel<true, 16, 2>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, qrange, rts, sigma, colshift, 0);
        else
            hipLaunchKernelGGL((chain_gemm_kernel<false, 16, 2>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, qrange, rts, sigma, colshift, 0);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    if (Dq) {
        if (!qp || !qflag || slice_q < 1) { lsq_set_error("chain_gemm: quantised output needs parameters"); return LSQ_EINVAL; }
        if (vec4 && fullk)
            hipLaunchKernelGGL((chain_gemm_kernel<true, 16, 1, true>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, Dq, slice_q, qp, lda, qflag, nullptr, 1, sigma, colshift, stagger);
        else if (vec4)
            hipLaunchKernelGGL((chain_gemm_kernel<true, 16, 1>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, Dq, slice_q, qp, lda, qflag, nullptr, 1, sigma, colshift, stagger);
        else
            hipLaunchKernelGGL((chain_gemm_kernel<false, 16, 1>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                               plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, Dq, slice_q, qp, lda, qflag, nullptr, 1, sigma, colshift, stagger);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    }
    if (vec4 && bk == 8)
        hipLaunchKernelGGL((chain_gemm_kernel<true, 8>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                           plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, nullptr, 1, nullptr, nullptr, stagger);
    else if (vec4 && fullk)
        hipLaunchKernelGGL((chain_gemm_kernel<true, 16, 0, true>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                           plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, nullptr, 1, nullptr, nullptr, stagger);
    else if (vec4)
        hipLaunchKernelGGL((chain_gemm_kernel<true, 16>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                           plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, nullptr, 1, nullptr, nullptr, stagger);
    else
        hipLaunchKernelGGL((chain_gemm_kernel<false, 16>), dim3((unsigned)blocks), dim3(256), 0, s, A, Bm, addv, alpha, M, N, Kd, h,
                           plane_stride, row_stride, D, row_tiles, col_tiles, slice, Mtot, rbase, nullptr, 0, nullptr, lda, nullptr, nullptr, 1, nullptr, nullptr, stagger);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

const void *lsq_probe_kernel_gemm() { return reinterpret_cast<const void *>(&sqnorms_kernel); }

int lsq_launch_sqnorms(hipStream_t s, const float *Kb, int rows, int d, float *sci) {
    if (rows <= 0) return LSQ_OK;
    hipLaunchKernelGGL(sqnorms_kernel, dim3((rows + 15) / 16), dim3(16), 0, s, Kb, rows, d, sci);      // one row per thread, a long dependent chain: spread over the CUs
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
