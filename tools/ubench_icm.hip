// tools/ubench_icm.hip -- micro-benchmark for the ICM node-update kernel structure (tuning aid, not
// part of the product).  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off \
//                                   tools/ubench_icm.hip -o gpurun_out/ubench_icm
// Each variant processes n vectors of one node-j launch: U_j stream (1 KiB/vector) + (M-1) gathered
// 1 KiB columns of block-row j + argmin.  All variants must produce identical codes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define H 256
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_self(float v) {
    const int iv = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, CTRL, ROW_MASK, 0xf, false));
}
__device__ inline float wave_min_builtin(float v) {
    v = fminf(v, dpp_self<0xB1, 0xf>(v));
    v = fminf(v, dpp_self<0x4E, 0xf>(v));
    v = fminf(v, dpp_self<0x141, 0xf>(v));
    v = fminf(v, dpp_self<0x140, 0xf>(v));
    v = fminf(v, dpp_self<0x142, 0xa>(v));
    v = fminf(v, dpp_self<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline float wave_min_asm(float v) {
    asm volatile(
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <bool ASM>
__device__ inline int first_argmin(f32x4 s) {
    const float lm = fminf(fminf(s.x, s.y), fminf(s.z, s.w));
    const float wm = ASM ? wave_min_asm(lm) : wave_min_builtin(lm);
    const int inl = (s.x == wm) ? 0 : (s.y == wm) ? 1 : (s.z == wm) ? 2 : 3;
    const uint64_t mask = __ballot(lm == wm);
    int best = 0;
    if (mask != 0) {
        const int L = __builtin_ctzll(mask);
        best = 4 * L + __builtin_amdgcn_readlane(inl, L);
    }
    const float s0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s.x)));
    if (s0 != s0) best = 0;
    return best;
}

__device__ inline uint64_t rfl64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <int M>
__device__ inline void issue(const float *__restrict__ Uj, const float *__restrict__ Tj, int64_t i, uint64_t rec, int j, int lane,
                             f32x4 &u, f32x4 (&c)[M - 1], bool nt) {
    const f32x4 *up = reinterpret_cast<const f32x4 *>(Uj + i * H) + lane;
    u = nt ? __builtin_nontemporal_load(up) : *up;
#pragma unroll
    for (int kk = 0; kk < M - 1; ++kk) {
        const int k = kk + (kk >= j ? 1 : 0);
        const uint32_t code = (uint32_t)(rec >> (8 * k)) & 0xffu;
        c[kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * H) + code) * H)[lane];
    }
}

// MODE 0: baseline (vector rec load from the in-place buffer, byte store)
// MODE 1: ping-pong records (const in -> scalar loads), 8-byte store
// MODE 2: MODE 1 + next record prefetched one iteration ahead
// MODE 3: MODE 2 + next vector's loads issued before the current vector's compute (2-deep pipeline)
// ABL 0 full, 1 skip gathers (U stream only), 2 skip U stream (gathers only), 3 skip argmin
template <int M, int MODE, bool NT, bool ASM, int ABL>
__global__ __launch_bounds__(256) void node_kernel(const float *__restrict__ Uj, const float *__restrict__ Tj,
                                                   const uint64_t *__restrict__ rec_in, uint64_t *__restrict__ rec_out,
                                                   int64_t n, int j) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (MODE <= 2) {
        uint64_t rnext = 0;
        if (MODE == 2 && i < n) rnext = rec_in[i];
        for (; i < n; i += nw) {
            uint64_t rec;
            if (MODE == 0) rec = rfl64(reinterpret_cast<const volatile uint64_t *>(rec_out)[i]);
            else if (MODE == 1) rec = rec_in[i];
            else { rec = rnext; if (i + nw < n) rnext = rec_in[i + nw]; }
            f32x4 u, c[M - 1];
            if (ABL == 1) {
                const f32x4 *up = reinterpret_cast<const f32x4 *>(Uj + i * H) + lane;
                u = NT ? __builtin_nontemporal_load(up) : *up;
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) c[kk] = (f32x4){(float)(rec & 1), 0.f, 0.f, 0.f};
            } else {
                issue<M>(Uj, Tj, i, rec, j, lane, u, c, NT);
                if (ABL == 2) u = (f32x4){1.f, 2.f, 3.f, 4.f};   // the U load result is dead -> compiler drops it
            }
            f32x4 s = u;
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];
            int best;
            if (ABL == 3) best = (int)(__float_as_int(s.x + s.y + s.z + s.w) & 0xff);
            else best = first_argmin<ASM>(s);
            if (ABL == 3) best = __builtin_amdgcn_readfirstlane(best);
            if (lane == 0) {
                if (MODE == 0) reinterpret_cast<uint8_t *>(rec_out)[i * 8 + j] = (uint8_t)best;
                else rec_out[i] = (rec & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
            }
        }
    } else {
        if (i >= n) return;
        uint64_t recA = rec_in[i], recB = (i + nw < n) ? rec_in[i + nw] : 0;
        f32x4 uA, cA[M - 1], uB, cB[M - 1];
        issue<M>(Uj, Tj, i, recA, j, lane, uA, cA, NT);
        for (;;) {
            // ---- A is current, B is next
            const int64_t ib = i + nw;
            uint64_t recC = 0;
            if (ib < n) { issue<M>(Uj, Tj, ib, recB, j, lane, uB, cB, NT); if (ib + nw < n) recC = rec_in[ib + nw]; }
            {
                f32x4 s = uA;
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) s = s + cA[kk];
                const int best = first_argmin<ASM>(s);
                if (lane == 0) rec_out[i] = (recA & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
            }
            if (ib >= n) break;
            // ---- B is current, A is next
            const int64_t ic = ib + nw;
            uint64_t recD = 0;
            if (ic < n) { issue<M>(Uj, Tj, ic, recC, j, lane, uA, cA, NT); if (ic + nw < n) recD = rec_in[ic + nw]; }
            {
                f32x4 s = uB;
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) s = s + cB[kk];
                const int best = first_argmin<ASM>(s);
                if (lane == 0) rec_out[ib] = (recB & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
            }
            if (ic >= n) break;
            i = ic; recA = recC; recB = recD;
        }
    }
}


// ---- role-split experiments: wave 0 of each block only streams U rows (the HBM traffic), the other
// waves only gather (L2 traffic).  No hand-off: this measures whether the two overlap when decoupled.
// LOADER 0: global_load_lds into an LDS ring (no VGPR destination), 1: plain loads into VGPRs.
template <int M, int WAVES, int LOADER, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void split_kernel(const float *__restrict__ Uj, const float *__restrict__ Tj,
                                                           const uint64_t *__restrict__ rec_in, uint64_t *__restrict__ rec_out,
                                                           int64_t n, int j) {
    __shared__ f32x4 ring[DEPTH * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int CONS = WAVES - 1;
    // block b owns vectors [b*per, (b+1)*per)
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = (lo + per < n) ? lo + per : n;
    if (wave == 0) {
        float sink = 0.f;
        for (int64_t i = lo; i < hi; i += DEPTH) {
            if (LOADER == 0) {
#pragma unroll
                for (int q = 0; q < DEPTH; ++q)
                    if (i + q < hi)
                        __builtin_amdgcn_global_load_lds(reinterpret_cast<const f32x4 *>(Uj + (i + q) * H) + lane,
                                                         (__attribute__((address_space(3))) void *)(ring + q * 64), 16, 0, 0);
                __builtin_amdgcn_s_waitcnt(0);
            } else {
                f32x4 r[DEPTH];
#pragma unroll
                for (int q = 0; q < DEPTH; ++q)
                    r[q] = (i + q < hi) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Uj + (i + q) * H) + lane) : (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) sink += r[q].x + r[q].y + r[q].z + r[q].w;
            }
        }
        if (LOADER == 1 && sink == 123.456f) rec_out[0] = 0;
    } else {
        uint64_t rnext = (lo + wave - 1 < hi) ? rec_in[lo + wave - 1] : 0;
        for (int64_t i = lo + wave - 1; i < hi; i += CONS) {
            const uint64_t rec = rnext;
            if (i + CONS < hi) rnext = rec_in[i + CONS];
            f32x4 c[M - 1];
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) {
                const int k = kk + (kk >= j ? 1 : 0);
                const uint32_t code = (uint32_t)(rec >> (8 * k)) & 0xffu;
                c[kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * H) + code) * H)[lane];
            }
            f32x4 s = (f32x4){1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];
            const int best = first_argmin<true>(s);
            if (lane == 0) rec_out[i] = (rec & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
        }
    }
}
#define MAKE_SPLIT(NAME, WAVES, LOADER, DEPTH)                                                                                     \
    static void launch_##NAME(int blocks, hipStream_t s, const float *U, const float *T, const uint64_t *ri, uint64_t *ro, int64_t n, int j) { \
        hipLaunchKernelGGL((split_kernel<8, WAVES, LOADER, DEPTH>), dim3(blocks), dim3(WAVES * 64), 0, s, U, T, ri, ro, n, j);      \
    }
MAKE_SPLIT(split8_lds16, 8, 0, 16)
MAKE_SPLIT(split8_vgpr8, 8, 1, 8)
MAKE_SPLIT(split16_lds16, 16, 0, 16)
MAKE_SPLIT(split16_lds32, 16, 0, 32)
MAKE_SPLIT(split4_lds16, 4, 0, 16)


// ---- E3: whole-CU role split.  One 1024-thread block per CU (LDS-padded), NPF of the 32 blocks of
// every XCD only stream U rows (HBM misses on THEIR L1), the rest only gather.  CONSUME_U: consumers
// also read their U row (an L2 hit if the prefetcher ran ahead, else a miss).
template <int M, int NPF, bool CONSUME_U>
__global__ __launch_bounds__(1024) void cu_split_kernel(const float *__restrict__ Uj, const float *__restrict__ Tj,
                                                        const uint64_t *__restrict__ rec_in, uint64_t *__restrict__ rec_out,
                                                        int64_t n, int j, unsigned *__restrict__ progress) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;          // block b sits on XCD b % 8 (observed)
    const int64_t per = (n + 7) / 8;
    const int64_t lo = xcd * per, hi = (lo + per < n) ? lo + per : n;
    constexpr int NC = 32 - NPF;
    if (threadIdx.x == 9999) pad[0] = 1.f;
    if (slot < NPF) {
        float sink = 0.f;
        const int64_t stride = NPF * 16;
        for (int64_t i = lo + slot * 16 + wave; i < hi; i += stride * 4) {
            f32x4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t ii = i + q * stride;
                r[q] = (ii < hi) ? *(reinterpret_cast<const f32x4 *>(Uj + ii * H) + lane) : (f32x4){0, 0, 0, 0};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) sink += r[q].x + r[q].y + r[q].z + r[q].w;
        }
        if (sink == 123.456f) rec_out[0] = 0;
    } else {
        const int cw = (slot - NPF) * 16 + wave;
        const int64_t stride = NC * 16;
        uint64_t rnext = (lo + cw < hi) ? rec_in[lo + cw] : 0;
        for (int64_t i = lo + cw; i < hi; i += stride) {
            const uint64_t rec = rnext;
            if (i + stride < hi) rnext = rec_in[i + stride];
            f32x4 c[M - 1];
            f32x4 s = (f32x4){1.f, 2.f, 3.f, 4.f};
            if (CONSUME_U) s = *(reinterpret_cast<const f32x4 *>(Uj + i * H) + lane);
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) {
                const int k = kk + (kk >= j ? 1 : 0);
                const uint32_t code = (uint32_t)(rec >> (8 * k)) & 0xffu;
                c[kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * H) + code) * H)[lane];
            }
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];
            const int best = first_argmin<true>(s);
            if (lane == 0) rec_out[i] = (rec & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
        }
    }
}
template <int NPF, bool CU>
static void launch_cusplit(hipStream_t s, const float *U, const float *T, const uint64_t *ri, uint64_t *ro, int64_t n, int j) {
    hipLaunchKernelGGL((cu_split_kernel<8, NPF, CU>), dim3(256), dim3(1024), 100 * 1024, s, U, T, ri, ro, n, j, (unsigned *)nullptr);
}


// ---- E4: prefetch the NEXT vector's U row into L2 one iteration ahead, (PF=1) through the SCALAR
// data path (8 x s_load_dword, one per 128-B line: misses are tracked by the scalar cache, not by the
// vector L1 whose in-order return FIFO would stall the gathers), or (PF=2) by a vector touch load
// with 8 active lanes.  The demand load of U then hits L2.
template <int M, int PF>
__global__ __launch_bounds__(256) void pf_kernel(const float *__restrict__ Uj, const float *__restrict__ Tj,
                                                 const uint64_t *__restrict__ rec_in, uint64_t *__restrict__ rec_out,
                                                 int64_t n, int j) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    uint64_t rnext = (i < n) ? rec_in[i] : 0;
    float sinkv = 0.f;
    for (; i < n; i += nw) {
        const uint64_t rec = rnext;
        const int64_t inx = (i + nw < n) ? i + nw : i;
        rnext = rec_in[inx];
        unsigned t0, t1, t2, t3, t4, t5, t6, t7;
        const float *pn = Uj + inx * H;
        if (PF == 1) {
            asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x80\n\ts_load_dword %2, %8, 0x100\n\ts_load_dword %3, %8, 0x180\n\t"
                         "s_load_dword %4, %8, 0x200\n\ts_load_dword %5, %8, 0x280\n\ts_load_dword %6, %8, 0x300\n\ts_load_dword %7, %8, 0x380"
                         : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7) : "s"(pn) : "memory");
        }
        float tv = 0.f;
        if (PF == 2 && lane < 8) tv = __builtin_nontemporal_load(pn + lane * 32);
        f32x4 u, c[M - 1];
        issue<M>(Uj, Tj, i, rec, j, lane, u, c, false);
        f32x4 s = u;
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];
        const int best = first_argmin<true>(s);
        if (lane == 0) rec_out[i] = (rec & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
        if (PF == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7) : "memory");
        if (PF == 2) sinkv += tv;
    }
    if (PF == 2 && sinkv == 123.456f) rec_out[0] = 0;
}
template <int PF>
static void launch_pf(int blocks, hipStream_t s, const float *U, const float *T, const uint64_t *ri, uint64_t *ro, int64_t n, int j) {
    hipLaunchKernelGGL((pf_kernel<8, PF>), dim3(blocks), dim3(256), 0, s, U, T, ri, ro, n, j);
}


// ---- E5: LDS-slice design.  Block = (slice of SL=16 candidates) x (vector range).  LDS holds
// T_j[k][b][a0..a0+16) for all k != j, b (7*256*64 B = 112 KiB at m=8).  U is stored slice-major
// Us[slice][i][16] so a wave load (lane = 4*v + q -> vector v of 16, candidates 4q..4q+3) is 1 KiB
// contiguous.  Output: partial (min value, local index) per (slice, vector); a combine kernel
// picks the lowest-index global minimum.
template <int M>
__global__ __launch_bounds__(1024) void slice_kernel(const float *__restrict__ Us, const float *__restrict__ Tj,
                                                     const uint64_t *__restrict__ rec_in, float2 *__restrict__ part,
                                                     int64_t n, int j, int nranges) {
    extern __shared__ f32x4 lds[];                       // [(M-1)*256][4] f32x4  (16 floats per (k,b))
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slice = blockIdx.x & 15, range = blockIdx.x >> 4;
    // stage the slice of block-row j: entry e = kk*256 + b  <-  T[j][k][b][16*slice .. +16)
    for (int e = threadIdx.x; e < (M - 1) * 256 * 4; e += 1024) {
        const int q = e & 3, eb = e >> 2, kk = eb >> 8, b = eb & 255;
        const int k = kk + (kk >= j ? 1 : 0);
        lds[e] = *reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * H) + b) * H + slice * 16 + q * 4);
    }
    __syncthreads();
    const int64_t per = (n + nranges - 1) / nranges;
    const int64_t lo = range * per, hi = (lo + per < n) ? lo + per : n;
    const int v = lane >> 2, q = lane & 3;
    const float *Ub = Us + (int64_t)slice * n * 16;
    // wave handles 16 vectors per iteration; 16 waves -> 256 vectors per block iteration
    int64_t i0 = lo + wave * 16;
    const int64_t step = 16 * 16;
    f32x4 un = (i0 + v < hi) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Ub + (i0 + v) * 16) + q) : (f32x4){0, 0, 0, 0};
    uint64_t rn = (i0 + v < hi) ? rec_in[i0 + v] : 0;
    for (; i0 < hi; i0 += step) {
        f32x4 s = un;
        const uint64_t rec = rn;
        const int64_t i1 = i0 + step;
        if (i1 + v < hi) {
            un = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Ub + (i1 + v) * 16) + q);
            rn = rec_in[i1 + v];
        }
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) {
            const int k = kk + (kk >= j ? 1 : 0);
            const uint32_t code = (uint32_t)(rec >> (8 * k)) & 0xffu;
            s = s + lds[(kk * 256 + code) * 4 + q];
        }
        // min over the 16 candidates of this vector: in-lane 4, then the quad
        float lm = fminf(fminf(s.x, s.y), fminf(s.z, s.w));
        int li = (s.x == lm) ? 0 : (s.y == lm) ? 1 : (s.z == lm) ? 2 : 3;
        li += 4 * q;
        // quad reduce (value, index): lower index wins ties
        {
            float ov = dpp_self<0xB1, 0xf>(lm); int oi = __builtin_amdgcn_update_dpp(li, li, 0xB1, 0xf, 0xf, false);
            if (ov < lm || (ov == lm && oi < li)) { lm = ov; li = oi; }
            ov = dpp_self<0x4E, 0xf>(lm); oi = __builtin_amdgcn_update_dpp(li, li, 0x4E, 0xf, 0xf, false);
            if (ov < lm || (ov == lm && oi < li)) { lm = ov; li = oi; }
        }
        if (q == 0 && i0 + v < hi) part[(int64_t)slice * n + i0 + v] = make_float2(lm, __int_as_float(li));
    }
}
__global__ __launch_bounds__(256) void combine_kernel(const float2 *__restrict__ part, const uint64_t *__restrict__ rec_in,
                                                      uint64_t *__restrict__ rec_out, int64_t n, int j) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float best = part[i].x; int bi = __float_as_int(part[i].y);
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
        const float2 p = part[(int64_t)sl * n + i];
        if (p.x < best) { best = p.x; bi = 16 * sl + __float_as_int(p.y); }
    }
    const uint64_t rec = rec_in[i];
    rec_out[i] = (rec & ~(0xffull << (8 * j))) | ((uint64_t)bi << (8 * j));
}
struct Variant { const char *name; void (*launch)(int blocks, hipStream_t, const float *, const float *, const uint64_t *, uint64_t *, int64_t, int); bool inplace; };

#define MAKE(NAME, M, MODE, NT, ASM, ABL)                                                                                          \
    static void launch_##NAME(int blocks, hipStream_t s, const float *U, const float *T, const uint64_t *ri, uint64_t *ro, int64_t n, int j) { \
        hipLaunchKernelGGL((node_kernel<M, MODE, NT, ASM, ABL>), dim3(blocks), dim3(256), 0, s, U, T, ri, ro, n, j);                \
    }

MAKE(m8_base, 8, 0, true, false, 0)
MAKE(m8_pp, 8, 1, true, false, 0)
MAKE(m8_pp_pref, 8, 2, true, false, 0)
MAKE(m8_pipe, 8, 3, true, false, 0)
MAKE(m8_pipe_asm, 8, 3, true, true, 0)
MAKE(m8_pref_asm, 8, 2, true, true, 0)
MAKE(m8_pipe_asm_nont, 8, 3, false, true, 0)
MAKE(m8_pref_Uonly, 8, 2, true, true, 1)
MAKE(m8_pref_Gonly, 8, 2, true, true, 2)
MAKE(m8_pref_noargmin, 8, 2, true, true, 3)
MAKE(m16_pref_asm, 16, 2, true, true, 0)
MAKE(m16_pipe_asm, 16, 3, true, true, 0)

int main(int argc, char **argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int j = 3;
    printf("n=%lld reps=%d\n", (long long)n, reps);
    float *U, *T;
    uint64_t *recA, *recB, *ref;
    const size_t tbytes = sizeof(float) * 16 * H * H;
    CK(hipMalloc(&U, sizeof(float) * n * H));
    CK(hipMalloc(&T, tbytes));
    CK(hipMalloc(&recA, 16 * n)); CK(hipMalloc(&recB, 16 * n)); CK(hipMalloc(&ref, 16 * n));
    std::vector<float> hU((size_t)n * H), hT((size_t)16 * H * H);
    std::vector<uint64_t> hrec((size_t)n);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (auto &v : hU) v = (float)(rnd() % 100000) * 0.01f;
    for (auto &v : hT) v = (float)(rnd() % 100000) * 0.001f;
    for (auto &v : hrec) v = rnd();
    CK(hipMemcpy(U, hU.data(), sizeof(float) * n * H, hipMemcpyHostToDevice));
    CK(hipMemcpy(T, hT.data(), tbytes, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    std::vector<Variant> vs = {
        {"m8_base", launch_m8_base, true}, {"m8_pp", launch_m8_pp, false}, {"m8_pp_pref", launch_m8_pp_pref, false},
        {"m8_pipe", launch_m8_pipe, false}, {"m8_pipe_asm", launch_m8_pipe_asm, false}, {"m8_pref_asm", launch_m8_pref_asm, false},
        {"m8_pipe_asm_nont", launch_m8_pipe_asm_nont, false},
        {"m8_pref_Uonly", launch_m8_pref_Uonly, false}, {"m8_pref_Gonly", launch_m8_pref_Gonly, false},
        {"m8_pref_noargmin", launch_m8_pref_noargmin, false},
        {"pf0_none", launch_pf<0>, false}, {"pf1_scalar", launch_pf<1>, false}, {"pf2_vtouch", launch_pf<2>, false},
    };
    std::vector<uint64_t> out((size_t)n), refh;
    const int grids[] = {2048, 1024, 1536, 4096};
    for (auto &v : vs) {
        for (int g : grids) {
            int blocks = g;
            if ((int64_t)blocks * 4 > n) blocks = (int)((n + 3) / 4);
            CK(hipMemcpy(recA, hrec.data(), 8 * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(recB, hrec.data(), 8 * n, hipMemcpyHostToDevice));
            // warm + correctness
            v.launch(blocks, s, U, T, recA, v.inplace ? recA : recB, n, j);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(out.data(), v.inplace ? recA : recB, 8 * n, hipMemcpyDeviceToHost));
            if (refh.empty()) refh = out;
            size_t bad = 0;
            for (int64_t q = 0; q < n; ++q) bad += (out[q] != refh[q]);
            CK(hipMemcpy(recA, hrec.data(), 8 * n, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) v.launch(blocks, s, U, T, recA, v.inplace ? recA : recB, n, j);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("%-20s grid=%5d  %8.1f us/launch  gather %6.2f TB/s  total(L1) %6.2f TB/s  mismatches=%zu\n", v.name, blocks, us,
                   n * 7.0 * 1024 / us * 1e-6, n * 8.0 * 1024 / us * 1e-6, bad);
        }
    }
    // m = 16 (block-row 3.75 MiB: the L2-pressure case)
    std::vector<Variant> v16 = {{"m16_pref_asm", launch_m16_pref_asm, false}, {"m16_pipe_asm", launch_m16_pipe_asm, false}};
    for (auto &v : v16)
        for (int g : {2048, 1024}) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) v.launch(g, s, U, T, recA, recB, n, j);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("%-20s grid=%5d  %8.1f us/launch  gather %6.2f TB/s\n", v.name, g, us, n * 15.0 * 1024 / us * 1e-6);
        }
    {
        struct SV { const char *name; void (*launch)(int, hipStream_t, const float *, const float *, const uint64_t *, uint64_t *, int64_t, int); int waves; };
        std::vector<SV> sv = {{"split8_lds16", launch_split8_lds16, 8}, {"split8_vgpr8", launch_split8_vgpr8, 8},
                              {"split16_lds16", launch_split16_lds16, 16}, {"split16_lds32", launch_split16_lds32, 16}, {"split4_lds16", launch_split4_lds16, 4}};
        for (auto &v : sv)
            for (int bpc : {1, 2, 4, 8}) {
                const int blocks = 256 * bpc;
                if (bpc * v.waves > 32) continue;
                v.launch(blocks, s, U, T, recA, recB, n, j);
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) v.launch(blocks, s, U, T, recA, recB, n, j);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps;
                printf("%-20s blocks/CU=%d  %8.1f us/launch  gather %6.2f TB/s  U-stream %5.2f TB/s\n", v.name, bpc, us, n * 7.0 * 1024 / us * 1e-6, n * 1024.0 / us * 1e-6);
            }
    }
    {
        struct CV { const char *name; void (*launch)(hipStream_t, const float *, const float *, const uint64_t *, uint64_t *, int64_t, int); };
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        CK(hipFuncSetAttribute((const void *)cu_split_kernel<8, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        std::vector<CV> cv = {{"cu_G_only(npf0)", launch_cusplit<0, false>}, {"cu_npf2", launch_cusplit<2, false>}, {"cu_npf4", launch_cusplit<4, false>},
                              {"cu_npf6", launch_cusplit<6, false>}, {"cu_full_noPF(npf0,U)", launch_cusplit<0, true>}, {"cu_npf4_consumeU", launch_cusplit<4, true>}};
        for (auto &v : cv) {
            v.launch(s, U, T, recA, recB, n, j);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) v.launch(s, U, T, recA, recB, n, j);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("%-24s %8.1f us/launch  gather %6.2f TB/s\n", v.name, us, n * 7.0 * 1024 / us * 1e-6);
        }
    }
    {
        // slice-major copy of U and the partial buffer
        float *Us; float2 *part;
        CK(hipMalloc(&Us, sizeof(float) * n * H));
        CK(hipMalloc(&part, sizeof(float2) * n * 16));
        std::vector<float> hUs((size_t)n * H);
        for (int64_t i = 0; i < n; ++i)
            for (int a = 0; a < H; ++a) hUs[((size_t)(a >> 4) * n + i) * 16 + (a & 15)] = hU[(size_t)i * H + a];
        CK(hipMemcpy(Us, hUs.data(), sizeof(float) * n * H, hipMemcpyHostToDevice));
        CK(hipMemcpy(recA, hrec.data(), 8 * n, hipMemcpyHostToDevice));
        const int ldsb = 7 * 256 * 64;
        CK(hipFuncSetAttribute((const void *)slice_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
        for (int nr : {16, 32}) {
            auto run = [&]() {
                hipLaunchKernelGGL((slice_kernel<8>), dim3(16 * nr), dim3(1024), ldsb, s, Us, T, recA, part, n, j, nr);
                hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, recA, recB, n, j);
            };
            run();
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(out.data(), recB, 8 * n, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int64_t q = 0; q < n; ++q) bad += (out[q] != refh[q]);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) run();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((slice_kernel<8>), dim3(16 * nr), dim3(1024), ldsb, s, Us, T, recA, part, n, j, nr);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("slice_lds ranges=%2d      %8.1f us/(slice+combine)   slice kernel alone %8.1f us   mismatches vs base=%zu\n", nr, us, ms * 1e3 / reps, bad);
        }
    }
    // debug: first mismatching records between base and pp
    {
        CK(hipMemcpy(recA, hrec.data(), 8 * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(recB, hrec.data(), 8 * n, hipMemcpyHostToDevice));
        launch_m8_pp(2048, s, U, T, recA, recB, n, j);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(out.data(), recB, 8 * n, hipMemcpyDeviceToHost));
        // CPU truth for the first 2000 vectors
        size_t base_bad = 0, pp_bad = 0;
        for (int64_t q = 0; q < 2000 && q < n; ++q) {
            float sv[H];
            for (int a = 0; a < H; ++a) sv[a] = hU[(size_t)q * H + a];
            for (int kk = 0; kk < 7; ++kk) {
                const int k = kk + (kk >= j ? 1 : 0);
                const unsigned code = (unsigned)(hrec[q] >> (8 * k)) & 0xffu;
                for (int a = 0; a < H; ++a) sv[a] = sv[a] + hT[((size_t)(k * H) + code) * H + a];
            }
            int best = 0;
            for (int a = 1; a < H; ++a) if (sv[a] < sv[best]) best = a;
            const uint64_t expect = (hrec[q] & ~(0xffull << (8 * j))) | ((uint64_t)best << (8 * j));
            base_bad += (refh[q] != expect);
            pp_bad += (out[q] != expect);
        }
        printf("CPU check (2000 vectors): base wrong %zu, pp wrong %zu\n", base_bad, pp_bad);
        int shown = 0;
        for (int64_t q = 0; q < n && shown < 6; ++q)
            if (out[q] != refh[q]) { printf("mismatch i=%lld in=%016llx base=%016llx pp=%016llx\n", (long long)q, (unsigned long long)hrec[q], (unsigned long long)refh[q], (unsigned long long)out[q]); ++shown; }
    }
    return 0;
}
