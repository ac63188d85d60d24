// lsq_icm_lane.hip -- LDS-walk ICM node update, ONE LANE PER VECTOR (schedule 3, m <= 8).
//
// STATUS: EXPERIMENT, OFF BY DEFAULT (option "lane").  Bit-exact (the gpu parity suite passes with it), but at
// 1024 threads per block (<= 128 VGPRs) hipcc spills ~180 B/lane to scratch and the kernel runs 2.5x SLOWER than
// icm_walk_kernel (409 vs 161 us per node update of 10^6 vectors, same box).  Kept as the starting point for a
// lower-pressure variant (8 candidates per lane); see DESIGN.md section 7.
//
// Same data-flow as icm_walk_kernel (lsq_icm.hip): a 1024-thread block walks the 16 slices of 16
// candidates for its own <= 4096 vectors, slice tables staged in LDS from the slice-major copy Ts,
// unaries streamed slice-major from HBM, exact skip of unchanged nodes through a compacted list.
// What changes is the lane mapping.  icm_walk_kernel gives a vector 4 lanes x 4 candidates; rocprofv3
// (SQ_INSTS_*, SQ_WAIT_*) showed that kernel dependency/issue-latency bound at 6.1 instructions per
// vector-slice with only 4 waves per SIMD.  Here a lane owns a whole vector for the slice:
//   * 16 accumulators per lane (4 x float4); a wave iteration covers 64 vectors, so every scalar,
//     address and control instruction is amortised over 4x more work (~3.8 instructions per
//     vector-slice);
//   * LDS: lane l reads the four 16-byte pieces of its table entry in the XOR-swizzled order
//     piece(r) = r ^ (l & 3).  Within a ds_read_b128 service group (16 lanes) the four lanes that read
//     the same piece index belong to different vectors with random codes -- exactly the conflict
//     statistics of the 4-lanes-per-vector layout (~2-way), not the 4..16-way of a naive layout;
//   * the same lane serves the same vector in every slice, so the running first-argmin (packed
//     orderable(value) << 32 | index, compared as one u64) is a private LDS word of that lane:
//     no atomics, no cross-lane reduction at all.
// Arithmetic is identical: s = U; s += T[k][b_k] for k ascending (plain f32 adds); lowest index of
// the minimum; NaN handling of the reference's strict-< scan (encode_icm.jl:76-119).
#include "lsq_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LANE_SL = 16;          // candidates per slice
constexpr int LANE_PP = 4096;        // vectors per block pass = 4 wave iterations of 64 x 16 waves

template <int M, int ABL>
__global__ __launch_bounds__(1024) void icm_lane_kernel(const float *__restrict__ Usj, const float *__restrict__ Tsj,
                                                        uint8_t *__restrict__ rec, unsigned short *__restrict__ valid,
                                                        int64_t n, int j, int per_pass, int use_skip,
                                                        unsigned long long *__restrict__ active_total) {
    static_assert(M >= 1 && M <= 8, "one 8-byte code record per vector");
    constexpr int SL = LANE_SL;
    constexpr int CS = 8;
    constexpr int NS = LSQ_H / SL;
    constexpr int CW = (M - 1 + 3) / 4;
    constexpr int RW = 2;
    constexpr int TAB = (M - 1) * LSQ_H * 4;            // f32x4 entries of one slice table
    constexpr int NST = (TAB + 1023) / 1024;
    extern __shared__ f32x4 lds_lane[];
    f32x4 *tab = lds_lane;
    unsigned long long *best64 = reinterpret_cast<unsigned long long *>(lds_lane + TAB);   // [LANE_PP] running first-argmin, touched only by the owning lane
    unsigned short *list = reinterpret_cast<unsigned short *>(best64 + LANE_PP);            // [LANE_PP] active local indices
    __shared__ int wave_tot[16];
    __shared__ int nact_s;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rot = lane & 3;

    uint32_t sel[CW > 0 ? CW : 1];
#pragma unroll
    for (int w = 0; w < CW; ++w) {
        uint32_t sv = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = 4 * w + t;
            const int k = kk + (kk >= j ? 1 : 0);
            sv |= (uint32_t)((kk < M - 1 ? k - 4 * w : 0) & 7) << (8 * t);
        }
        sel[w] = sv;
    }

    struct Item { f32x4 u[4]; uint32_t r[RW]; };

    const int64_t npass = (n + per_pass - 1) / per_pass;
    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int64_t lo = pass * per_pass;
        const int64_t hi = (lo + per_pass < n) ? lo + per_pass : n;
        const int cnt = (int)(hi - lo);

        // ---- compact list of the vectors whose node j must be recomputed (exact skip) ----
        {
            const int base = (int)threadIdx.x * 4;
            int f[4], c = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = base + e;
                f[e] = 0;
                if (idx < cnt) f[e] = (!use_skip) || !((valid[lo + idx] >> j) & 1);
                c += f[e];
            }
            int inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            if (lane == 63) wave_tot[wave] = inc;
            __syncthreads();
            int wbase = 0;
            for (int w2 = 0; w2 < wave; ++w2) wbase += wave_tot[w2];
            int pos = wbase + inc - c;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (f[e]) list[pos++] = (unsigned short)(base + e);
            if (threadIdx.x == 1023) nact_s = wbase + inc;
            __syncthreads();
        }
        const int nact = nact_s;
        if (nact == 0) continue;                               // block-uniform
        if (threadIdx.x == 0 && active_total) atomicAdd(active_total, (unsigned long long)nact);

        // iterations of this wave per slice: vectors wave*64 + t*1024 + lane, t < ipw <= LANE_IT
        const int ipw = (wave * 64 < nact) ? (nact - wave * 64 + 1023) / 1024 : 0;
        int ls = 0, lit = 0;                                   // (slice, iteration) of the next load to issue
        auto load_next = [&](Item &it) {                       // flat pipeline over (slice, iteration); clamped, never masked
            int ci = wave * 64 + lit * 1024 + lane;
            ci = ci < nact ? ci : nact - 1;
            const int lsc = ls < NS ? ls : NS - 1;
            const int64_t i = lo + list[ci];
            const f32x4 *up = reinterpret_cast<const f32x4 *>(Usj + ((int64_t)lsc * n + i) * SL);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ABL == 1) it.u[r] = (f32x4){(float)i, 1.f, 2.f, (float)r};
                else it.u[r] = __builtin_nontemporal_load(up + (r ^ rot));
            }
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rec + i * CS);
            it.r[0] = rp[0];
            it.r[1] = rp[1];
            if (++lit >= ipw) { lit = 0; ++ls; }
        };

        Item nextit;
        load_next(nextit);

        for (int slice = 0; slice < NS; ++slice) {
            {
                // stage this slice's table: the L2 loads are issued BEFORE the barrier (their latency overlaps the
                // wait for the slowest wave); the registers are live only across the barrier, when nothing else is
                f32x4 stg[NST > 0 ? NST : 1];
                const f32x4 *src = reinterpret_cast<const f32x4 *>(Tsj) + (int64_t)slice * TAB;
#pragma unroll
                for (int r = 0; r < NST; ++r) {
                    const int e = (int)threadIdx.x + r * 1024;
                    stg[r] = (e < TAB) ? src[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();                               // everyone is done with the previous slice table
#pragma unroll
                for (int r = 0; r < NST; ++r) {
                    const int e = (int)threadIdx.x + r * 1024;
                    if (e < TAB) tab[e] = stg[r];
                }
                __syncthreads();
            }
#pragma unroll 1
            for (int t = 0; t < ipw; ++t) {                    // wave-uniform trip count (<= LANE_IT)
                {
                    f32x4 acc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = nextit.u[r];
                    const uint32_t r0 = nextit.r[0], r1 = nextit.r[1];
                    load_next(nextit);                         // next (slice, iteration) item travels under this one's work
                    // conditioning codes, ascending k with j skipped, compacted by v_perm_b32
                    uint32_t cwv[2];
                    cwv[0] = __builtin_amdgcn_perm(r1, r0, sel[0]);
                    cwv[1] = (CW > 1) ? __builtin_amdgcn_perm(0u, r1, sel[CW > 1 ? 1 : 0]) : 0u;
                    // two table entries in flight (32 VGPRs): reads of kk+1 are issued before the adds of kk;
                    // the memory clobber keeps the compiler from hoisting all 28 reads at once (which spills at 128 VGPRs)
                    f32x4 ta[4], tb[4];
                    auto rd = [&](int kk, f32x4 (&dst)[4]) {
                        const uint32_t code = (cwv[kk >> 2] >> (8 * (kk & 3))) & 0xffu;
                        const f32x4 *ent = tab + (kk * LSQ_H + code) * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r] = ent[r ^ rot];
                    };
                    if (M > 1 && ABL != 2) {
                        rd(0, ta);
#pragma unroll
                        for (int kk = 0; kk < M - 1; ++kk) {
                            if (kk + 1 < M - 1) { if ((kk & 1) == 0) rd(kk + 1, tb); else rd(kk + 1, ta); }
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = acc[r] + (((kk & 1) == 0) ? ta[r] : tb[r]);    // ascending k, plain f32 adds
                            asm volatile("" ::: "memory");      // LDS reads of kk+2 may not be hoisted above this point
                        }
                    }
                    if (ABL == 2) acc[0].x += (float)(cwv[0] & 0xff);
                    // first-argmin over the 16 candidates: per accumulator (min, first index) -> packed key -> u64 min
                    unsigned long long key = ~0ull;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x4 a = acc[r];
                        const float lm = fminf(fminf(a.x, a.y), fminf(a.z, a.w));           // NaN-ignoring, like the strict-< scan
                        const uint32_t e = (a.x == lm) ? 0u : (a.y == lm) ? 1u : (a.z == lm) ? 2u : 3u;
                        const uint32_t idx = (uint32_t)(SL * slice) + 4u * (uint32_t)(r ^ rot) + e;
                        const uint32_t bits = __float_as_uint(lm + 0.0f);                  // -0 -> +0 (equal in the reference)
                        uint32_t ord = bits ^ ((uint32_t)((int32_t)bits >> 31) | 0x80000000u);  // monotone float -> uint
                        ord = (lm != lm) ? 0xffffffffu : ord;                              // all four NaN: never wins
                        const unsigned long long k64 = ((unsigned long long)ord << 32) | idx;
                        key = k64 < key ? k64 : key;
                    }
                    // s[0] is element x of the accumulator that holds piece 0, i.e. acc[rot]; NaN there: the scan keeps index 0
                    const float s0 = (rot == 0) ? acc[0].x : (rot == 1) ? acc[1].x : (rot == 2) ? acc[2].x : acc[3].x;
                    key = ((slice == 0) & (s0 != s0)) ? 0ull : key;
                    const int ci = wave * 64 + t * 1024 + lane;                            // this lane owns the vector in every slice
                    if (ci < nact) {
                        const unsigned long long old = (slice == 0) ? ~0ull : best64[ci];
                        if (key < old) best64[ci] = key;                                   // (value, index) order: lowest index keeps ties
                    }
                }
            }
        }
        // ---- write the new codes and validity masks of the recomputed vectors ----
        for (int t = 0; t < ipw; ++t) {
            const int ci = wave * 64 + t * 1024 + lane;
            if (ci < nact) {
                const int64_t i = lo + list[ci];
                const unsigned bi = (unsigned)(best64[ci] & 0xffffffffull);
                const uint8_t code = (uint8_t)(bi > 255 ? 0 : bi);
                const uint8_t old = rec[i * CS + j];
                rec[i * CS + j] = code;
                if (valid) valid[i] = (code != old) ? (unsigned short)(1u << j) : (unsigned short)(valid[i] | (1u << j));
            }
        }
        __syncthreads();                                       // list / tab are reused by the next pass
    }
}

template <int M, int ABL>
int launch_lane_t(hipStream_t s, const float *Usj, const float *Ts, uint8_t *rec, unsigned short *valid, int64_t n, int j,
                  int use_skip, unsigned long long *active_total) {
    constexpr int NS = LSQ_H / LANE_SL;
    constexpr int TAB = (M - 1) * LSQ_H * 4;
    constexpr int LDS_BYTES = TAB * 16 + LANE_PP * 8 + LANE_PP * 2;
    static_assert(LDS_BYTES + 256 <= 160 * 1024, "slice table + list must fit the 160 KiB LDS");
    static bool attr_set[64] = {false};
    int dev = 0;
    LSQ_HIP(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        LSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&icm_lane_kernel<M, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    const int64_t rounds = (n + 256 * (int64_t)LANE_PP - 1) / (256 * (int64_t)LANE_PP);
    int64_t per_pass = (n + 256 * rounds - 1) / (256 * rounds);
    if (per_pass > LANE_PP) per_pass = LANE_PP;
    if (per_pass < 1) per_pass = 1;
    const int64_t npass = (n + per_pass - 1) / per_pass;
    const unsigned grid = (unsigned)(npass < 256 ? npass : 256);
    const float *Tsj = Ts + (int64_t)j * NS * TAB * 4;
    hipLaunchKernelGGL((icm_lane_kernel<M, ABL>), dim3(grid), dim3(1024), LDS_BYTES, s, Usj, Tsj, rec, valid, n, j, (int)per_pass,
                       (use_skip && valid) ? 1 : 0, active_total);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

}  // namespace

int lsq_launch_icm_lane(hipStream_t s, const float *Usj, const float *Ts, uint8_t *rec, unsigned short *valid, int64_t n, int m, int j,
                        int use_skip, unsigned long long *active_total, int ablation) {
    if (n <= 0) return LSQ_OK;
    if (ablation == 1 && m == 8) return launch_lane_t<8, 1>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
    if (ablation == 2 && m == 8) return launch_lane_t<8, 2>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
    switch (m) {
        case 1: return launch_lane_t<1, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 2: return launch_lane_t<2, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 3: return launch_lane_t<3, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 4: return launch_lane_t<4, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 5: return launch_lane_t<5, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 6: return launch_lane_t<6, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 7: return launch_lane_t<7, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        case 8: return launch_lane_t<8, 0>(s, Usj, Ts, rec, valid, n, j, use_skip, active_total);
        default: lsq_set_error("icm_lane: m = %d unsupported (1..8)", m); return LSQ_EINVAL;
    }
}
