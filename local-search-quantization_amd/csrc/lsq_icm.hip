// lsq_icm.hip -- the ILS/ICM encoder kernels for gfx950 (wave64).
//
// Replaces (does NOT translate) the reference's CUDA kernels src/encodings/cuda/cudautils.cu:
//   condition_icm3 (:236-339)  -> icm_walk_kernel  (LDS-staged table slices, slice-major unary stream)
//   perturb        (:27-80)    -> perturb_kernel   (Philox counter RNG, no state buffer)
//   veccost2       (:145-183)  -> cost_kernel / cost2_kernel (fused with the accept rule)
//   setup_kernel / vec_add     -> gone (counter-based RNG; ||c||^2 is the GEMM epilogue)
// Semantics follow the reference CPU path (src/encodings/encode_icm.jl), restated in
// oracle/lsq_oracle.c: conditioning adds in ascending k (plain f32 adds), argmin = LOWEST index
// of the minimum (the reference CUDA tree reduction is not -- SURVEY 2b), accept iff strictly
// better.
//
// Data layout in HBM:
//   U   [m][256/SL][n][SL] f32  slice-major unary planes: one wave load = 1 KiB = 64/(SL/4) vectors x SL candidates
//   T   [m][m][256][256] f32; T[j][k][b][:] = the 1 KiB column added to node j when codebook k
//       holds code b (row-major; light blocks gather whole columns from L2)
//   Ts  [m][256/SL][m-1][256][SL] f32: the same columns regrouped per slice (one contiguous block per staged slice)
//   rec [n][cs] u8        code records, cs = 8 (m <= 8) or 16: one aligned 8/16-byte load
//   X   [n][d] f32,  K [m*256][d] f32  (the Julia buffers, read in place)
// (The three schedules of round 1 -- per-node L2 gathers, fused sweeps, slices + combine -- were removed in round 4; `git log` has them.)
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "lsq_wave.h"
#include <atomic>

#include "lsq_cost.h"

namespace {

// ---- LDS-walk kernel -------------------------------------------------------------------------------------------------------
// ONE block walks the 256/SL slices of a node for its own range of <= 4096 vectors, keeping the running (min value, index) of
// every vector in LDS as a packed 64-bit key.  Ts is the slice-major copy of the pair tables, Ts[j][slice][kk][b][SL]
// (kk = rank of k among k != j), so that staging one slice is one contiguous, fully coalesced copy of (m-1) x 256 x SL x 4 B.
// NT: threads per block (1024 or 512); DEPTH: U items in flight per wave (<= 8); ABL (tuning build only): timing-only ablations
// (1: no U stream, 2: no table adds, 3: no slice barriers, 4: U stream only).
template <int M, int SL, int ABL = 0, int DEPTH = 2, int NT = 1024>
__global__ __launch_bounds__(NT) void icm_walk_kernel(const float *__restrict__ U, const float *__restrict__ Ts, const float *__restrict__ T,
                                                        uint8_t *__restrict__ rec, unsigned short *__restrict__ valid,
                                                        int64_t n, const WalkNodes nodes, int per_pass, int use_skip, int direct_max,
                                                        unsigned long long *__restrict__ active_total,
                                                        const uint8_t *__restrict__ ref_rec, const unsigned short *__restrict__ ref_valid,
                                                        const int *__restrict__ idle_if_set) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int NS = LSQ_H / SL;
    if (idle_if_set && *idle_if_set) return;            // the filtered walk (icm_walkq_kernel) did this launch's work
    constexpr int LPV = SL / 4;
    constexpr int VPW = 64 / LPV;
    constexpr int CW = (M - 1 + 3) / 4;
    constexpr int RW = CS / 4;
    constexpr int TAB = (M - 1) * LSQ_H * LPV;          // f32x4 entries of one slice table
    constexpr int PP = LSQ_WALK_PP(M, SL);               // vectors per pass (LDS budget)
    extern __shared__ f32x4 lds_walk[];
    f32x4 *tab = lds_walk;
    // running first-argmin per (compact) vector: one packed 64-bit key = orderable(value) << 32 | candidate index,
    // minimised with ONE LDS atomic per lane -- the atomic does the cross-lane and the cross-slice reduction,
    // and the packed compare returns the LOWEST index among equal values (encode_icm.jl:105-119).
    unsigned long long *best64 = reinterpret_cast<unsigned long long *>(lds_walk + TAB);      // [PP]
    unsigned short *list = reinterpret_cast<unsigned short *>(best64 + PP);                    // [PP] active local indices
    __shared__ int wave_tot[16];
    __shared__ int nact_s;
    __shared__ unsigned stat_s[4 + LSQ_WALK_TRACE];      // per-block statistics, flushed once per launch (per-node device atomics stalled every node's first barrier)
    for (int e = threadIdx.x; e < 4 + LSQ_WALK_TRACE; e += NT) stat_s[e] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int v = lane / LPV, q = lane % LPV;
    constexpr int NW = NT / 64;                          // waves per block
    constexpr int EPT = 4096 / NT;                       // vectors per thread in the compaction prologue
    constexpr int step = NW * VPW;

    struct Item { f32x4 u; uint32_t r[RW]; };

    // Walks slices [s_begin, s_end) of node j for the `nact` vectors listed in list[] (indices relative to `lo`; dense = the list is
    // the identity) and leaves every vector's packed running minimum in best64[] (initialised by the caller).  Ends with a barrier.
    auto walk_slices = [&](const int j, const int64_t lo, const int nact, const bool dense, const int s_begin, const int s_end) {
        const float *__restrict__ Usj = U + (int64_t)j * n * LSQ_H;
        const float *__restrict__ Tsj = Ts + (int64_t)j * NS * TAB * 4;
        uint32_t sel[CW > 0 ? CW : 1];      // v_perm_b32 selectors: compact word w takes the conditioning codes k(4w..4w+3) (ascending k, j skipped)
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
        constexpr int NST = (TAB + NT - 1) / NT;               // float4 table entries staged per thread
        f32x4 nxt[NST > 0 ? NST : 1];
        auto prefetch_tab = [&](int sl) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Tsj) + (int64_t)sl * TAB;
#pragma unroll
            for (int r = 0; r < NST; ++r) {
                const int e = (int)threadIdx.x + r * NT;
                nxt[r] = (e < TAB) ? src[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        };
        prefetch_tab(s_begin);

        // The U stream is ONE flat software pipeline over (slice, iteration), DEPTH items in flight per
        // wave (3 at 1024 threads: the most that fits 128 VGPRs without spills, 1 % faster than 2; 4..8 at 512 threads: slower at m = 8): the loads of the first iterations of slice s+1 are in flight
        // while slice s finishes.  The two item buffers have STATIC roles (loop unrolled by two, the
        // roles swap when a slice has an odd iteration count) so no register copies are issued: the
        // kernel is instruction-issue bound (ablations in DESIGN.md), every slot counts.
        const int ipw = (wave * VPW < nact) ? (nact - wave * VPW + step - 1) / step : 0;   // iterations per slice, this wave
        int ls = s_begin, lit = 0;                             // (slice, iteration) of the next load to issue
        auto load_next = [&](Item &it) {
            // always in bounds (indices clamped): no exec-mask juggling; results of clamped lanes are discarded
            int ci = wave * VPW + lit * step + v;
            ci = ci < nact ? ci : nact - 1;
            const int lsc = ls < s_end ? ls : s_end - 1;
            // wave-uniform 64-bit bases (SGPRs) + 32-bit lane offsets: one VALU instruction per address instead of a 64-bit chain
            uint32_t li = (uint32_t)ci;                        // dense block (every vector active): the list is the identity,
            if (!dense) li = list[ci];                         // skip the LDS round trip in front of the load addresses
            const char *ub = reinterpret_cast<const char *>(Usj + ((int64_t)lsc * n + lo) * SL);
            const char *rb = reinterpret_cast<const char *>(rec + lo * CS);
            const uint32_t uo = li * (uint32_t)(SL * 4) + (uint32_t)q * 16u;
            if (ABL == 1) it.u = (f32x4){(float)li, 1.f, 2.f, 3.f};
            else it.u = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(ub + uo));      // streamed once per node update: non-temporal measured 3 % faster than a cached load
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rb + li * (uint32_t)CS);
#pragma unroll
            for (int w = 0; w < RW; ++w) it.r[w] = rp[w];
            if (++lit >= ipw) { lit = 0; ++ls; }
        };
        auto gather = [&](const Item &cur) -> f32x4 {
            f32x4 s = cur.u;
#pragma unroll
            for (int w = 0; w < CW; ++w) {
                const uint32_t hiw = (w + 1 < RW) ? cur.r[w + 1] : 0u;
                const uint32_t cw = __builtin_amdgcn_perm(hiw, cur.r[w], sel[w]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * w + t;
                    if (kk < M - 1) {
                        uint32_t code;
                        // low byte through an opaque v_and so that the address is and + lshl_add (the optimiser's own form,
                        // (cw << 6) & 0x3fc0 then + base, is one VALU instruction longer); the others are bfe/lshr + lshl_add
                        if (t == 0) asm("v_and_b32 %0, 0xff, %1" : "=v"(code) : "v"(cw));
                        else code = (cw >> (8 * t)) & 0xffu;
                        if (ABL != 2 && ABL != 4) s = s + tab[(kk * LSQ_H + code) * LPV + q];      // ascending k, plain f32 add
                        else s.x += (float)code;
                    }
                }
            }
            return s;
        };
        auto finish = [&](f32x4 s, int slice, int c0) {
            // first-argmin: in-lane over 4 candidates, then one packed LDS atomic min per lane
            float lm = fminf(fminf(s.x, s.y), fminf(s.z, s.w));
            uint32_t li = (s.z == lm) ? 2u : 3u;                               // three selects, no divergent control flow
            li = (s.y == lm) ? 1u : li;
            li = (s.x == lm) ? 0u : li;
            li += 4u * q + (uint32_t)(SL * slice);
            const uint32_t bits = __float_as_uint(lm + 0.0f);                  // -0 -> +0: they compare equal in the reference
            uint32_t ord = bits ^ ((uint32_t)((int32_t)bits >> 31) | 0x80000000u);   // monotone float -> uint
            // only the lane(s) holding the minimum of the vector's LPV lanes post it (DPP mins within the quad): 4x fewer LDS
            // atomics and no same-address serialisation (15 % of the LDS cycles); equal minima all post, the packed key orders them
            float vm = lm;
            if (LPV >= 2) vm = fminf(vm, dpp_self<DPP_XOR1, 0xf>(vm));
            if (LPV >= 4) vm = fminf(vm, dpp_self<DPP_XOR2, 0xf>(vm));
            bool post = (lm == vm);                                            // false for a NaN lane: NaN never wins ...
            if (__builtin_expect(__ballot((lm != lm) | (s.x != s.x)) != 0ull, 0)) {       // wave-uniform, rare: a NaN is present
                if ((slice == 0) & (q == 0) & (s.x != s.x)) { ord = 0u; li = 0u; post = true; }   // ... except s[0]: the strict-< scan keeps index 0
            }
            if ((c0 + v < nact) & post) atomicMin(&best64[c0 + v], ((unsigned long long)ord << 32) | li);
        };
        auto compute = [&](const Item &cur, int slice, int c0) { finish(gather(cur), slice, c0); };
        Item buf[DEPTH];
#pragma unroll
        for (int e = 0; e < DEPTH; ++e) load_next(buf[e]);
        int phase = 0;                                         // index of the buffer holding the next item to consume

        for (int slice = s_begin; slice < s_end; ++slice) {
            if (ABL < 3) __syncthreads();                      // everyone is done with the previous slice table
#pragma unroll
            for (int r = 0; r < NST; ++r) {                    // commit the table prefetched one slice ago
                const int e = (int)threadIdx.x + r * NT;
                if (e < TAB && ABL != 4) tab[e] = nxt[r];
            }
            if (ABL < 3) __syncthreads();
            if (slice + 1 < s_end && ABL != 4) prefetch_tab(slice + 1);       // next slice's table travels L2 -> VGPRs under this slice's work
            int c0 = wave * VPW, t = 0;
            auto run = [&](auto P_) {                          // P = buffer consumed first; all buffer indices are compile-time
                constexpr int P = decltype(P_)::value;
                for (; t + DEPTH <= ipw; t += DEPTH) {
#pragma unroll
                    for (int e = 0; e < DEPTH; ++e) {
                        compute(buf[(P + e) % DEPTH], slice, c0); load_next(buf[(P + e) % DEPTH]); c0 += step;
                    }
                }
#pragma unroll
                for (int e = 0; e < DEPTH - 1; ++e)
                    if (t < ipw) {
                        compute(buf[(P + e) % DEPTH], slice, c0); load_next(buf[(P + e) % DEPTH]); c0 += step;
                        ++t; phase = (P + e + 1) % DEPTH;
                    }
            };
            bool ran = false;                                  // run() changes `phase`: exactly one instantiation per slice
            auto try_phase = [&](auto P_) {
                if constexpr (decltype(P_)::value < DEPTH) {
                    if (!ran && phase == decltype(P_)::value) { run(P_); ran = true; }
                }
            };
            try_phase(std::integral_constant<int, 0>{}); try_phase(std::integral_constant<int, 1>{});
            try_phase(std::integral_constant<int, 2>{}); try_phase(std::integral_constant<int, 3>{});
            try_phase(std::integral_constant<int, 4>{}); try_phase(std::integral_constant<int, 5>{});
            try_phase(std::integral_constant<int, 6>{}); try_phase(std::integral_constant<int, 7>{});
        }
        __syncthreads();
    };

    const int64_t npass = (n + per_pass - 1) / per_pass;
    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int64_t lo = pass * per_pass;
        const int64_t hi = (lo + per_pass < n) ? lo + per_pass : n;
        const int cnt = (int)(hi - lo);

        // A block owns its vector range for the whole launch, so the launch may carry a SEQUENCE of node updates
        // (a full ILS iteration: icmiter sweeps x m nodes, encode_icm.jl:72-76): node update t+1 of a vector only
        // depends on node update t of the same vector, which this block wrote itself (ordered by the barriers).
        for (int nu = 0; nu < nodes.count; ++nu) {
        const int j = nodes.j[nu];
        const float *__restrict__ Usj = U + (int64_t)j * n * LSQ_H;
        // ---- compact list of the vectors whose node j must be recomputed (exact skip: a node whose
        // conditioning codes did not change since it was last minimised keeps the same argmin)
        {
            const int base = (int)threadIdx.x * EPT;
            int f[EPT], c = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
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
            for (int e = 0; e < EPT; ++e)
                if (f[e]) list[pos++] = (unsigned short)(base + e);
            if (threadIdx.x == NT - 1) nact_s = wbase + inc;
            __syncthreads();
        }
        const int nact = nact_s;
        if (nact == 0) { __syncthreads(); continue; }          // block-uniform (the barrier protects nact_s / wave_tot reuse)
        if (threadIdx.x == 0) {                            // [0] node updates recomputed, [1] staged / [2] light block-node-updates ([3]: filtered walk)
            stat_s[0] += (unsigned)nact;
            stat_s[nact <= direct_max ? 2 : 1] += 1u;
            stat_s[4 + ((nodes.pos0 + nu) & (LSQ_WALK_TRACE - 1))] += (unsigned)nact;
        }
        if (nact <= direct_max) {
            // LIGHT block (few active vectors: small n, or a late sweep): staging the whole (m-1) x 256 KiB table through
            // LDS would cost more than the vectors need.  One wave per vector instead, the (m-1) 1 KiB table columns
            // gathered straight from L2 (row-major T) -- the icm_node_kernel arithmetic on the slice-major U layout.
            const float *__restrict__ Tj = T + (int64_t)j * M * LSQ_H * LSQ_H;
            constexpr int LB = LSQ_LIGHT_LB(M);
            for (int r0 = wave; r0 < nact; r0 += NW * LB) {
                int64_t vi[LB];
                bool on[LB];
#pragma unroll
                for (int e = 0; e < LB; ++e) {
                    const int r = r0 + e * NW;
                    on[e] = r < nact;
                    vi[e] = lo + __builtin_amdgcn_readfirstlane((int)list[on[e] ? r : r0]);
                }
                light_update<M, CS, LB>(rec, valid, ref_rec, ref_valid, Usj, Tj, n, SL, j, vi, on, lane);
            }
            __syncthreads();
            continue;
        }
        for (int ci = threadIdx.x; ci < nact; ci += NT) best64[ci] = ~0ull;      // ordered before the first atomics by the slice-0 barriers

        walk_slices(j, lo, nact, nact == cnt, 0, NS);
        {   // all of a thread's loads before its first store: one global round trip for its PP / NT vectors
            constexpr int EPD = (PP + NT - 1) / NT;
            int64_t vi[EPD];
            uint32_t vcode[EPD];
            bool von[EPD];
#pragma unroll
            for (int e = 0; e < EPD; ++e) {
                const int ci = (int)threadIdx.x + e * NT;
                von[e] = ci < nact;
                vi[e] = lo + (von[e] ? list[ci] : 0);
                vcode[e] = von[e] ? (uint32_t)(best64[ci] & 0xffffffffull) : 0u;
            }
            apply_node_results<CS, EPD>(rec, valid, vi, vcode, von, j, ref_rec, ref_valid);
        }
        __syncthreads();
        }   // node updates
    }
    __syncthreads();
    if (active_total)
        for (int e = threadIdx.x; e < 4 + LSQ_WALK_TRACE; e += NT)
            if (stat_s[e]) atomicAdd(active_total + e, (unsigned long long)stat_s[e]);
}

// Ts[j][slice][kk][b][SL] <- T[j][k(kk)][b][slice*SL ..]   (one thread per float4)
template <int SL>
__global__ __launch_bounds__(256) void tables_to_slices_kernel(const float *__restrict__ T, float *__restrict__ Ts, int m) {
    constexpr int NS = LSQ_H / SL, LPV = SL / 4;
    const int64_t total = (int64_t)m * NS * (m - 1) * LSQ_H * LPV;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int qq = (int)(e % LPV);
    int64_t r = e / LPV;
    const int b = (int)(r % LSQ_H); r /= LSQ_H;
    const int kk = (int)(r % (m - 1)); r /= (m - 1);
    const int slice = (int)(r % NS);
    const int j = (int)(r / NS);
    const int k = kk + (kk >= j ? 1 : 0);
    reinterpret_cast<f32x4 *>(Ts)[e] =
        *reinterpret_cast<const f32x4 *>(T + (((int64_t)j * m + k) * LSQ_H + b) * LSQ_H + slice * SL + qq * 4);
}

// ---- small chunks: a WAVE owns its vectors through every node update of the launch ------------------------------------------------
// When a chunk is so small that every block of the walk kernel would be "light" anyway (at most `light` vectors per block), the block
// structure only costs: per node update a compaction (validity words from memory), barriers, and the bookkeeping round trips.  Here a wave keeps
// the records and validity words of its LSQ_LIGHT_LB vectors in scalar registers for the whole launch; a node update is the light routine's
// arithmetic (unary row from the slice-major planes, (m-1) table rows from L2, plain f32 adds in ascending k, first argmin) and nothing else;
// records and validity words are written once, at the end.  Same codes, same memoisation rules (skip, fall-back) as icm_walk_kernel.
template <int M>
__global__ __launch_bounds__(256) void icm_wave_kernel(const float *__restrict__ U, const float *__restrict__ T, uint8_t *__restrict__ rec,
                                                       unsigned short *__restrict__ valid, int64_t n, const WalkNodes nodes, int SL, int use_skip,
                                                       unsigned long long *__restrict__ active_total, const uint8_t *__restrict__ ref_rec,
                                                       const unsigned short *__restrict__ ref_valid, const int *__restrict__ idle_if_set) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int LB = LSQ_LIGHT_LB(M);
    if (idle_if_set && *idle_if_set) return;
    __shared__ unsigned stat_s[4 + LSQ_WALK_TRACE];
    for (int e = threadIdx.x; e < 4 + LSQ_WALK_TRACE; e += 256) stat_s[e] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t i0 = wv * LB;
    const int LPV = SL / 4;
    const bool have_ref = ref_rec && ref_valid;
    CodeRec cr[LB], rr[LB];
    uint32_t vm[LB], rv[LB];
    bool on[LB];
#pragma unroll
    for (int e = 0; e < LB; ++e) {
        on[e] = i0 + e < n;
        cr[e].lo = cr[e].hi = rr[e].lo = rr[e].hi = 0ull;
        vm[e] = rv[e] = 0u;
        if (on[e]) {
            cr[e] = load_rec<CS>(rec, i0 + e);
            if (valid) vm[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)valid[i0 + e]);      // maintained whenever the pointer is given (lsq_internal.h), consulted only with use_skip
            if (have_ref) {
                rr[e] = load_rec<CS>(ref_rec, i0 + e);
                rv[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref_valid[i0 + e]);
            }
        }
    }
    unsigned total = 0, wave_nodes = 0;
    for (int nu = 0; nu < nodes.count; ++nu) {
        const int j = nodes.j[nu];
        bool need[LB];
        unsigned cnt = 0;
#pragma unroll
        for (int e = 0; e < LB; ++e) {
            need[e] = on[e] && (!use_skip || !((vm[e] >> j) & 1u));
            cnt += need[e] ? 1u : 0u;
        }
        if (cnt == 0) continue;
        const float *__restrict__ Usj = U + (int64_t)j * n * LSQ_H;
        const float *__restrict__ Tj = T + (int64_t)j * M * LSQ_H * LSQ_H;
        f32x4 s[LB], c[LB][M > 1 ? M - 1 : 1];
#pragma unroll
        for (int e = 0; e < LB; ++e) {
            s[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (need[e]) s[e] = (reinterpret_cast<const f32x4 *>(Usj + ((int64_t)(lane / LPV) * n + (i0 + e)) * SL))[lane % LPV];      // small chunks: the planes live in L2 / the Infinity Cache across sweeps
        }
#pragma unroll
        for (int e = 0; e < LB; ++e)
            if (need[e]) {
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) {
                    const int k = kk + (kk >= j ? 1 : 0);
                    c[e][kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * LSQ_H) + cr[e].get(k)) * LSQ_H)[lane];
                }
            }
#pragma unroll
        for (int e = 0; e < LB; ++e)
            if (need[e]) {
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) s[e] = s[e] + c[e][kk];      // ascending k, plain f32 adds
                const uint32_t code = (uint32_t)wave_first_argmin(s[e], lane);
                if (code != cr[e].get(j)) { cr[e].set(j, code); vm[e] = 1u << j; }      // a changed code invalidates every other node
                else vm[e] |= 1u << j;                                                  // an unchanged one confirms node j
                if (have_ref && cr[e].lo == rr[e].lo && (CS == 8 || cr[e].hi == rr[e].hi)) vm[e] |= rv[e];      // known_valid()
            }
        total += cnt;
        ++wave_nodes;
        if (lane == 0) atomicAdd(&stat_s[4 + ((nodes.pos0 + nu) & (LSQ_WALK_TRACE - 1))], cnt);
    }
#pragma unroll
    for (int e = 0; e < LB; ++e)
        if (on[e] && lane == 0) {
            *reinterpret_cast<uint64_t *>(rec + (i0 + e) * CS) = cr[e].lo;
            if (CS == 16) *reinterpret_cast<uint64_t *>(rec + (i0 + e) * CS + 8) = cr[e].hi;
            if (valid) valid[i0 + e] = (unsigned short)vm[e];
        }
    if (lane == 0) { atomicAdd(&stat_s[0], total); atomicAdd(&stat_s[2], wave_nodes); }      // [2]: light (wave, node) updates
    __syncthreads();
    if (active_total)
        for (int e = threadIdx.x; e < 4 + LSQ_WALK_TRACE; e += 256)
            if (stat_s[e]) atomicAdd(active_total + e, (unsigned long long)stat_s[e]);
}

template <int CS>
__global__ __launch_bounds__(256) void perturb_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t n,
                                                      int m, int npert, uint64_t seed, uint32_t it, uint64_t goff,
                                                      const unsigned short *__restrict__ vsrc, unsigned short *__restrict__ vdst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t w[2];
    const uint64_t *p = reinterpret_cast<const uint64_t *>(src + i * CS);
    w[0] = p[0];
    w[1] = (CS == 16) ? p[1] : 0ull;
    const bool changed = perturb_record(w, m, npert, seed, it, goff + (uint64_t)i);
    uint64_t *q = reinterpret_cast<uint64_t *>(dst + i * CS);
    q[0] = w[0];
    if (CS == 16) q[1] = w[1];
    // a changed code invalidates every node: the others' conditioning changed, and it is itself no argmin
    if (vdst) vdst[i] = changed ? (unsigned short)0 : vsrc[i];
}

// ---- cost (+ accept) ----------------------------------------------------------------------------
// utils.jl:225-254 per vector, reduction order = oracle cost_one(); mode 1 applies
// encode_icm.jl:178-186 (keep the new codes iff strictly better) and counts ==/< .
template <int M>
__global__ __launch_bounds__(256) void cost_kernel(const float *__restrict__ X, const float *__restrict__ K,
                                                   const uint8_t *rec, uint8_t *cur, float *__restrict__ prev,
                                                   unsigned long long *__restrict__ counters, int64_t n, int d, int mode,
                                                   const unsigned short *vnew, unsigned short *vcur, const lsq_perturb_next pn) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int RW = CS / 4;
    constexpr int NV = 2;                     // vectors in flight per wave (memory-level parallelism)
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    unsigned n_eq = 0, n_lt = 0;
    // A wave takes 64 consecutive vectors, lane l looks at vector base + l: in accept mode a vector whose candidate record
    // equals its current record has, bit for bit, the cost it already has (same codes, same arithmetic) -- it is counted as
    // "equal" (encode_icm_cuda.jl:199-204) without touching X or the codebooks.  ~60 % of the vectors from the second ILS
    // iteration on.  (A NaN cost is never "equal" in the reference's comparison, so those are evaluated.)
    for (int64_t base = w * 64; base < n; base += nwaves * 64) {
        const int64_t il = base + lane;
        const bool live = il < n;
        const int64_t ic = live ? il : n - 1;
        uint32_t rn[RW], cw[RW];
        bool same = (mode == 1);
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            rn[q] = reinterpret_cast<const uint32_t *>(rec + ic * CS)[q];
            cw[q] = (mode == 1) ? reinterpret_cast<const uint32_t *>(cur + ic * CS)[q] : rn[q];
            same = same && (rn[q] == cw[q]);
        }
        const float pl = (mode == 1) ? prev[ic] : 0.0f;
        const bool skip = live && same && (pl == pl);
        n_eq += (unsigned)__popcll(__ballot(skip)) * (lane == 0 ? 1u : 0u);
        unsigned short vfin = (live && vcur) ? vcur[il] : (unsigned short)0;       // the vector's validity word after this kernel (for the fused perturbation)
        const unsigned short vn = (live && vcur && mode == 1) ? vnew[il] : (unsigned short)0;
        if (same && live && vcur) { vfin = (unsigned short)(vfin | vn); vcur[il] = vfin; }      // same tuple: what the sweeps learnt about it is kept
        uint64_t accepted = 0;                                                       // bit l: the candidate of vector base + l replaced the current record
        uint64_t todo = __ballot(live && !skip);
        while (todo) {
            CodeRec cr[NV];
            int64_t ii[NV];
            float pcv[NV], part[NV];
            bool have[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {        // every independent load is issued up front
                have[v] = todo != 0;
                const int src = have[v] ? __builtin_ctzll(todo) : 0;
                if (have[v]) todo &= todo - 1;
                ii[v] = base + src;
                cr[v].lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rn[0], src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rn[1], src) << 32);
                cr[v].hi = 0;
                if (RW == 4) cr[v].hi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rn[RW - 2], src) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rn[RW - 1], src) << 32);
                pcv[v] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl), src));
                part[v] = 0.0f;
            }
            for (int t0 = 0; t0 < d; t0 += 64) {
                const int t = t0 + lane;
                const bool valid = t < d;
                const int tt = valid ? t : 0;
                float xv[NV], kv[NV][M];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    xv[v] = X[ii[v] * (int64_t)d + tt];
#pragma unroll
                    for (int k = 0; k < M; ++k) kv[v][k] = K[((int64_t)(k * LSQ_H) + cr[v].get(k)) * d + tt];
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float cb = 0.0f;
#pragma unroll
                    for (int k = 0; k < M; ++k) cb = cb + kv[v][k];          // k ascending from 0 (utils.jl:238-244)
                    const float r = cb - xv[v];
                    const float sq = r * r;                                   // never fused (-ffp-contract=off)
                    part[v] = part[v] + (valid ? sq : 0.0f);                  // lane partial: t = lane + 64 q, q ascending
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float cost = wave_sum_tree(part[v]);
                if (!have[v]) continue;
                if (mode == 0) {
                    if (lane == 0) prev[ii[v]] = cost;
                } else {
                    const float pc = pcv[v];
                    if (lane == 0) n_eq += (cost == pc);
                    if (cost < pc) {                                          // strict improvement only (encode_icm.jl:183-186)
                        accepted |= 1ull << (int)(ii[v] - base);
                        if (lane == 0) {
                            ++n_lt;
                            prev[ii[v]] = cost;
                            uint64_t *q = reinterpret_cast<uint64_t *>(cur + ii[v] * CS);
                            q[0] = cr[v].lo;
                            if (CS == 16) q[1] = cr[v].hi;
                            if (vcur) vcur[ii[v]] = vnew[ii[v]];
                        }
                    }
                }
            }
        }
        if (pn.on && live) {
            const bool acc = (accepted >> lane) & 1ull;
            uint32_t fin[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) fin[q] = acc ? rn[q] : cw[q];
            perturb_next_store<CS>(pn, il, fin, acc ? vn : vfin);
        }
    }
    if (mode == 1 && lane == 0 && (n_eq | n_lt)) {
        if (n_eq) atomicAdd(&counters[0], (unsigned long long)n_eq);
        if (n_lt) atomicAdd(&counters[1], (unsigned long long)n_lt);
    }
}

// Even d: half a wave per vector, 8-byte loads.  Lane l' (0..31) of a half owns dimensions
// t = 128c + {2l', 2l'+1, 2l'+64, 2l'+65}: the two residues x = 2l', 2l'+1 (mod 64) of the canonical 64 strided
// partial sums, each accumulated in ascending t, so the reduction order is exactly oracle cost_one()'s:
// level 1 in-lane (p[2l'] + p[2l'+1]), levels 2..32 across the 32 lanes of the half (DPP), result in the
// half's last lane.  Half the load instructions of cost_kernel and twice the bytes per instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int M>
__global__ __launch_bounds__(256) void cost2_kernel(const float *__restrict__ X, const float *__restrict__ K,
                                                    const uint8_t *rec, uint8_t *cur, float *__restrict__ prev,
                                                    unsigned long long *__restrict__ counters, int64_t n, int d, int mode,
                                                    const unsigned short *vnew, unsigned short *vcur, const lsq_perturb_next pn) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int RW = CS / 4;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, lp = lane & 31;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    unsigned n_eq = 0, n_lt = 0;
    // 64 consecutive vectors per wave batch; vectors whose candidate record equals the current one keep their cost (see
    // cost_kernel) and are skipped; the others are taken two at a time, one per half-wave.
    for (int64_t base = w * 64; base < n; base += nwaves * 64) {
        const int64_t il = base + lane;
        const bool livel = il < n;
        const int64_t ic = livel ? il : n - 1;
        uint32_t rn[RW], cw[RW];
        bool same = (mode == 1);
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            rn[q] = reinterpret_cast<const uint32_t *>(rec + ic * CS)[q];
            cw[q] = (mode == 1) ? reinterpret_cast<const uint32_t *>(cur + ic * CS)[q] : rn[q];
            same = same && (rn[q] == cw[q]);
        }
        const float pl = (mode == 1) ? prev[ic] : 0.0f;
        const bool skip = livel && same && (pl == pl);
        const unsigned nskip = (unsigned)__popcll(__ballot(skip));              // all lanes vote, lane 0 keeps the wave's counters
        if (lane == 0) n_eq += nskip;
        unsigned short vfin = (livel && vcur) ? vcur[il] : (unsigned short)0;      // the vector's validity word after this kernel (for the fused perturbation)
        const unsigned short vn = (livel && vcur && mode == 1) ? vnew[il] : (unsigned short)0;
        if (same && livel && vcur) { vfin = (unsigned short)(vfin | vn); vcur[il] = vfin; }     // same tuple: what the sweeps learnt about it is kept
        uint64_t accepted = 0;                                                      // bit l: the candidate of vector base + l replaced the current record
        uint64_t todo = __ballot(livel && !skip);
        while (todo) {
            const int sa = __builtin_ctzll(todo);
            todo &= todo - 1;
            const bool haveb = todo != 0;
            const int sb = haveb ? __builtin_ctzll(todo) : sa;                // odd count: the second half repeats the first, unwritten
            if (haveb) todo &= todo - 1;
            const bool live = half ? haveb : true;
            const int64_t i = base + (half ? sb : sa);
            uint32_t r[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) {
                const uint32_t ra_ = (uint32_t)__builtin_amdgcn_readlane((int)rn[q], sa), rb_ = (uint32_t)__builtin_amdgcn_readlane((int)rn[q], sb);
                r[q] = half ? rb_ : ra_;
            }
            const float pca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl), sa));
            const float pcb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl), sb));
            const float pc = half ? pcb : pca;
            const float *x = X + i * (int64_t)d;
            const float *kb[M];
#pragma unroll
            for (int k = 0; k < M; ++k) kb[k] = K + ((int64_t)(k * LSQ_H) + ((r[k >> 2] >> (8 * (k & 3))) & 0xffu)) * d;
            float p0 = 0.0f, p1 = 0.0f;
            for (int c0 = 0; c0 < d; c0 += 128) {
                const int ta = c0 + 2 * lp, tb = ta + 64;
                const bool va = ta < d, vb = tb < d;
                const int ua = va ? ta : 0, ub = vb ? tb : 0;
                const f32x2 xa = *reinterpret_cast<const f32x2 *>(x + ua), xb = *reinterpret_cast<const f32x2 *>(x + ub);
                f32x2 ka[M], kbv[M];
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    ka[k] = *reinterpret_cast<const f32x2 *>(kb[k] + ua);
                    kbv[k] = *reinterpret_cast<const f32x2 *>(kb[k] + ub);
                }
                f32x2 ca = (f32x2){0.f, 0.f}, cb = (f32x2){0.f, 0.f};
#pragma unroll
                for (int k = 0; k < M; ++k) { ca = ca + ka[k]; cb = cb + kbv[k]; }      // k ascending from 0 (utils.jl:238-244)
                const f32x2 ra = ca - xa, rb = cb - xb;
                const f32x2 sa2 = ra * ra, sb2 = rb * rb;                               // never fused (-ffp-contract=off)
                p0 = p0 + (va ? sa2.x : 0.0f);                                          // residue 2l':   t ascending
                p1 = p1 + (va ? sa2.y : 0.0f);                                          // residue 2l'+1
                p0 = p0 + (vb ? sb2.x : 0.0f);
                p1 = p1 + (vb ? sb2.y : 0.0f);
            }
            float v = p0 + p1;                                                          // tree level 1
            v = v + dpp_self<DPP_XOR1, 0xf>(v);                                         // levels 2, 4, 8, 16: within the row
            v = v + dpp_self<DPP_XOR2, 0xf>(v);
            v = v + dpp_self<DPP_HALF_MIRROR, 0xf>(v);
            v = v + dpp_self<DPP_MIRROR, 0xf>(v);
            v = v + dpp_zero<DPP_BCAST15, 0xa>(v);                                      // level 32: rows 1 and 3 add rows 0 and 2
            const float costA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
            const float costB = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
            const float cost = half ? costB : costA;
            if (mode == 0) {
                if (live && lp == 0) prev[i] = cost;
            } else {
                const bool eq = live && (cost == pc), lt = live && (cost < pc);          // strict improvement only (encode_icm.jl:183-186)
                const uint64_t bl = __ballot(lt && lp == 0);                              // bit 0: first half's vector (sa), bit 32: second half's (sb)
                const unsigned ne = (unsigned)__popcll(__ballot(eq && lp == 0)), nl = (unsigned)__popcll(bl);
                accepted |= ((bl & 1ull) ? 1ull << sa : 0ull) | (((bl >> 32) & 1ull) ? 1ull << sb : 0ull);
                if (lane == 0) { n_eq += ne; n_lt += nl; }
                if (lt && lp == 0) {
                    prev[i] = cost;
                    uint32_t *qd = reinterpret_cast<uint32_t *>(cur + i * CS);
#pragma unroll
                    for (int q = 0; q < RW; ++q) qd[q] = r[q];
                    if (vcur) vcur[i] = vnew[i];
                }
            }
        }
        if (pn.on && livel) {
            const bool acc = (accepted >> lane) & 1ull;
            uint32_t fin[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) fin[q] = acc ? rn[q] : cw[q];
            perturb_next_store<CS>(pn, il, fin, acc ? vn : vfin);
        }
    }
    if (mode == 1 && lane == 0 && (n_eq | n_lt)) {
        if (n_eq) atomicAdd(&counters[0], (unsigned long long)n_eq);
        if (n_lt) atomicAdd(&counters[1], (unsigned long long)n_lt);
    }
}

// NQ = 1 (d <= 64): the compiler's own register budget; NQ = 2: four waves per SIMD (128 VGPRs) so that the 18 loads of a round really are in flight together
template <int M, int MODE, int HASV, int NQ>
__global__ __launch_bounds__(256) void cost4_kernel(const float *__restrict__ X, const float *__restrict__ K, const uint8_t *rec, uint8_t *cur,
                                                    float *__restrict__ prev, unsigned long long *__restrict__ counters, int64_t n, int d,
                                                    const unsigned short *vnew, unsigned short *vcur, const lsq_perturb_next pn) {
    cost4_body<M, MODE, HASV, NQ>(X, K, rec, cur, prev, counters, 0, n, __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))),
                                  (int64_t)gridDim.x * 4, d, vnew, vcur, pn);
}
template <int M, int MODE, int HASV, int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void cost4w_kernel(const float *__restrict__ X, const float *__restrict__ K,
                                                    const uint8_t *rec, uint8_t *cur, float *__restrict__ prev,
                                                    unsigned long long *__restrict__ counters, int64_t n, int d,
                                                    const unsigned short *vnew, unsigned short *vcur, const lsq_perturb_next pn) {
    cost4_body<M, MODE, HASV, NQ>(X, K, rec, cur, prev, counters, 0, n, __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))),
                                  (int64_t)gridDim.x * 4, d, vnew, vcur, pn);
}

__global__ __launch_bounds__(256) void sum_f64_kernel(const float *__restrict__ v, int64_t n, double *__restrict__ sum) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += (double)v[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---- layout conversion ---------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void codes_expand_kernel(const uint8_t *__restrict__ tight, int64_t n, int m, uint8_t *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t w[2] = {0ull, 0ull};
    for (int j = 0; j < m; ++j) w[j >> 3] |= (uint64_t)tight[i * m + j] << (8 * (j & 7));
    uint64_t *q = reinterpret_cast<uint64_t *>(rec + i * CS);
    q[0] = w[0];
    if (CS == 16) q[1] = w[1];
}
template <int CS>
__global__ __launch_bounds__(256) void codes_compact_kernel(const uint8_t *__restrict__ rec, int64_t n, int m, uint8_t *__restrict__ tight) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < m; ++j) tight[i * m + j] = rec[i * CS + j];
}
template <int CS>
__global__ __launch_bounds__(256) void codes_from_i16_kernel(const int16_t *__restrict__ B, int64_t n, int m, int h,
                                                             uint8_t *__restrict__ rec, int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t w[2] = {0ull, 0ull};
    for (int j = 0; j < m; ++j) {
        const int v = (int)B[i * m + j];
        if (v < 1 || v > h) { *bad = 1; continue; }
        w[j >> 3] |= (uint64_t)(v - 1) << (8 * (j & 7));
    }
    uint64_t *q = reinterpret_cast<uint64_t *>(rec + i * CS);
    q[0] = w[0];
    if (CS == 16) q[1] = w[1];
}
template <int CS>
__global__ __launch_bounds__(256) void codes_to_i16_kernel(const uint8_t *__restrict__ rec, int64_t n, int m, int16_t *__restrict__ B) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < m; ++j) B[i * m + j] = (int16_t)(rec[i * CS + j] + 1);
}

// ---- synthetic generators (benchmark harness; mirrored by the oracle) -----------------------------
__global__ __launch_bounds__(256) void synth_data_u8_kernel(uint64_t seed, uint64_t goff, int64_t n, int d, float *__restrict__ X) {
    // one thread per 4 consecutive t (one Philox block)
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d4 = (d + 3) >> 2;
    if (q >= n * d4) return;
    const int64_t i = q / d4;
    const int t0 = (int)(q % d4) * 4;
    const lsq_u32x4 r = lsq_rng_block(seed, goff + (uint64_t)i, (uint32_t)(t0 >> 10), LSQ_DOM_DATA, (uint32_t)((t0 & 1023) >> 2));
    for (int e = 0; e < 4 && t0 + e < d; ++e) X[i * (int64_t)d + t0 + e] = (float)(r.v[e] >> 24);
}
__global__ __launch_bounds__(256) void randinit_kernel(uint64_t seed, uint64_t goff, int64_t n, int m, int h, uint8_t *__restrict__ tight) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < m; ++j)
        tight[i * m + j] = (uint8_t)lsq_mulhi32(lsq_rng_word(seed, goff + (uint64_t)i, 0, LSQ_DOM_INIT, (uint32_t)j), (uint32_t)h);
}
__global__ __launch_bounds__(256) void synth_codebooks_kernel(uint64_t seed, int m, int h, int d, float *__restrict__ K) {
    // codeword (j,a) = (1/m) * synthetic data vector number pick(j,a) of the stream (seed ^ 0x5bd1e995)
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)m * h * d) return;
    const int64_t row = q / d;
    const int t = (int)(q % d);
    const uint64_t pick = lsq_rng_word(seed, (uint64_t)row, 0, LSQ_DOM_CODEBOOK, 0);
    const uint32_t w = lsq_rng_word(seed ^ 0x5bd1e995ull, pick, (uint32_t)(t >> 10), LSQ_DOM_DATA, (uint32_t)(t & 1023));
    K[q] = (float)(w >> 24) / (float)m;
}

inline unsigned wave_grid(int64_t n) {      // persistent grid: 4 waves per block, <= 8 blocks per CU on 256 CUs
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}
inline unsigned thread_grid(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace


#define LSQ_DISPATCH_M(m, EXPR)                                                              \
    switch (m) {                                                                             \
        case 1: { constexpr int M_ = 1; EXPR; } break;   case 2: { constexpr int M_ = 2; EXPR; } break;   \
        case 3: { constexpr int M_ = 3; EXPR; } break;   case 4: { constexpr int M_ = 4; EXPR; } break;   \
        case 5: { constexpr int M_ = 5; EXPR; } break;   case 6: { constexpr int M_ = 6; EXPR; } break;   \
        case 7: { constexpr int M_ = 7; EXPR; } break;   case 8: { constexpr int M_ = 8; EXPR; } break;   \
        case 9: { constexpr int M_ = 9; EXPR; } break;   case 10: { constexpr int M_ = 10; EXPR; } break; \
        case 11: { constexpr int M_ = 11; EXPR; } break; case 12: { constexpr int M_ = 12; EXPR; } break; \
        case 13: { constexpr int M_ = 13; EXPR; } break; case 14: { constexpr int M_ = 14; EXPR; } break; \
        case 15: { constexpr int M_ = 15; EXPR; } break; case 16: { constexpr int M_ = 16; EXPR; } break; \
        default: lsq_set_error("m = %d out of range 1..16", m); return LSQ_EINVAL;          \
    }

// geometry of a walk launch over n vectors: vectors per pass (<= the LDS budget PP), number of passes (= segments), grid
void lsq_walk_geometry(int64_t n, int m, int *per_pass, int *npass, int *pp_cap) {
    const int PP = lsq_walk_pp(m, lsq_walk_slice_width(m));
    const int64_t rounds = (n + 256 * (int64_t)PP - 1) / (256 * (int64_t)PP);          // passes per CU
    int64_t per = rounds > 0 ? (n + 256 * rounds - 1) / (256 * rounds) : 1;
    if (per > PP) per = PP;
    if (per < 1) per = 1;
    *per_pass = (int)per;
    *npass = (int)((n + per - 1) / per);
    if (pp_cap) *pp_cap = PP;
}

template <int M, int SL, int ABL = 0, int DEPTH = 2, int NT = 1024>
static int launch_walk_t(hipStream_t s, const float *U, const float *Ts, const float *T, uint8_t *rec, unsigned short *valid, int64_t n,
                         const WalkNodes &nodes, int use_skip, unsigned long long *active_total, int light,
                         const uint8_t *ref_rec, const unsigned short *ref_valid, const int *idle_if_set) {
    constexpr int TAB = (M - 1) * LSQ_H * (SL / 4);
    constexpr int PP = LSQ_WALK_PP(M, SL);
    constexpr int LDS_BYTES = TAB * 16 + PP * 8 + PP * 2;                // slice table + packed running best + active list
    static_assert(LDS_BYTES + 256 <= 160 * 1024, "slice table + running best must fit the 160 KiB LDS");
    int per_pass = 1, npass = 1;
    lsq_walk_geometry(n, M, &per_pass, &npass, nullptr);
    // blocks with at most this many active vectors gather from L2 instead of staging (option "light"; thresholds 96..1024 measured)
    const int direct_max = light >= 0 ? light : LSQ_KNOB("LSQ_WALK_DIRECT", 256);
    const int skip = (use_skip && valid) ? 1 : 0;
    static LdsOptIn optin;
    LSQ_TRY(optin_lds(optin, &icm_walk_kernel<M, SL, ABL, DEPTH, NT>, LDS_BYTES));
    const unsigned grid = (unsigned)(npass < 256 ? npass : 256);
    hipLaunchKernelGGL((icm_walk_kernel<M, SL, ABL, DEPTH, NT>), dim3(grid), dim3(NT), LDS_BYTES, s, U, Ts, T, rec, valid, n, nodes, per_pass,
                       skip, (T && ABL == 0) ? direct_max : 0, active_total, skip ? ref_rec : nullptr, skip ? ref_valid : nullptr, idle_if_set);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_walk_slice_width(int m) {
    return (m <= 8 && LSQ_KNOB("LSQ_WALK_SL", 16) != 8) ? 16 : 8;
}

// `order[nnodes]`: the node updates to run back to back inside the launch (1 = one node; icmiter*m = a whole ILS iteration)
int lsq_launch_icm_walk(hipStream_t s, const float *U, const float *Ts, const float *T, uint8_t *rec, unsigned short *valid, int64_t n, int m,
                        const int32_t *order, int nnodes, int pos0, int use_skip, unsigned long long *active_total, int ablation, int light,
                        const uint8_t *ref_rec, const unsigned short *ref_valid, const int *idle_if_set) {
    if (n <= 0 || nnodes <= 0) return LSQ_OK;
    if (m < 1 || m > LSQ_MAX_M) { lsq_set_error("m = %d out of range 1..16", m); return LSQ_EINVAL; }
    for (int done = 0; done < nnodes; done += LSQ_WALK_MAX_NODES) {
        WalkNodes nodes;
        nodes.count = (nnodes - done < LSQ_WALK_MAX_NODES) ? nnodes - done : LSQ_WALK_MAX_NODES;
        nodes.pos0 = pos0 + done;
        for (int t = 0; t < nodes.count; ++t) {
            const int j = order[done + t];
            if (j < 0 || j >= m) { lsq_set_error("node %d out of range 0..%d", j, m - 1); return LSQ_EINVAL; }
            nodes.j[t] = (uint8_t)j;
        }
        // m >= 14: up to 15 table reads in flight + 8 staged table registers per thread do not fit 128 VGPRs (measured at
        // m = 16: 67..100 spilled registers, 1.5..2.5x slower) -> 512-thread blocks (256 VGPRs per wave), more U items in
        // flight instead.  m = 9..13 fit (<= 4 spills) and are 3-5 % faster with 1024 threads (measured for every m).
#define LSQ_WALK_ARGS s, U, Ts, T, rec, valid, n, nodes, use_skip, active_total, light, ref_rec, ref_valid, idle_if_set
#define LSQ_WALK_CASE_MID(MM) case MM: LSQ_TRY((launch_walk_t<MM, 8, 0, 2>(LSQ_WALK_ARGS))); break;
#define LSQ_WALK_CASE_BIG(MM) case MM: LSQ_TRY((launch_walk_t<MM, 8, 0, 4, 512>(LSQ_WALK_ARGS))); break;
#define LSQ_WALK_CASE(MM, SLL) case MM: LSQ_TRY((launch_walk_t<MM, SLL, 0, 3>(LSQ_WALK_ARGS))); break;
#ifdef LSQ_TUNING      // timing-only variants and alternative shapes: profiling library only (results of the ablations are garbage)
        bool handled = true;
        if (m <= 8 && lsq_walk_slice_width(m) == 8) {
            switch (m) {
                LSQ_WALK_CASE(1, 8) LSQ_WALK_CASE(2, 8) LSQ_WALK_CASE(3, 8) LSQ_WALK_CASE(4, 8)
                LSQ_WALK_CASE(5, 8) LSQ_WALK_CASE(6, 8) LSQ_WALK_CASE(7, 8) LSQ_WALK_CASE(8, 8)
            }
        } else if (m == 8 && ablation == 1) { LSQ_TRY((launch_walk_t<8, 16, 1>(LSQ_WALK_ARGS)));
        } else if (m == 8 && ablation == 2) { LSQ_TRY((launch_walk_t<8, 16, 2>(LSQ_WALK_ARGS)));
        } else if (m == 8 && ablation == 3) { LSQ_TRY((launch_walk_t<8, 16, 3>(LSQ_WALK_ARGS)));
        } else if (m == 8 && ablation == 4) { LSQ_TRY((launch_walk_t<8, 16, 4>(LSQ_WALK_ARGS)));
        } else if (m >= 14 && LSQ_KNOB("LSQ_WALK_BIG_NT", 512) == 1024) {
            switch (m) { case 14: LSQ_TRY((launch_walk_t<14, 8>(LSQ_WALK_ARGS))); break; case 15: LSQ_TRY((launch_walk_t<15, 8>(LSQ_WALK_ARGS))); break;
                         case 16: LSQ_TRY((launch_walk_t<16, 8>(LSQ_WALK_ARGS))); break; }
        } else handled = false;
        if (handled) continue;
#else
        (void)ablation;
#endif
        switch (m) {
            LSQ_WALK_CASE(1, 16) LSQ_WALK_CASE(2, 16) LSQ_WALK_CASE(3, 16) LSQ_WALK_CASE(4, 16)
            LSQ_WALK_CASE(5, 16) LSQ_WALK_CASE(6, 16) LSQ_WALK_CASE(7, 16) LSQ_WALK_CASE(8, 16)
            LSQ_WALK_CASE_MID(9) LSQ_WALK_CASE_MID(10) LSQ_WALK_CASE_MID(11) LSQ_WALK_CASE_MID(12)
            LSQ_WALK_CASE_MID(13) LSQ_WALK_CASE_BIG(14) LSQ_WALK_CASE_BIG(15) LSQ_WALK_CASE_BIG(16)
        }
#undef LSQ_WALK_ARGS
#undef LSQ_WALK_CASE
#undef LSQ_WALK_CASE_BIG
#undef LSQ_WALK_CASE_MID
    }
    return LSQ_OK;
}

int lsq_launch_tables_to_slices(hipStream_t s, const float *T, float *Ts, int m, int sl) {
    if (m < 2) return LSQ_OK;
    const int64_t total = (int64_t)m * (LSQ_H / sl) * (m - 1) * LSQ_H * (sl / 4);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (sl == 16) hipLaunchKernelGGL(tables_to_slices_kernel<16>, dim3(grid), dim3(256), 0, s, T, Ts, m);
    else hipLaunchKernelGGL(tables_to_slices_kernel<8>, dim3(grid), dim3(256), 0, s, T, Ts, m);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

#define LSQ_CS_LAUNCH(m, KERNEL, GRID, ...)                                                       \
    do {                                                                                           \
        if (lsq_code_stride(m) == 8) hipLaunchKernelGGL(KERNEL<8>, dim3(GRID), dim3(256), 0, s, __VA_ARGS__);  \
        else hipLaunchKernelGGL(KERNEL<16>, dim3(GRID), dim3(256), 0, s, __VA_ARGS__);             \
        LSQ_HIP(hipGetLastError());                                                                \
    } while (0)

// small chunks (at most `light` vectors per block of the walk kernel): every node update of the sequence in launches of <= 64, a wave per
// LSQ_LIGHT_LB vectors
int lsq_launch_icm_wave(hipStream_t s, const float *U, const float *T, uint8_t *rec, unsigned short *valid, int64_t n, int m, const int32_t *order,
                        int nnodes, int pos0, int use_skip, unsigned long long *active_total, const uint8_t *ref_rec, const unsigned short *ref_valid,
                        const int *idle_if_set) {
    if (n <= 0 || nnodes <= 0) return LSQ_OK;
    if (m < 1 || m > LSQ_MAX_M) { lsq_set_error("m = %d out of range 1..16", m); return LSQ_EINVAL; }
    const int skip = (use_skip && valid) ? 1 : 0;
    for (int done = 0; done < nnodes; done += LSQ_WALK_MAX_NODES) {
        WalkNodes nodes;
        nodes.count = (nnodes - done < LSQ_WALK_MAX_NODES) ? nnodes - done : LSQ_WALK_MAX_NODES;
        nodes.pos0 = pos0 + done;
        for (int t = 0; t < nodes.count; ++t) {
            const int j = order[done + t];
            if (j < 0 || j >= m) { lsq_set_error("node %d out of range 0..%d", j, m - 1); return LSQ_EINVAL; }
            nodes.j[t] = (uint8_t)j;
        }
        const int lb = LSQ_LIGHT_LB(m);
        const unsigned grid = (unsigned)((n + 4 * lb - 1) / (4 * lb));
        LSQ_DISPATCH_M(m, hipLaunchKernelGGL(icm_wave_kernel<M_>, dim3(grid), dim3(256), 0, s, U, T, rec, valid, n, nodes, lsq_walk_slice_width(m), skip,
                                             active_total, skip ? ref_rec : nullptr, skip ? ref_valid : nullptr, idle_if_set));
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_perturb(hipStream_t s, const uint8_t *src, uint8_t *dst, int64_t n, int m, int npert, uint64_t seed,
                       uint32_t it, uint64_t global_offset, const unsigned short *vsrc, unsigned short *vdst) {
    if (n <= 0) return LSQ_OK;
    LSQ_CS_LAUNCH(m, perturb_kernel, thread_grid(n), src, dst, n, m, npert, seed, it, global_offset, vsrc, vdst);
    return LSQ_OK;
}

const void *lsq_probe_kernel_icm() { return reinterpret_cast<const void *>(&tables_to_slices_kernel<16>); }

int lsq_launch_cost(hipStream_t s, const float *X, const float *K, const uint8_t *rec, uint8_t *cur, float *prev,
                    unsigned long long *counters, int64_t n, int d, int m, int mode, const unsigned short *vnew, unsigned short *vcur,
                    const lsq_perturb_next *next) {
    if (n <= 0) return LSQ_OK;
    lsq_perturb_next pn = {};
    if (next) pn = *next;
    pn.abl = LSQ_KNOB("LSQ_COST_ABL", 0);
    const int use_v2 = LSQ_KNOB("LSQ_COST_V2", 1), use_v4 = LSQ_KNOB("LSQ_COST_V4", 1);
    // a quarter wave per vector with 16-byte loads (any d that is a multiple of 4); else half a wave per vector with 8-byte loads (measured 13 %
    // faster than the scalar kernel at d = 128, 7 % slower at d = 960); else one wave per vector, 4-byte loads
    if (use_v4 && d % 4 == 0 && ((uintptr_t)X | (uintptr_t)K) % 16 == 0) {
        // persistent grid: as many 256-thread blocks as are resident at once (a second, partial generation of blocks would idle the CUs it does not reach)
        const int64_t want = (n + 255) / 256;
#define LSQ_COST4(KERN_, MODE_, HASV_, NQ_)                                                                                                        \
        LSQ_DISPATCH_M(m, {                                                                                                                 \
            static std::atomic<int> per_cu_known{0};      /* (lsq_multi_* runs one host thread per device through here) */                  \
            int per_cu = per_cu_known.load(std::memory_order_relaxed);                                                                      \
            if (per_cu == 0) {                                                                                                              \
                int nb = 0;                                                                                                                 \
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERN_<M_, MODE_, HASV_, NQ_>, 256, 0) != hipSuccess || nb < 1) nb = 4;   \
                per_cu = nb > 8 ? 8 : nb;                                                                                                   \
                per_cu_known.store(per_cu, std::memory_order_relaxed);                                                                      \
            }                                                                                                                               \
            const int64_t cap = 256 * (int64_t)per_cu;                                                                                      \
            hipLaunchKernelGGL((KERN_<M_, MODE_, HASV_, NQ_>), dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, s, X, K, rec, cur, prev, counters, n, d, vnew, vcur, pn); \
        })
        if (mode == 1 && (!vnew || !vcur)) { lsq_set_error("lsq_launch_cost: accept mode needs both validity arrays"); return LSQ_EINVAL; }
        if ((int64_t)m * LSQ_H * d >= (1ll << 31)) { lsq_set_error("lsq_launch_cost: codebook matrix too large"); return LSQ_EINVAL; }
        if (d > 64 && m <= 8) {             // two 64-float steps of d in flight (18 loads per lane); above m = 8 the codeword rows of ONE step already fill the registers
            if (mode == 1) { LSQ_COST4(cost4w_kernel, 1, 1, 2); } else if (vcur) { LSQ_COST4(cost4w_kernel, 0, 1, 2); } else { LSQ_COST4(cost4w_kernel, 0, 0, 2); }
        } else {
            if (mode == 1) { LSQ_COST4(cost4_kernel, 1, 1, 1); } else if (vcur) { LSQ_COST4(cost4_kernel, 0, 1, 1); } else { LSQ_COST4(cost4_kernel, 0, 0, 1); }
        }
#undef LSQ_COST4
    } else if (use_v2 && d % 2 == 0 && d <= 256 && ((uintptr_t)X | (uintptr_t)K) % 8 == 0) {
        LSQ_DISPATCH_M(m, hipLaunchKernelGGL(cost2_kernel<M_>, dim3(wave_grid((n + 1) / 2)), dim3(256), 0, s, X, K, rec, cur, prev, counters, n, d, mode, vnew, vcur, pn));
    } else {
        LSQ_DISPATCH_M(m, hipLaunchKernelGGL(cost_kernel<M_>, dim3(wave_grid((n + 1) / 2)), dim3(256), 0, s, X, K, rec, cur, prev, counters, n, d, mode, vnew, vcur, pn));
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_sum_f64(hipStream_t s, const float *v, int64_t n, double *sum) {
    if (n <= 0) return LSQ_OK;
    unsigned g = thread_grid(n);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(sum_f64_kernel, dim3(g), dim3(256), 0, s, v, n, sum);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_launch_codes_expand(hipStream_t s, const uint8_t *tight, int64_t n, int m, uint8_t *rec) {
    if (n <= 0) return LSQ_OK;
    LSQ_CS_LAUNCH(m, codes_expand_kernel, thread_grid(n), tight, n, m, rec);
    return LSQ_OK;
}
int lsq_launch_codes_compact(hipStream_t s, const uint8_t *rec, int64_t n, int m, uint8_t *tight) {
    if (n <= 0) return LSQ_OK;
    LSQ_CS_LAUNCH(m, codes_compact_kernel, thread_grid(n), rec, n, m, tight);
    return LSQ_OK;
}
int lsq_launch_codes_from_i16(hipStream_t s, const int16_t *B, int64_t n, int m, int h, uint8_t *rec, int *bad_flag) {
    if (n <= 0) return LSQ_OK;
    LSQ_CS_LAUNCH(m, codes_from_i16_kernel, thread_grid(n), B, n, m, h, rec, bad_flag);
    return LSQ_OK;
}
int lsq_launch_codes_to_i16(hipStream_t s, const uint8_t *rec, int64_t n, int m, int16_t *B) {
    if (n <= 0) return LSQ_OK;
    LSQ_CS_LAUNCH(m, codes_to_i16_kernel, thread_grid(n), rec, n, m, B);
    return LSQ_OK;
}

int lsq_launch_synth_data_u8(hipStream_t s, uint64_t seed, uint64_t global_offset, int64_t n, int d, float *X) {
    if (n <= 0) return LSQ_OK;
    const int64_t items = n * ((d + 3) >> 2);
    hipLaunchKernelGGL(synth_data_u8_kernel, dim3(thread_grid(items)), dim3(256), 0, s, seed, global_offset, n, d, X);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
int lsq_launch_randinit(hipStream_t s, uint64_t seed, uint64_t global_offset, int64_t n, int m, int h, uint8_t *tight) {
    if (n <= 0) return LSQ_OK;
    hipLaunchKernelGGL(randinit_kernel, dim3(thread_grid(n)), dim3(256), 0, s, seed, global_offset, n, m, h, tight);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
int lsq_launch_synth_codebooks(hipStream_t s, uint64_t seed, int m, int h, int d, float *K) {
    hipLaunchKernelGGL(synth_codebooks_kernel, dim3(thread_grid((int64_t)m * h * d)), dim3(256), 0, s, seed, m, h, d, K);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
