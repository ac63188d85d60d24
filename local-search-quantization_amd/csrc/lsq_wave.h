// lsq_wave.h -- wave64 cross-lane helpers and the pieces shared by the ICM walk kernels (gfx950 only; not a public header).
#pragma once

#include <mutex>

#include "lsq_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- wave64 cross-lane helpers (DPP; no LDS traffic) -------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_self(float v) {          // disabled lanes keep their own value
    const int iv = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_zero(float v) {          // disabled lanes receive +0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143 };

// minimum over the 64 lanes (NaN-ignoring, like the reference's strict-< scan), wave-uniform result
__device__ inline float wave_min(float v) {
    v = fminf(v, dpp_self<DPP_XOR1, 0xf>(v));
    v = fminf(v, dpp_self<DPP_XOR2, 0xf>(v));
    v = fminf(v, dpp_self<DPP_HALF_MIRROR, 0xf>(v));
    v = fminf(v, dpp_self<DPP_MIRROR, 0xf>(v));
    v = fminf(v, dpp_self<DPP_BCAST15, 0xa>(v));
    v = fminf(v, dpp_self<DPP_BCAST31, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum over the 64 lanes as a balanced pairwise tree, adjacent pairs first -- the exact order
// of oracle cost_one() ([build-defined 2]); f32 add is commutative so the mirrored DPP sources
// give the same bits.  Result valid in lane 63, returned wave-uniform.
__device__ inline float wave_sum_tree(float v) {
    v = v + dpp_self<DPP_XOR1, 0xf>(v);
    v = v + dpp_self<DPP_XOR2, 0xf>(v);
    v = v + dpp_self<DPP_HALF_MIRROR, 0xf>(v);
    v = v + dpp_self<DPP_MIRROR, 0xf>(v);
    v = v + dpp_zero<DPP_BCAST15, 0xa>(v);
    v = v + dpp_zero<DPP_BCAST31, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ inline uint64_t readfirstlane64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

struct CodeRec {           // wave-uniform code record (lives in SGPRs)
    uint64_t lo, hi;
    __device__ inline uint32_t get(int k) const { return (uint32_t)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))) & 0xffu); }
    __device__ inline void set(int k, uint32_t v) {
        if (k < 8) lo = (lo & ~(0xffull << (8 * k))) | ((uint64_t)v << (8 * k));
        else       hi = (hi & ~(0xffull << (8 * (k - 8)))) | ((uint64_t)v << (8 * (k - 8)));
    }
};

template <int CS>
__device__ inline CodeRec load_rec(const uint8_t *rec, int64_t i) {
    CodeRec r;
    const uint64_t *p = reinterpret_cast<const uint64_t *>(rec + i * CS);
    r.lo = readfirstlane64(p[0]);
    r.hi = (CS == 16) ? readfirstlane64(p[1]) : 0ull;
    return r;
}

// lowest index of the minimum of the wave's 256 conditioned values (encode_icm.jl:105-119)
__device__ inline int wave_first_argmin(f32x4 s, int lane) {
    const float lm = fminf(fminf(s.x, s.y), fminf(s.z, s.w));
    const float wm = wave_min(lm);
    const int inl = (s.x == wm) ? 0 : (s.y == wm) ? 1 : (s.z == wm) ? 2 : 3;
    const uint64_t mask = __ballot(lm == wm);
    int best = 0;
    if (mask != 0) {
        const int L = __builtin_ctzll(mask);
        best = 4 * L + __builtin_amdgcn_readlane(inl, L);
    }
    // strict '<' scan semantics: if s[0] is NaN nothing ever replaces it
    const float s0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s.x)));
    if (s0 != s0) best = 0;
    (void)lane;
    return best;
}

// vectors per block pass of the LDS-walk kernel: table + 10 B per vector must fit the 160 KiB LDS
constexpr int lsq_walk_pp(int M, int SL) {
    const int avail = 160 * 1024 - 256 - (M - 1) * LSQ_H * (SL / 4) * 16;      // bytes left beside the slice table
    const int pp = avail / 10 / 64 * 64;
    return pp > 4096 ? 4096 : pp;                                             // 4096 up to m = 14 (SL = 8), 4032 at m = 16
}
#define LSQ_WALK_PP(M, SL) lsq_walk_pp(M, SL)

// Code j of vector i changes: the whole record leaves as aligned words (a wave's stores cover whole lines: no byte-masked partial writes);
// callers skip the store when the code is unchanged.
template <int CS>
__device__ inline void store_code(uint8_t *__restrict__ rec, int64_t i, int j, uint8_t code, const uint32_t (&rw)[CS / 4]) {
    constexpr int RW = CS / 4;
    uint32_t out[RW];
#pragma unroll
    for (int w = 0; w < RW; ++w) out[w] = (w == (j >> 2)) ? ((rw[w] & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)code << (8 * (j & 3)))) : rw[w];
    if (RW == 2) *reinterpret_cast<uint64_t *>(rec + i * CS) = (uint64_t)out[0] | ((uint64_t)out[1] << 32);
    else {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4_t *>(rec + i * CS) = (u32x4_t){out[0], out[RW > 1 ? 1 : 0], out[RW > 2 ? 2 : 0], out[RW > 3 ? 3 : 0]};
    }
}

// Validity is a property of the code tuple alone ("code j is the first argmin of node j given the other codes").  When the
// candidate tuple, after node j took `code`, equals the vector's CURRENT tuple (the state the ILS iteration started from, whose
// validity bits were established earlier and are read-only during the sweeps), everything known about that tuple holds for
// the candidate too: its bits are OR-ed in.  A vector that has fallen back to a known fixed point stops being recomputed at
// once instead of being re-verified for another sweep.  Exact: only true statements about the same tuple are imported.
template <int RW>
__device__ inline unsigned short known_valid(const uint32_t (&rw)[RW], int j, uint8_t code, const uint8_t *ref, const unsigned short *refv) {
    if (!ref || !refv) return 0;
    bool same = true;
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        uint32_t mine = rw[w];
        if (w == (j >> 2)) mine = (mine & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)code << (8 * (j & 3)));
        same = same && (mine == reinterpret_cast<const uint32_t *>(ref)[w]);
    }
    return same ? *refv : (unsigned short)0;
}

// Result of a node update for vector i: code j <- the index part of the packed minimum key; validity bookkeeping (exact skip):
// a changed code invalidates every other node, an unchanged one confirms node j; what is known about the vector's current
// state is imported when the candidate tuple equals it (known_valid).
template <int CS>
__device__ inline void apply_node_result(uint8_t *__restrict__ rec, unsigned short *__restrict__ valid, int64_t i, int j, unsigned long long key,
                                         const uint8_t *__restrict__ ref_rec, const unsigned short *__restrict__ ref_valid) {
    constexpr int RW = CS / 4;
    const unsigned bi = (unsigned)(key & 0xffffffffull);
    const uint8_t code = (uint8_t)(bi > 255 ? 0 : bi);
    uint32_t rw[RW];                                     // the record before the update (one aligned load instead of a byte load)
#pragma unroll
    for (int w2 = 0; w2 < RW; ++w2) rw[w2] = reinterpret_cast<const uint32_t *>(rec + i * CS)[w2];
    const uint8_t old = (uint8_t)(rw[j >> 2] >> (8 * (j & 3)));
    if (code != old) store_code<CS>(rec, i, j, code, rw);
    if (valid) {
        unsigned short vm = (code != old) ? (unsigned short)(1u << j) : (unsigned short)(valid[i] | (1u << j));
        vm = (unsigned short)(vm | known_valid<RW>(rw, j, code, ref_rec ? ref_rec + i * CS : nullptr, ref_valid ? ref_valid + i : nullptr));
        valid[i] = vm;
    }
}

// The same bookkeeping for N vectors of one thread with ALL loads issued before the first store (one global round trip instead of N).
template <int CS, int N>
__device__ inline void apply_node_results(uint8_t *__restrict__ rec, unsigned short *__restrict__ valid, const int64_t (&idx)[N], const uint32_t (&code)[N],
                                          const bool (&on)[N], int j, const uint8_t *__restrict__ ref_rec, const unsigned short *__restrict__ ref_valid) {
    constexpr int RW = CS / 4;
    uint32_t rw[N][RW], rr[N][RW];
    unsigned short vo[N], rv[N];
    const bool have_ref = ref_rec && ref_valid;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        vo[e] = 0; rv[e] = 0;
#pragma unroll
        for (int w2 = 0; w2 < RW; ++w2) { rw[e][w2] = 0; rr[e][w2] = 0; }
        if (on[e]) {
#pragma unroll
            for (int w2 = 0; w2 < RW; ++w2) rw[e][w2] = reinterpret_cast<const uint32_t *>(rec + idx[e] * CS)[w2];
            if (valid) vo[e] = valid[idx[e]];
            if (have_ref) {
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) rr[e][w2] = reinterpret_cast<const uint32_t *>(ref_rec + idx[e] * CS)[w2];
                rv[e] = ref_valid[idx[e]];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        if (!on[e]) continue;
        const uint8_t c8 = (uint8_t)(code[e] > 255u ? 0u : code[e]);
        const uint8_t old = (uint8_t)(rw[e][j >> 2] >> (8 * (j & 3)));
        if (c8 != old) store_code<CS>(rec, idx[e], j, c8, rw[e]);
        if (valid) {
            unsigned short vm = (c8 != old) ? (unsigned short)(1u << j) : (unsigned short)(vo[e] | (1u << j));
            if (have_ref) {
                bool same = true;
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) {
                    uint32_t mine = rw[e][w2];
                    if (w2 == (j >> 2)) mine = (mine & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)c8 << (8 * (j & 3)));
                    same = same && (mine == rr[e][w2]);
                }
                if (same) vm = (unsigned short)(vm | rv[e]);
            }
            valid[idx[e]] = vm;
        }
    }
}

// LIGHT routine (blocks with few active vectors, and the filtered walk's last resort): one wave per vector, everything in f32 --
// the unary row from the slice-major planes (slices of SL floats), the (m-1) 1 KiB table rows gathered from L2 (row-major T), plain
// f32 adds in ascending k, first argmin, validity bookkeeping.  LB vectors of a wave are in flight together and every load of
// the bookkeeping (record, validity, reference record) is issued with the first batch: a vector costs one dependent round trip
// (record -> table rows) instead of three, shared between LB vectors.  `on[e]` is wave-uniform.
template <int M, int CS, int LB>
__device__ inline void light_update(uint8_t *__restrict__ rec, unsigned short *__restrict__ valid, const uint8_t *__restrict__ ref_rec,
                                    const unsigned short *__restrict__ ref_valid, const float *__restrict__ Usj, const float *__restrict__ Tj,
                                    int64_t n, int SL, int j, const int64_t (&vi)[LB], const bool (&on)[LB], int lane,
                                    unsigned short *vmir = nullptr, int64_t vmir_lo = 0) {      // vmir: the block's LDS mirror of valid[vmir_lo ..) (filtered walk), kept in step
    constexpr int RW = CS / 4;
    const int LPV = SL / 4;
    const bool have_ref = ref_rec && ref_valid;
    CodeRec cr[LB], rr[LB];
    uint32_t vo[LB], rv[LB];
    f32x4 s[LB];
#pragma unroll
    for (int e = 0; e < LB; ++e) {
        cr[e].lo = cr[e].hi = rr[e].lo = rr[e].hi = 0ull;
        vo[e] = rv[e] = 0u;
        s[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (on[e]) {
            cr[e] = load_rec<CS>(rec, vi[e]);
            s[e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Usj + ((int64_t)(lane / LPV) * n + vi[e]) * SL) + (lane % LPV));
            if (valid) vo[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)valid[vi[e]]);
            if (have_ref) {
                rr[e] = load_rec<CS>(ref_rec, vi[e]);
                rv[e] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref_valid[vi[e]]);
            }
        }
    }
    f32x4 c[LB][M > 1 ? M - 1 : 1];
#pragma unroll
    for (int e = 0; e < LB; ++e)
        if (on[e]) {
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) {
                const int k = kk + (kk >= j ? 1 : 0);
                c[e][kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * LSQ_H) + cr[e].get(k)) * LSQ_H)[lane];
            }
        }
#pragma unroll
    for (int e = 0; e < LB; ++e)
        if (on[e]) {
#pragma unroll
            for (int kk = 0; kk < M - 1; ++kk) s[e] = s[e] + c[e][kk];      // ascending k, plain f32 adds
            const uint8_t code = (uint8_t)wave_first_argmin(s[e], lane);
            if (lane == 0) {
                if (code != cr[e].get(j)) {
                    CodeRec mine = cr[e];
                    mine.set(j, code);
                    *reinterpret_cast<uint64_t *>(rec + vi[e] * CS) = mine.lo;
                    if (RW == 4) *reinterpret_cast<uint64_t *>(rec + vi[e] * CS + 8) = mine.hi;
                }
                if (valid) {
                    unsigned short vm = (code != cr[e].get(j)) ? (unsigned short)(1u << j) : (unsigned short)(vo[e] | (1u << j));
                    if (have_ref) {
                        CodeRec mine = cr[e];
                        mine.set(j, code);
                        if (mine.lo == rr[e].lo && (RW == 2 || mine.hi == rr[e].hi)) vm = (unsigned short)(vm | rv[e]);      // known_valid()
                    }
                    valid[vi[e]] = vm;
                    if (vmir) vmir[vi[e] - vmir_lo] = vm;
                }
            }
        }
}
// vectors of a wave in flight in the light routine: the table rows of one vector take 4 (m-1) VGPRs
#define LSQ_LIGHT_LB(M) ((M) <= 8 ? 2 : 1)

#define LSQ_WALK_MAX_NODES 64
struct WalkNodes { int count; int pos0; uint8_t j[LSQ_WALK_MAX_NODES]; };      // kernel argument: the node updates of one launch, in order; pos0 = position of j[0] in the ILS iteration's node sequence (trace counters)


// One-time, per-device opt-in to > 64 KiB of dynamic LDS for one kernel instantiation.  lsq_multi_* runs one host thread
// per device through the launchers, so the "done" flags are guarded (unsynchronised function-local statics).
struct LdsOptIn {
    std::mutex mu;
    bool done[64] = {};
};
template <class Kern>
int optin_lds(LdsOptIn &st, Kern kernel, int bytes) {
    int dev = 0;
    LSQ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(st.mu);
    if (dev < 0 || dev >= 64 || !st.done[dev]) {
        LSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (dev >= 0 && dev < 64) st.done[dev] = true;
    }
    return LSQ_OK;
}


}  // namespace
