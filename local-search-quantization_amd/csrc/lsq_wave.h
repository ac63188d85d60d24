// lsq_wave.h -- wave64 cross-lane helpers shared by the ICM kernels (gfx950 only; not a public header).
#pragma once

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

}  // namespace
