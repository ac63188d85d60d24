// lsq_q16.h -- level arithmetic and LDS placements of the 16-bit FILTERED node-update kernel (lsq_icmq.hip: one block walks all slices of its
// vectors).  gfx950 only; not a public header.  (Round 4's XCD-cooperative variant, schedule 7, lives under the tag r05-schedule7-and-fused-launch.)
#pragma once

#include "lsq_wave.h"

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// LDS addressed by number (address space 3 pointers made from integers)
typedef const u32x4 __attribute__((address_space(3))) lds_cu32x4;
typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef const unsigned short __attribute__((address_space(3))) lds_u16;
typedef char __attribute__((address_space(3))) lds_char;


__device__ inline uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));      // v_pk_add_u16
}
__device__ inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// smallest / middle of three (v_min3_u32 / v_med3_u32): the two smallest of a triple in two instructions
__device__ inline uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ inline uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// two smallest of the union of two (lo <= hi) pairs
__device__ inline void top2_merge(uint32_t &lo, uint32_t &hi, uint32_t olo, uint32_t ohi) {
    const uint32_t nhi = umin3(umax(lo, olo), hi, ohi);
    lo = umin(lo, olo);
    hi = nhi;
}
// quad permutations only (every source lane exists): old = 0 + bound_ctrl lets the compiler fold the permutation into the consuming
// v_min_u32 / v_max_u32 (4 instructions per top2_merge stage instead of 7 with a self-referencing old operand)
template <int CTRL>
__device__ inline uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}


// exact f32 conditioned value of candidate a of node j for vector i (canonical order, encode_icm.jl:84-101), from the f32 unaries
template <int M, int RW>
__device__ inline float q16_exact_value(const float *__restrict__ U, const float *__restrict__ T, int64_t n, int SLF, int j, int64_t i,
                            const uint32_t (&rw)[RW], int a) {
    float s = U[(int64_t)j * n * LSQ_H + ((int64_t)(a / SLF) * n + i) * SLF + (a % SLF)];
    const float *Tj = T + (int64_t)j * M * LSQ_H * LSQ_H;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (k == j) continue;
        const uint32_t bk = (rw[k >> 2] >> (8 * (k & 3))) & 0xffu;
        s = s + Tj[((int64_t)(k * LSQ_H) + bk) * LSQ_H + a];
    }
    return s;
}


// LDS placement of a slice table.  A lane owns CPL = 8 NR candidates of its vector's slice: NR 16-byte chunks, chunk c = r LPV + q (r-th read of
// lane q).  The table is stored as NR planes (plane r = the chunks every lane reads r-th, LPV per row), later planes skewed by 128 bytes:
// the layout tools/ubench_lds.hip measured best for two lanes per vector (NR = 1: plain rows).  Global Tq keeps plain rows.
template <int SLQ, int CPL>
struct WalkqTab {
    static constexpr int NR = CPL / 8, LPV = SLQ / CPL, EPR = SLQ / 8;                      // reads per lane and table / lanes per vector / entries per row
    static constexpr int PLANE_E = LSQ_H * LPV + (NR > 1 ? 8 : 0), TS_E = NR * PLANE_E;     // 16-byte entries per plane / per table
    __device__ static constexpr int entry(int kk, int code, int c) { return kk * TS_E + (c / LPV) * PLANE_E + code * LPV + (c % LPV); }
    static constexpr int lds_entries(int m) { return (m - 1) * TS_E; }
    // vectors per block pass with BPC blocks per CU: the slice table + 10 B per vector within the block's share of the 160 KiB
    static constexpr int pp_for(int m, int bpc, int bytes_per_vector) {
        const int avail = 160 * 1024 / bpc - 768 - lds_entries(m) * 16;
        const int v = avail / bytes_per_vector / 64 * 64;
        return v > 4096 / bpc ? 4096 / bpc : v;
    }
    // Round 5: the block keeps a MIRROR of its vectors' validity words in LDS (2 more bytes per vector) so that the compaction at the head of every node
    // update reads LDS instead of waiting for L2 -- where that costs at most 1/64 of the vectors per pass (plain placement, m = 8: 4032 instead of 4096; m = 16 would drop
    // from 3968 to 3328 and need a second pass per 10^6-vector chunk: no mirror there)
    static constexpr bool mirror(int m, int bpc) { return pp_for(m, bpc, 12) * 64 >= pp_for(m, bpc, 10) * 63; }
    static constexpr int pp(int m, int bpc) { return mirror(m, bpc) ? pp_for(m, bpc, 12) : pp_for(m, bpc, 10); }
};

// Round 5 (late): "rotated rows" placement for the default geometry up to m = 8 (slices of 32 candidates, four lanes of 16 bytes per vector).
// The slice table is stored code-major in 256-byte lines: byte address = group * 65536 + code * 256 + slot * 64 + lane_q * 16, table kk = 4 group + slot
// (group 0: tables 0..3, group 1: tables 4..m-2).  Two things follow.  (i) Code and bank are decoupled: the bank quarter of a read is its SLOT, so when the
// four vectors of a 16-lane group read four different slots the read is conflict-free whatever their codes are -- vector v reads, as the t-th read of a
// group, slot (t + v) mod (tables of the group); integer level sums do not care about the order (the old placement, 64-byte rows indexed by code, cost
// E[max rows per bank quarter] = 2.125 LDS cycles per 16-lane group instead of 1: profiles/r05_ubench_lds.txt).  (ii) The code sits in byte 1 of the address:
// one v_perm_b32 builds an address from the (rotated) code bytes and a lane constant holding the four slot | lane_q bytes -- 9 VALU instructions per item for
// the seven addresses instead of 16 (extract, shift-add).  With 5..7 tables the free slot 3 of group 1 (64 bytes in every 256) holds the smallest keys
// (bestA: word ci at line ci / 16, 16 words per line), so the table costs no more LDS than the plain placement: 128 KiB + 8 bytes per vector.
// Above m = 8 (slices of 16 candidates, two lanes of 16 bytes per vector: 32-byte rows) a line holds EIGHT slots: group 0 = tables 0..7, group 1 = tables 8..m-2
// (at most seven: slot 7 of its lines is free and holds the active list, 16 entries per line); the eight vectors of a 16-lane group read eight different slots.
template <int M>
struct WalkqRot {
    static constexpr int NTAB = M - 1;
    static constexpr int SPL = M <= 8 ? 4 : 8;                                               // slots per 256-byte line
    static constexpr int SLOT_BYTES = 256 / SPL, EPS = SLOT_BYTES / 16;                      // bytes / 16-byte entries per slot (= per table row of a slice)
    static constexpr int NT0 = NTAB < SPL ? NTAB : SPL, NT1 = NTAB - NT0;                    // tables of group 0 / group 1
    static constexpr int NG = NT1 > 0 ? 2 : (NT0 > 0 ? 1 : 0);
    static constexpr int TAB_BYTES = NG * 65536;
    static constexpr int G0_ENTRIES = NT0 * LSQ_H * EPS;                                     // 16-byte entries of group 0 (global and LDS)
    static constexpr bool HOLE = M <= 8 && NT1 > 0;                                          // m <= 8: bestA lives in slot 3 of group 1
    static constexpr bool LHOLE = M > 8 && NT1 > 0;                                          // m > 8: the active list lives in slot 7 of group 1
    static constexpr int HOLE_BYTE0 = 65536 + 256 - SLOT_BYTES;
    static constexpr int pp_for(int bytes_per_vector) {
        const int avail = 160 * 1024 - 768 - TAB_BYTES;
        const int v = avail / bytes_per_vector / 64 * 64;
        return v > 4096 ? 4096 : v;
    }
    static constexpr int KEY_BYTES = HOLE ? 4 : 8;                                           // per vector outside the table: bestB (+ bestA)
    static constexpr int LIST_BYTES = LHOLE ? 0 : 2;                                         // ... and the active list
    // validity mirror (2 bytes per vector, see WalkqTab::mirror): kept where a block pass still holds 10^6 / 256 vectors (BASELINE configs[1] in one pass)
    static constexpr bool mirror() { return M <= 8 && pp_for(KEY_BYTES + LIST_BYTES + 2) * 256 >= 1000000; }
    static constexpr int pp() { return pp_for(KEY_BYTES + LIST_BYTES + (mirror() ? 2 : 0)); }
    static constexpr int lds_bytes() { return TAB_BYTES + pp() * (KEY_BYTES + LIST_BYTES + (mirror() ? 2 : 0)); }
};

template <int N, class F, int I = 0>
__device__ inline void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, F, I + 1>(static_cast<F &&>(f)); }
}

// Row (kk, code) of a slice table in GLOBAL memory, in rows of SLQ levels.  Slices of 32 candidates up to m = 8 are stored as the LDS image of the rotated
// placement without its free slot -- [group][code][slot] -- so that a slice is staged by a straight, fully coalesced copy; so are slices of 16 above m = 8
// (eight slots per line); every other geometry keeps [kk][code].
template <int SLQ>
__host__ __device__ inline int q16_row_index(int m, int kk, int code) {
    if ((SLQ == 32 && m <= 8) || (SLQ == 16 && m > 8)) {
        const int spl = SLQ == 32 ? 4 : 8;
        const int nt0 = m - 1 < spl ? m - 1 : spl, g = kk >= spl ? 1 : 0;
        return g ? nt0 * LSQ_H + code * (m - 1 - nt0) + (kk - spl) : code * nt0 + kk;
    }
    return kk * LSQ_H + code;
}

// which geometry takes the rotated-rows placement
constexpr bool walkq_rot(int m, int slq, int cpl, int nt, int bpc) { return ((m <= 8 && slq == 32) || (m > 8 && slq == 16)) && cpl == 8 && nt == 1024 && bpc == 1; }


}  // namespace
