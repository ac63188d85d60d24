// lsq_icmx.hip -- the 16-bit filtered ICM node update with the SLICES OF A NODE SPREAD OVER THE CUs OF AN XCD (schedule 7; gfx950).
//
// Same arithmetic, same levels, same bound, same exact refinement as icm_walkq_kernel (lsq_icmq.hip) -- and therefore the same codes (reference:
// src/encodings/encode_icm.jl:72-125, lowest index on ties) -- but a different decomposition of the work.  icm_walkq_kernel gives a block ~4000
// vectors and lets it walk all NS slices of every node: NS x 112 KiB of table staged L2 -> LDS per block and node (7.3 GB per launch at cfg2), two
// barriers per slice, and compaction / decide / refinement phases in which the whole block waits for one global round trip each.  Those phases are
// the 35 us floor of a sparse node update and 30 of the 136 us of a dense one (profiles/r03j_walkq_phases.txt).  Here:
//
//   GROUP    the G = NS CUs of a group sit on ONE XCD (role = f(HW_REG_XCC_ID, arrival order)) and own a contiguous range of vectors for the whole
//            launch.  CU s of the group holds slice s of the current node's table in LDS -- staged ONCE per node and CU (G x fewer staging bytes) --
//            and computes, for every active vector of the group, the two smallest keys of its 32 (16) candidates: the PARTIAL keys, 8 bytes per
//            (vector, slice), written to a ring in the XCD's L2.
//   MERGE    after all G CUs have published their partials of a task, each CU merges 1/G of the active vectors: G partials -> the two smallest keys
//            of the node update -> decide / apply / exact refinement exactly as icm_walkq_kernel does.
//   ROLES    no phase ever holds the block: of a block's 16 waves W = 12 are WALKERS (the slice walk, nothing else), two are LISTERS (lister l builds the
//            active lists of the tasks t = l mod 2 from the validity words: 16 bytes per lane and load, 16 loads in flight, bit tricks instead of a scan) and
//            NM = 2 are MERGERS, each owning WHOLE tasks (t = k mod NM) with two register sets so that the loads of the next 64 vectors are issued before the
//            stores of the current ones.  They hand tasks to each other through LDS counters; progress between CUs is two words per CU (partial keys
//            published: atomic max by the LAST walker of a task; results applied: the contiguous prefix of finished merges), polled with sc1 loads.
//   TASKS    a group's range is cut into Q cohorts (<= LCAP active vectors each; at least two, so that the merge of one overlaps the walk of the
//            other); task t = (node t / Q, cohort t % Q).  Dependencies: the walk of (node, cohort) needs the merge of (previous node, cohort) by
//            ALL CUs of the group; nothing ever crosses a group, let alone an XCD.
//   VISIBILITY  everything another CU has written in this launch (partials, records, validity words, progress words) is read with agent-scope
//            relaxed atomic loads (sc1: bypass the CU's L1, served by the XCD's L2 -- which is coherent for the CUs of ONE XCD); writers use plain
//            stores (L1 is write-through), wait for vmcnt(0), then publish a progress word.  No fences.  This is only sound because a group lives on
//            one XCD: the start barrier below checks that every XCD received exactly 32 blocks (it is also the co-residency check a persistent
//            kernel needs); if that fails within ~0.1 s NOTHING has been touched, sync->gate = 2, and the filtered walk kernel that the host enqueues
//            right behind this launch (predicated on that word) does the launch's work instead.
// Every spin is bounded (give-up code in sync->abort and in the call's error word, reported by the host as an error).
//
// MEASURED (round 4): bit-exact, and slower than icm_walkq_kernel -- 53.5 vs 41.6 ms of ICM per cfg2 step in the same build.  The walkers' own time sums
// to what icm_walkq_kernel's slice loops take; the lists, the partial keys' trip through L2 and the merges cost as many issue slots as the phases they replace, and
// every dependent global round trip on a CU whose walkers keep the memory pipe full costs 3 - 5 us.  Kept in the tuning build as an independent implementation.
#include <stdlib.h>

#include <type_traits>

#include "lsq_q16.h"

#ifdef LSQ_TUNING
__device__ unsigned long long *g_xs_dbg = nullptr;      // tools only: [task][16] clock stamps (wall_clock64, 100 MHz) of ONE block (group 0, slice 0) of the latest launch
#define XS_STAMP(t, k) do { if (dbgp && lane == 0 && (t) < 1024) dbgp[(size_t)(t) * 16 + (k)] = wall_clock64(); } while (0)
#define XS_NOTE(t, k, v) do { if (dbgp && lane == 0 && (t) < 1024) dbgp[(size_t)(t) * 16 + (k)] = (unsigned long long)(v); } while (0)
#else
#define XS_STAMP(t, k) do {} while (0)
#define XS_NOTE(t, k, v) do {} while (0)
#endif

namespace {

constexpr int XS_R = 4;                          // ring of partial-key slots per group (tasks in flight between walk and merge)
constexpr unsigned XS_SPIN_START = 1u << 16;     // polls (~1.5 us each) of the start barrier: ~0.1 s
constexpr unsigned XS_SPIN_LIMIT = 1u << 21;     // polls of every later wait

struct XsSync {                                   // device memory, zeroed by the host before EVERY launch
    unsigned xslot[8];                            // blocks arrived per XCD
    unsigned arrive;                              // blocks arrived in total
    unsigned gate;                                // the start barrier's ONE verdict (set by a single compare-and-swap): 1 = go, 2 = not this time -- nothing was
                                                  // touched and the predicated fallback launch does the work (census wrong, or a block waited too long)
    unsigned abort;                               // != 0: a wait AFTER the gate gave up (code): the launch's results are invalid
    unsigned pad[5];
    unsigned progb[32][16];                       // [group][CU]: tasks whose partial keys are complete in L2
    unsigned progc[32][16];                       // [group][CU]: tasks whose results (records, validity words) are complete in L2
};

struct XsArgs {
    const float *U; const uint16_t *Uq; const uint16_t *Tq; const float *T;
    uint8_t *rec; unsigned short *valid; const uint8_t *ref_rec; const unsigned short *ref_valid;
    const lsq_q16_params *P; const unsigned short *qflag;
    unsigned long long *part; XsSync *sync; unsigned long long *active_total;
    unsigned *err;                                // per CALL (zeroed by the host at its start): [0] = largest give-up code, [1] = launches the gate turned away
    int64_t n;
    int per_group, clen, Q, use_skip, SLF;
};

// agent-scope relaxed load: global_load ... sc1 -- never served by this CU's L1
template <class T> __device__ inline T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline unsigned lds_load(unsigned *w) { return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_store(unsigned *w, unsigned v) { __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_add(unsigned *w, unsigned v) { __hip_atomic_fetch_add(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// control words of a block (LDS)
enum { XC_LIST0 = 0, XC_LIST1, XC_NACT0, XC_NACT1, XC_NACT2, XC_NACT3, XC_WALK0, XC_WALK1, XC_TABBAR, XC_MDONE0, XC_MDONE1, XC_MDONE2, XC_MDONE3, XC_CDONE, XC_ABORT, XC_GROUP, XC_SLICE, XC_WORDS = 24 };

struct XsCtl {
    unsigned *c;            // LDS control words
    XsSync *sync;
    unsigned *err;
    __device__ inline void give_up(unsigned code) const {
        if ((threadIdx.x & 63) == 0) {
            lds_store(c + XC_ABORT, code);
            atomicCAS(&sync->abort, 0u, code);
            atomicMax(err, code);
        }
    }
    // wait until the LDS word reaches `target` (wave-uniform); false = abort
    __device__ inline bool wait_lds(int word, unsigned target, unsigned code) const {
        unsigned spins = 0;
        while (lds_load(c + word) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 31u) == 0) {
                if (lds_load(c + XC_ABORT) != 0u) return false;
                if (spins > XS_SPIN_LIMIT * 8u) { give_up(code); return false; }
            }
        }
        return true;
    }
    // wait until the G progress words of the group's row all reach `target`
    __device__ inline bool wait_prog(const unsigned *row, int G, unsigned target, unsigned code) const {
        const int lane = threadIdx.x & 63;
        unsigned spins = 0;
        for (;;) {
            const unsigned v = lane < G ? ld_l2(row + lane) : 0xffffffffu;
            if (__ballot(v < target) == 0ull) return true;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 15u) == 0) {
                if (lds_load(c + XC_ABORT) != 0u) return false;
                if (ld_l2(&sync->abort) != 0u) { if (lane == 0) lds_store(c + XC_ABORT, 99u); return false; }
                if (spins > XS_SPIN_LIMIT) { give_up(code); return false; }
            }
        }
    }
};

// ---- exact refinement of a merger wave's ambiguous vectors ----------------------------------------------------------------------------------
// The routine of lsq_icmq.hip (q16_refine: 16 lanes per vector, one dependent global round trip in the common case), fed from records that carry
// everything the bookkeeping needs (the merger loaded it with the partial keys): {li | a1 << 16 | a2 << 24, limit, record words, vo | rv << 16,
// reference record words}.  li = the vector's offset in the group's range.
template <int M, int SLQ>
__device__ inline int xs_refine(const XsArgs &A, int j, int64_t gb, const uint32_t *arec, int namb, bool have_ref) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int RW = CS / 4;
    constexpr int AREC = 3 + 2 * RW;
    constexpr int NS = LSQ_H / SLQ;
    constexpr int TAB = (M - 1) * LSQ_H * (SLQ / 8);
    const int64_t n = A.n;
    const int SLF = A.SLF;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4, t16 = lane & 15;
    const int sl = (16 * t16) / SLQ, off = (16 * t16) % SLQ;              // slice / offset (in candidates) of this lane's 16 candidates
    const uint16_t *__restrict__ Uqj = A.Uq + (int64_t)j * n * LSQ_H;
    const uint16_t *__restrict__ Tqj = A.Tq + ((int64_t)j * NS + sl) * TAB * 8 + off;
    const float *__restrict__ Tj = A.T + (int64_t)j * M * LSQ_H * LSQ_H;
    const float *__restrict__ U = A.U;
    int nexact = 0;
    for (int r0 = 0; r0 < namb; r0 += 4) {
        const int r = r0 + grp;
        const bool act = r < namb;
        const uint32_t *ar = arec + (act ? r : r0) * AREC;
        const uint32_t key = ar[0], limit = ar[1];
        uint32_t rw[RW], rr[RW];
#pragma unroll
        for (int w2 = 0; w2 < RW; ++w2) { rw[w2] = ar[2 + w2]; rr[w2] = ar[3 + RW + w2]; }
        const uint32_t vv = ar[2 + RW];
        const unsigned short vo = (unsigned short)(vv & 0xffffu), rv = (unsigned short)(vv >> 16);
        const int a1 = (int)((key >> 16) & 0xffu), a2 = (int)(key >> 24);
        const int64_t i = gb + (int64_t)(key & 0xffffu);
        // ---- the one round trip: unary levels, table levels, speculative exact terms (one per lane)
        const u32x4 *up = reinterpret_cast<const u32x4 *>(Uqj + ((int64_t)sl * n + i) * SLQ + off);
        u32x4 s0 = up[0], s1 = up[1];
        float term = 0.0f;                                                // lanes 0..M-1: terms of a1, lanes 8..8+M-1: terms of a2 (M <= 8); M > 8: slow path only
        if (M <= 8) {
            const int cand = (t16 < 8) ? a1 : a2, tt = t16 & 7;           // term tt: 0 = unary, q >= 1 = table of the q-th conditioning codebook
            if (tt == 0) term = U[(int64_t)j * n * LSQ_H + ((int64_t)(cand / SLF) * n + i) * SLF + (cand % SLF)];
            else if (tt < M) {
                const int k = (tt - 1) + ((tt - 1) >= j ? 1 : 0);
                const uint32_t bk = (rw[k >> 2] >> (8 * (k & 3))) & 0xffu;
                term = Tj[((int64_t)(k * LSQ_H) + bk) * LSQ_H + cand];
            }
        }
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) {
            const int k = kk + (kk >= j ? 1 : 0);
            const uint32_t bk = (rw[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const u32x4 *tp = reinterpret_cast<const u32x4 *>(Tqj + (int64_t)q16_row_index<SLQ>(M, kk, (int)bk) * SLQ);
            const u32x4 b0 = tp[0], b1 = tp[1];
            s0.x = pk_add_u16(s0.x, b0.x); s0.y = pk_add_u16(s0.y, b0.y); s0.z = pk_add_u16(s0.z, b0.z); s0.w = pk_add_u16(s0.w, b0.w);
            s1.x = pk_add_u16(s1.x, b1.x); s1.y = pk_add_u16(s1.y, b1.y); s1.z = pk_add_u16(s1.z, b1.z); s1.w = pk_add_u16(s1.w, b1.w);
        }
        uint32_t mask = 0;                                                // bit p: candidate 16 t16 + p survives
#define LSQ_SURV(W, B) mask |= (((W) & 0xffffu) <= limit ? 1u : 0u) << (B); mask |= (((W) >> 16) <= limit ? 1u : 0u) << ((B) + 1);
        LSQ_SURV(s0.x, 0) LSQ_SURV(s0.y, 2) LSQ_SURV(s0.z, 4) LSQ_SURV(s0.w, 6) LSQ_SURV(s1.x, 8) LSQ_SURV(s1.y, 10) LSQ_SURV(s1.z, 12) LSQ_SURV(s1.w, 14)
#undef LSQ_SURV
        if (!act) mask = 0;
        float bv = __builtin_inff();
        int bi = 0x7fffffff;
        if (M <= 8) {
            // canonical sums of a1 (lanes 0..7 of the group) and a2 (lanes 8..15): ((u + t1) + t2) + ...  in ascending k
            float e = __shfl(term, (lane & ~7), 64);
#pragma unroll
            for (int q = 1; q < M; ++q) e = e + __shfl(term, (lane & ~7) + q, 64);
            const float e1 = __shfl(e, lane & ~15, 64), e2 = __shfl(e, (lane & ~15) + 8, 64);
            if (a1 / 16 == t16) mask &= ~(1u << (a1 % 16));              // a1 and a2 are survivors by construction: ranked here
            if (a2 / 16 == t16) mask &= ~(1u << (a2 % 16));
            bv = e1; bi = a1;
            if (e2 < bv || (e2 == bv && a2 < bi)) { bv = e2; bi = a2; }
            if (act && t16 == 0) nexact += 2;
        }
        while (mask) {                                                    // third survivors (or every survivor when M > 8): one more trip
            const int a = 16 * t16 + __builtin_ctz(mask);
            mask &= mask - 1;
            const float ev = q16_exact_value<M, RW>(U, A.T, n, SLF, j, i, rw, a);
            ++nexact;
            if (ev < bv || (ev == bv && a < bi)) { bv = ev; bi = a; }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 16);
            const int oi = __shfl_xor(bi, o, 16);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (act && t16 == 0) {
            const uint8_t c8 = (uint8_t)bi;
            const uint8_t old = (uint8_t)(rw[j >> 2] >> (8 * (j & 3)));
            if (c8 != old) store_code<CS>(A.rec, i, j, c8, rw);
            if (A.valid) {
                unsigned short vm = (c8 != old) ? (unsigned short)(1u << j) : (unsigned short)(vo | (1u << j));
                if (have_ref) {
                    bool same = true;
#pragma unroll
                    for (int w2 = 0; w2 < RW; ++w2) {
                        uint32_t mine = rw[w2];
                        if (w2 == (j >> 2)) mine = (mine & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)c8 << (8 * (j & 3)));
                        same = same && (mine == rr[w2]);
                    }
                    if (same) vm = (unsigned short)(vm | rv);
                }
                A.valid[i] = vm;
            }
        }
    }
    return nexact;
}

// One vector in full f32 by the whole wave (a unary outside the sampled level range): light_update's arithmetic with the bookkeeping words handed
// in (wave-uniform) instead of loaded.
template <int M, int CS>
__device__ inline void xs_f32_update(const XsArgs &A, int j, int64_t vi, const uint32_t (&rw)[CS / 4], uint32_t vo, const uint32_t (&rr)[CS / 4], uint32_t rv,
                                     bool have_ref, int lane) {
    constexpr int RW = CS / 4;
    const int SL = A.SLF, LPV = SL / 4;
    const float *__restrict__ Usj = A.U + (int64_t)j * A.n * LSQ_H;
    const float *__restrict__ Tj = A.T + (int64_t)j * M * LSQ_H * LSQ_H;
    f32x4 s = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Usj + ((int64_t)(lane / LPV) * A.n + vi) * SL) + (lane % LPV));
    f32x4 c[M > 1 ? M - 1 : 1];
#pragma unroll
    for (int kk = 0; kk < M - 1; ++kk) {
        const int k = kk + (kk >= j ? 1 : 0);
        const uint32_t bk = (rw[k >> 2] >> (8 * (k & 3))) & 0xffu;
        c[kk] = reinterpret_cast<const f32x4 *>(Tj + ((int64_t)(k * LSQ_H) + bk) * LSQ_H)[lane];
    }
#pragma unroll
    for (int kk = 0; kk < M - 1; ++kk) s = s + c[kk];                     // ascending k, plain f32 adds
    const uint8_t code = (uint8_t)wave_first_argmin(s, lane);
    if (lane == 0) {
        const uint8_t old = (uint8_t)(rw[j >> 2] >> (8 * (j & 3)));
        if (code != old) store_code<CS>(A.rec, vi, j, code, rw);
        if (A.valid) {
            unsigned short vm = (code != old) ? (unsigned short)(1u << j) : (unsigned short)(vo | (1u << j));
            if (have_ref) {
                bool same = true;
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) {
                    uint32_t mine = rw[w2];
                    if (w2 == (j >> 2)) mine = (mine & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)code << (8 * (j & 3)));
                    same = same && (mine == rr[w2]);
                }
                if (same) vm = (unsigned short)(vm | rv);
            }
            A.valid[vi] = vm;
        }
    }
}

template <int M, int SLQ, int NMERGE = 2>
struct XsCfg {
    static constexpr int CS = (M <= 8) ? 8 : 16;
    static constexpr int RW = CS / 4;
    static constexpr int NS = LSQ_H / SLQ;                 // slices of a node = CUs of a group
    static constexpr int G = NS;
    static constexpr int GPX = 32 / G;                     // groups per XCD
    static constexpr int NG = 8 * GPX;
    static constexpr int NL = 2, NM = NMERGE, W = 16 - NL - NM;      // waves: W walkers, NL listers (lister l builds the lists of the tasks t = l mod NL), NM mergers
    static constexpr int NT = 64 * (W + NL + NM);
    static constexpr int DEPTH = (M <= 8) ? 3 : 2;
    static constexpr int LCAP = (M <= 8) ? 7680 : 6144;    // active vectors per task (list entries)
    static constexpr int SCAP = LCAP / G;                  // ... of which this CU merges at most this many (its share)
    static constexpr int AREC = 3 + 2 * RW;
    static constexpr int ACAP = ((M <= 8) ? 320 : 192) / NM;       // ambiguous-vector records per merger wave
    using TL = WalkqTab<SLQ, 8>;
    static constexpr int LTAB = TL::lds_entries(M);        // 16-byte entries
    static constexpr int OFF_LIST = LTAB * 16;
    static constexpr int OFF_SHARE = OFF_LIST + 2 * LCAP * 2;   // ring of XS_R share lists: the list buffer is free as soon as the walkers are done with it
    static constexpr int OFF_AREC = OFF_SHARE + XS_R * SCAP * 2;
    static constexpr int OFF_CTL = OFF_AREC + NM * ACAP * AREC * 4;
    static constexpr int OFF_TRACE = OFF_CTL + XC_WORDS * 4;   // per merger wave: recomputed node updates by position in the ILS iteration
    static constexpr int LDS_BYTES = OFF_TRACE + NM * LSQ_WALK_TRACE * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "slice table + lists + records must fit the 160 KiB LDS");
    static_assert(NT == 1024, "16 waves");
};

template <int M, int SLQ, int NMERGE>
__global__ __launch_bounds__(1024) void icm_xs_kernel(const XsArgs A, const WalkNodes nodes) {
    using C = XsCfg<M, SLQ, NMERGE>;
    using TL = typename C::TL;
    constexpr int CS = C::CS, RW = C::RW, NS = C::NS, G = C::G, GPX = C::GPX, W = C::W, NL = C::NL, NM = C::NM, DEPTH = C::DEPTH, LCAP = C::LCAP;
    static_assert(NL == 2, "one lister per list buffer");
    constexpr int AREC = C::AREC, ACAP = C::ACAP, SCAP = C::SCAP;
    constexpr int LPV = SLQ / 8, VPW = 64 / LPV, EPR = SLQ / 8;
    constexpr int TAB = (M - 1) * LSQ_H * EPR;             // 16-byte entries of one slice table
    constexpr int CW = (M - 1 + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_xs[];
    u32x4 *tab = reinterpret_cast<u32x4 *>(lds_xs);
    unsigned short *list0 = reinterpret_cast<unsigned short *>(lds_xs + C::OFF_LIST);
    unsigned short *share0 = reinterpret_cast<unsigned short *>(lds_xs + C::OFF_SHARE);
    uint32_t *arec0 = reinterpret_cast<uint32_t *>(lds_xs + C::OFF_AREC);
    unsigned *ctl = reinterpret_cast<unsigned *>(lds_xs + C::OFF_CTL);
    XsSync *sync = A.sync;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // ---- start barrier = role assignment = co-residency and placement check: every XCD must have received exactly 32 blocks.
    if (threadIdx.x < XC_WORDS) ctl[threadIdx.x] = 0u;
    __syncthreads();
    if (wave == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        unsigned slot = 0, seq = 0;
        if (lane == 0) {
            slot = atomicAdd(&sync->xslot[xcc], 1u);
            __threadfence();                                                          // the census word before the arrival count (both are L2 / device atomics)
            seq = atomicAdd(&sync->arrive, 1u);
        }
        slot = (unsigned)__builtin_amdgcn_readfirstlane((int)slot);
        seq = (unsigned)__builtin_amdgcn_readfirstlane((int)seq);
        if (seq == gridDim.x - 1u) {                                                  // the last arriver judges the census: 32 blocks on every XCD
            const unsigned v = lane < 8 ? ld_l2(&sync->xslot[lane]) : 32u;
            const bool fine = gridDim.x == 256u && __ballot(v != 32u) == 0ull;
            if (lane == 0 && atomicCAS(&sync->gate, 0u, fine ? 1u : 2u) == 0u && !fine) atomicAdd(A.err + 1, 1u);
        }
        unsigned gate = 0, spins = 0;
        for (;;) {
            gate = ld_l2(&sync->gate);
            if (gate != 0u) break;
            if (++spins > XS_SPIN_START) {                                            // not all blocks are resident (another process' kernels on the device?)
                if (lane == 0 && atomicCAS(&sync->gate, 0u, 2u) == 0u) atomicAdd(A.err + 1, 1u);      // one verdict for everybody: whichever swap came first
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) {
            ctl[XC_ABORT] = gate == 1u ? 0u : 1u;
            ctl[XC_GROUP] = xcc * GPX + slot / G;
            ctl[XC_SLICE] = slot % G;
        }
    }
    __syncthreads();
    if (ctl[XC_ABORT] != 0u) return;
    const int group = (int)ctl[XC_GROUP], slice = (int)ctl[XC_SLICE];
    const XsCtl X{ctl, sync, A.err};

    const int64_t n = A.n;
    const int Q = A.Q, clen = A.clen;
    const int64_t gb = (int64_t)group * A.per_group;                                 // first vector of the group's range
    const int glen = (int)((n - gb) < 0 ? 0 : ((n - gb) > A.per_group ? A.per_group : (n - gb)));
    if (glen == 0) return;                                                           // the whole group agrees
    const int ntasks = nodes.count * Q;
    unsigned *progb = &sync->progb[group][0], *progc = &sync->progc[group][0];
    unsigned long long *part_g = A.part + (size_t)group * XS_R * G * (size_t)clen;   // [ring slot][slice][position]
    const bool have_ref = A.ref_rec && A.ref_valid;
#ifdef LSQ_TUNING
    unsigned long long *dbgp = (g_xs_dbg && group == 0 && slice == 0) ? g_xs_dbg : nullptr;
#endif

    if (wave < W) {
        // =========================================================== WALKERS ===========================================================
        const int v = lane / LPV, q = lane % LPV;
        constexpr int step = W * VPW;
        struct Item { u32x4 u; uint32_t r[RW]; };
        unsigned nstaged = 0;
        uint32_t sel[CW > 0 ? CW : 1];
        for (int t = 0; t < ntasks; ++t) {
            const int jn = t / Q;
            const int j = nodes.j[jn];
            if (t - jn * Q == 0) {
                // a new node: every walker must be done with the old table, then slice `slice` of node j's table goes to LDS (once per node and CU)
                if (t > 0 && !X.wait_lds(XC_WALK0 + ((t - 1) & 1), (unsigned)(W * ((t - 1) / 2 + 1)), 10u)) return;
                if (wave == 0) XS_STAMP(t, 3);
                const u32x4 *src = reinterpret_cast<const u32x4 *>(A.Tq) + ((int64_t)j * NS + slice) * TAB;
                for (int e = wave * 64 + lane; e < TAB; e += W * 64)
                    tab[TL::entry(e / (LSQ_H * EPR), (e / EPR) % LSQ_H, e % EPR)] = src[q16_row_index<SLQ>(M, e / (LSQ_H * EPR), (e / EPR) % LSQ_H) * EPR + e % EPR];
                if (lane == 0) lds_add(ctl + XC_TABBAR, 1u);
                ++nstaged;
                if (!X.wait_lds(XC_TABBAR, (unsigned)W * nstaged, 11u)) return;
                if (wave == 0) XS_STAMP(t, 4);
#pragma unroll
                for (int w = 0; w < CW; ++w) {
                    uint32_t sv = 0;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int kk = 4 * w + tt;
                        const int k = kk + (kk >= j ? 1 : 0);
                        sv |= (uint32_t)((kk < M - 1 ? k - 4 * w : 0) & 7) << (8 * tt);
                    }
                    sel[w] = sv;
                }
            }
            if (wave == 0) XS_STAMP(t, 0);
            if (!X.wait_lds(XC_LIST0 + (t & 1), (unsigned)(t + 1), 12u)) return;
            if (wave == 0) XS_STAMP(t, 1);
            const int nact = (int)ctl[XC_NACT0 + (t % XS_R)];
            const int ipw = (wave * VPW < nact) ? (nact - wave * VPW + step - 1) / step : 0;
            if (ipw > 0) {
                const unsigned short *list = list0 + (t & 1) * LCAP;
                const int cq = t - jn * Q;
                (void)cq;
                const char *ub = reinterpret_cast<const char *>(A.Uq + (int64_t)j * n * LSQ_H + ((int64_t)slice * n + gb) * SLQ);
                const char *rb = reinterpret_cast<const char *>(A.rec + gb * CS);
                unsigned long long *pslot = part_g + ((size_t)(t % XS_R) * G + slice) * (size_t)clen;
                auto load_item = [&](Item &it, int kk) {
                    int ci = wave * VPW + kk * step + v;
                    ci = ci < nact ? ci : nact - 1;
                    const uint32_t li = list[ci];
                    it.u = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ub + li * (uint32_t)(SLQ * 2) + (uint32_t)q * 16u));
                    const unsigned long long *rp = reinterpret_cast<const unsigned long long *>(rb + li * (uint32_t)CS);
#pragma unroll
                    for (int w = 0; w < RW; w += 2) {
                        const unsigned long long x = ld_l2(rp + w / 2);      // written by another CU's merger at the previous node: never from L1
                        it.r[w] = (uint32_t)x;
                        it.r[w + 1] = (uint32_t)(x >> 32);
                    }
                };
                auto compute = [&](const Item &cur, int kk) {
                    u32x4 s = cur.u;
                    uint32_t code[M > 1 ? M - 1 : 1];
#pragma unroll
                    for (int w = 0; w < CW; ++w) {
                        const uint32_t hiw = (w + 1 < RW) ? cur.r[w + 1] : 0u;
                        const uint32_t cw = __builtin_amdgcn_perm(hiw, cur.r[w], sel[w]);
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const int k2 = 4 * w + tt;
                            if (k2 < M - 1) {
                                if (tt == 0) asm("v_and_b32 %0, 0xff, %1" : "=v"(code[k2]) : "v"(cw));
                                else code[k2] = (cw >> (8 * tt)) & 0xffu;
                            }
                        }
                    }
                    // packed 16-bit level sums never carry (lsq_q16_node::hiq): plain 32-bit three-operand adds, two table rows per instruction
#pragma unroll
                    for (int kk2 = 0; kk2 < M - 1; kk2 += 2) {
                        const u32x4 ra = tab[kk2 * TL::TS_E + (int)code[kk2] * LPV + q];
                        if (kk2 + 1 < M - 1) {
                            const u32x4 rb2 = tab[(kk2 + 1) * TL::TS_E + (int)code[kk2 + 1 < M - 1 ? kk2 + 1 : kk2] * LPV + q];
                            s.x = s.x + ra.x + rb2.x; s.y = s.y + ra.y + rb2.y; s.z = s.z + ra.z + rb2.z; s.w = s.w + ra.w + rb2.w;
                        } else {
                            s.x += ra.x; s.y += ra.y; s.z += ra.z; s.w += ra.w;
                        }
                    }
                    constexpr uint32_t HI = 0xffff0000u;
                    const uint32_t k0 = (s.x << 16), k1 = (s.x & HI) | 1u;
                    const uint32_t k2 = (s.y << 16) | 2u, k3 = (s.y & HI) | 3u;
                    const uint32_t k4 = (s.z << 16) | 4u, k5 = (s.z & HI) | 5u;
                    const uint32_t k6 = (s.w << 16) | 6u, k7 = (s.w & HI) | 7u;
                    uint32_t l0 = umin3(k0, k1, k2), h0 = umed3(k0, k1, k2);
                    const uint32_t lb = umin3(k3, k4, k5), hb = umed3(k3, k4, k5);
                    const uint32_t lc = umin(k6, k7), hc = umax(k6, k7);
                    top2_merge(l0, h0, lb, hb);
                    top2_merge(l0, h0, lc, hc);
                    const uint32_t base = (uint32_t)(SLQ * slice) + 8u * (uint32_t)q;
                    l0 += base; h0 += base;                                          // candidate < 256: never carries into the level
                    if (LPV >= 2) top2_merge(l0, h0, dpp_u32<DPP_XOR1>(l0), dpp_u32<DPP_XOR1>(h0));
                    if (LPV >= 4) top2_merge(l0, h0, dpp_u32<DPP_XOR2>(l0), dpp_u32<DPP_XOR2>(h0));
                    const int ci = wave * VPW + kk * step + v;
                    if ((q == 0) & (ci < nact)) pslot[ci] = (unsigned long long)l0 | ((unsigned long long)h0 << 32);
                };
                Item buf[DEPTH];
#pragma unroll
                for (int e = 0; e < DEPTH; ++e) load_item(buf[e], e);
                int kk = 0;
                for (; kk + DEPTH <= ipw; kk += DEPTH) {
#pragma unroll
                    for (int e = 0; e < DEPTH; ++e) { compute(buf[e], kk + e); load_item(buf[e], kk + e + DEPTH); }
                }
#pragma unroll
                for (int e = 0; e < DEPTH - 1; ++e)
                    if (kk + e < ipw) compute(buf[e], kk + e);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's partial keys are in L2
            if (wave == 0) XS_STAMP(t, 2);
            if (lane == 0) {
                const unsigned before = __hip_atomic_fetch_add(ctl + XC_WALK0 + (t & 1), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                // the LAST walker of the task tells the group: this CU's partial keys of task t are in L2 (a maximum: stores of two waves may overtake)
                if (before + 1u == (unsigned)(W * (t / 2 + 1))) { atomicMax(progb + slice, (unsigned)(t + 1)); XS_STAMP(t, 9); }
            }
        }
        return;
    }

    __builtin_amdgcn_s_setprio(3);      // listers and mergers: little work on the critical path -- ahead of the walkers in the SIMD's arbitration
    if (wave < W + NL) {
        // =========================================================== LISTERS ===========================================================
        // Lister l owns list buffer l and the tasks t = l (mod 2).  It scans the cohort's validity words -- 16 bytes (8 vectors) per lane and load,
        // NB loads in flight: one wave has to stream 2 bytes per vector of the whole cohort out of L2 -- and appends the active vectors in ascending order.
        const int use_skip = A.use_skip && A.valid;
        for (int t = wave - W; t < ntasks; t += NL) {
            const int jn = t / Q, cq = t - jn * Q;
            const int j = nodes.j[jn];
            XS_STAMP(t, 5);
            int need = (jn > 0) ? t - Q + 1 : 0;                                      // the cohort's results of the previous node, from every CU of the group
            if (t - XS_R + 1 > need) need = t - XS_R + 1;                             // ... and the ring slot of the partial keys must be free
            if (need > 0 && !X.wait_prog(progc, G, (unsigned)need, 20u)) return;
            if (t >= 2 && !X.wait_lds(XC_WALK0 + (t & 1), (unsigned)(W * ((t - 2) / 2 + 1)), 21u)) return;      // the list buffer: the walkers are done with task t - 2
            if (t >= XS_R && !X.wait_lds(XC_CDONE, (unsigned)(t - XS_R + 1), 22u)) return;                      // the share slot: this CU's mergers are done with task t - XS_R
            XS_STAMP(t, 6);
            unsigned short *list = list0 + (t & 1) * LCAP;
            const int clo = cq * clen, chi = (clo + clen < glen) ? clo + clen : glen;
            int cnt = 0;
            constexpr int NB = 16;                                                    // 8192 validity words per round trip
            for (int base = clo; base < chi; base += 512 * NB) {
                unsigned long long w[NB][2];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int idx0 = base + b * 512 + 8 * lane;
                    w[b][0] = 0ull; w[b][1] = 0ull;
                    if (use_skip) {
                        const unsigned long long *vp = reinterpret_cast<const unsigned long long *>(A.valid + gb + idx0);
                        if (idx0 < chi) w[b][0] = ld_l2(vp);
                        if (idx0 + 4 < chi) w[b][1] = ld_l2(vp + 1);
                    }
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int idx0 = base + b * 512 + 8 * lane;
                    if (base + b * 512 >= chi) continue;                              // wave-uniform
                    // bit j of the lane's 8 validity words, two per 32-bit register: vector e sits at bit (e >> 1) + 16 (e & 1) of `set`
                    constexpr uint32_t M1 = 0x00010001u;
                    const uint32_t set = (((uint32_t)w[b][0] >> j) & M1) | ((((uint32_t)(w[b][0] >> 32) >> j) & M1) << 1) |
                                         ((((uint32_t)w[b][1] >> j) & M1) << 2) | ((((uint32_t)(w[b][1] >> 32) >> j) & M1) << 3);
                    int nin = chi - idx0;
                    nin = nin < 0 ? 0 : (nin > 8 ? 8 : nin);                           // vectors of this lane inside the cohort
                    const uint32_t inr = ((1u << ((nin + 1) >> 1)) - 1u) | (((1u << (nin >> 1)) - 1u) << 16);
                    uint32_t act = ~set & inr;
                    const int c = __builtin_popcount(act);                            // 0 .. 8
                    int below = 0, total = 0;
#pragma unroll
                    for (int bit = 0; bit < 4; ++bit) {
                        const unsigned long long mk = __ballot((c >> bit) & 1);
                        below += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u)) << bit;
                        total += __builtin_popcountll(mk) << bit;
                    }
                    int pos = cnt + below;
                    while (act) {                                                     // the lane's active vectors, in a fixed (not ascending) order: every
                        const int bp = __builtin_ctz(act);                            // CU of the group builds the identical list
                        act &= act - 1u;
                        list[pos++] = (unsigned short)(idx0 + (((bp & 15) << 1) | (bp >> 4)));
                    }
                    cnt += total;
                }
            }
            {   // this CU's share of the task's merge work: its own copy, so that the list buffer is free once the walkers are done
                const int slo = (int)(((int64_t)cnt * slice) / G), shi = (int)(((int64_t)cnt * (slice + 1)) / G);
                unsigned short *sh = share0 + (t % XS_R) * SCAP;
                for (int e = lane; e < shi - slo; e += 64) sh[e] = list[slo + e];
            }
            XS_STAMP(t, 7);
            XS_NOTE(t, 8, cnt);
            if (lane == 0) {                                                         // the list entries above, the count, then the flag: one wave, in order
                ctl[XC_NACT0 + (t % XS_R)] = (unsigned)cnt;
                lds_store(ctl + XC_LIST0 + (t & 1), (unsigned)(t + 1));
            }
        }
        return;
    }

    {
        // =========================================================== MERGERS ===========================================================
        const int mk = wave - W - NL;                                                 // 0 .. NM-1
        // Merger mk owns the tasks t = mk (mod NM) WHOLE (this CU's share of them): NM merges are in flight at once, each several global round
        // trips long (a round trip on a CU whose walkers stream costs 3 - 5 us), so the mergers keep pace with the walkers by throughput, not latency.
        uint32_t *arec = arec0 + mk * ACAP * AREC;
        unsigned *trace = reinterpret_cast<unsigned *>(lds_xs + C::OFF_TRACE) + mk * LSQ_WALK_TRACE;
        for (int e = lane; e < LSQ_WALK_TRACE; e += 64) trace[e] = 0u;
        unsigned st_nodes = 0, st_tasks = 0, st_amb = 0, st_exact = 0, st_f32 = 0;
        for (int t = mk; t < ntasks; t += NM) {
            const int jn = t / Q;
            const int j = nodes.j[jn];
            if (!X.wait_prog(progb, G, (unsigned)(t + 1), 31u)) return;              // every CU's partial keys of task t
            XS_STAMP(t, 10);
            const int nact = (int)lds_load(ctl + XC_NACT0 + (t % XS_R));
            const unsigned short *share = share0 + (t % XS_R) * SCAP;
            const int slo = (int)(((int64_t)nact * slice) / G), shi = (int)(((int64_t)nact * (slice + 1)) / G);      // this CU's share of the task
            const unsigned long long *pslot = part_g + (size_t)(t % XS_R) * G * (size_t)clen;
            const int window = A.P->node[j].window;
            int namb = 0;
            st_nodes += (unsigned)(shi - slo);
            if (lane == 0) trace[(nodes.pos0 + jn) & (LSQ_WALK_TRACE - 1)] += (unsigned)(shi - slo);
            if (slice == 0 && nact > 0) st_tasks += 1u;
            // Two register sets (index E is a compile-time constant everywhere: plain registers, no private memory): the loads of the next 64 vectors
            // are issued BEFORE the stores of the current ones (vmcnt counts both, in order), so a round trip overlaps the previous set's work.
            bool s_on[2];
            uint32_t s_li[2], s_rw[2][RW], s_rr[2][RW], s_vo[2], s_rv[2], s_qf[2];
            unsigned long long s_pk[2][G];
            auto mload = [&](auto EC, int p0) {
                constexpr int E = decltype(EC)::value;
                const int p = p0 + lane;
                s_on[E] = p < shi;
                s_li[E] = share[(s_on[E] ? p : slo) - slo];
                const int64_t vi = gb + s_li[E];
                s_vo[E] = 0; s_rv[E] = 0; s_qf[E] = 0;
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) { s_rw[E][w2] = 0; s_rr[E][w2] = 0; }
#pragma unroll
                for (int s2 = 0; s2 < G; ++s2) s_pk[E][s2] = ~0ull;
                if (s_on[E]) {
#pragma unroll
                    for (int s2 = 0; s2 < G; ++s2) s_pk[E][s2] = ld_l2(pslot + (size_t)s2 * clen + p);
                    const unsigned long long *rp = reinterpret_cast<const unsigned long long *>(A.rec + vi * CS);
#pragma unroll
                    for (int w2 = 0; w2 < RW; w2 += 2) {
                        const unsigned long long x = ld_l2(rp + w2 / 2);
                        s_rw[E][w2] = (uint32_t)x; s_rw[E][w2 + 1] = (uint32_t)(x >> 32);
                    }
                    if (A.valid) s_vo[E] = ld_l2(A.valid + vi);
                    s_qf[E] = A.qflag[vi];
                    if (have_ref) {
#pragma unroll
                        for (int w2 = 0; w2 < RW; ++w2) s_rr[E][w2] = reinterpret_cast<const uint32_t *>(A.ref_rec + vi * CS)[w2];
                        s_rv[E] = A.ref_valid[vi];
                    }
                }
            };
            auto mproc = [&](auto EC) {
                constexpr int E = decltype(EC)::value;
                if (namb > ACAP - 64) {                                                // room for a whole round of ambiguous vectors
                    st_exact += (unsigned)xs_refine<M, SLQ>(A, j, gb, arec, namb, have_ref);
                    st_amb += (unsigned)namb;
                    namb = 0;
                }
                uint32_t kA = (uint32_t)s_pk[E][0], kB = (uint32_t)(s_pk[E][0] >> 32);
#pragma unroll
                for (int s2 = 1; s2 < G; ++s2) top2_merge(kA, kB, (uint32_t)s_pk[E][s2], (uint32_t)(s_pk[E][s2] >> 32));
                const int64_t vi = gb + s_li[E];
                const bool vf32 = s_on[E] && ((s_qf[E] >> j) & 1);                    // a unary of this node fell outside the sampled level range
                const bool vamb = s_on[E] && !vf32 && ((int)(kB >> 16) - (int)(kA >> 16) <= window);
                const bool von = s_on[E] && !vamb && !vf32;
                uint32_t rwl[RW], rrl[RW];
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) { rwl[w2] = s_rw[E][w2]; rrl[w2] = s_rr[E][w2]; }
                if (von) {                                                             // second - best > window: the best key IS the exact argmin
                    const uint8_t c8 = (uint8_t)(kA & 0xffu);
                    const uint8_t old = (uint8_t)(rwl[j >> 2] >> (8 * (j & 3)));
                    if (c8 != old) store_code<CS>(A.rec, vi, j, c8, rwl);
                    if (A.valid) {
                        unsigned short vm = (c8 != old) ? (unsigned short)(1u << j) : (unsigned short)(s_vo[E] | (1u << j));
                        if (have_ref) {
                            bool same = true;
#pragma unroll
                            for (int w2 = 0; w2 < RW; ++w2) {
                                uint32_t mine = rwl[w2];
                                if (w2 == (j >> 2)) mine = (mine & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)c8 << (8 * (j & 3)));
                                same = same && (mine == rrl[w2]);
                            }
                            if (same) vm = (unsigned short)(vm | s_rv[E]);
                        }
                        A.valid[vi] = vm;
                    }
                }
                {   // ambiguous: every candidate within the window of the best is evaluated exactly (xs_refine); records appended in lane order
                    const unsigned long long am = __ballot(vamb);
                    if (am != 0ull) {
                        const int slot = namb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                        if (vamb) {
                            uint32_t *ar = arec + slot * AREC;
                            ar[0] = s_li[E] | ((kA & 0xffu) << 16) | ((kB & 0xffu) << 24);
                            ar[1] = (kA >> 16) + (uint32_t)window;
#pragma unroll
                            for (int w2 = 0; w2 < RW; ++w2) { ar[2 + w2] = rwl[w2]; ar[3 + RW + w2] = rrl[w2]; }
                            ar[2 + RW] = s_vo[E] | (s_rv[E] << 16);
                        }
                        namb += __builtin_popcountll(am);
                    }
                }
                unsigned long long fm = __ballot(vf32);                                // outside the level range: the whole wave, full f32, one vector at a time
                while (fm != 0ull) {
                    const int L = __builtin_ctzll(fm);
                    fm &= fm - 1ull;
                    uint32_t urw[RW], urr[RW];
#pragma unroll
                    for (int w2 = 0; w2 < RW; ++w2) {
                        urw[w2] = (uint32_t)__builtin_amdgcn_readlane((int)rwl[w2], L);
                        urr[w2] = (uint32_t)__builtin_amdgcn_readlane((int)rrl[w2], L);
                    }
                    const uint32_t uli = (uint32_t)__builtin_amdgcn_readlane((int)s_li[E], L);
                    const uint32_t uvo = (uint32_t)__builtin_amdgcn_readlane((int)s_vo[E], L), urv = (uint32_t)__builtin_amdgcn_readlane((int)s_rv[E], L);
                    xs_f32_update<M, CS>(A, j, gb + uli, urw, uvo, urr, urv, have_ref, lane);
                    st_f32 += 1u;
                }
            };
            using E0 = std::integral_constant<int, 0>;
            using E1 = std::integral_constant<int, 1>;
            mload(E0{}, slo);
            for (int p0 = slo; p0 < shi; p0 += 128) {
                mload(E1{}, p0 + 64);
                mproc(E0{});
                mload(E0{}, p0 + 128);
                mproc(E1{});
            }
            if (namb > 0) {
                st_exact += (unsigned)xs_refine<M, SLQ>(A, j, gb, arec, namb, have_ref);
                st_amb += (unsigned)namb;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's records and validity words are in L2
            XS_STAMP(t, 11); XS_NOTE(t, 13, namb);
            if (lane == 0) {
                // merges finish out of order: publish the contiguous prefix.  Store my flag, THEN read the others' (the LDS executes a wave's
                // operations in order and everybody's in one sequence: of two mergers finishing together at least one sees both flags)
                lds_store(ctl + XC_MDONE0 + (t % XS_R), (unsigned)(t + 1));
                unsigned c = lds_load(ctl + XC_CDONE);
                const unsigned c0 = c;
                while (c < (unsigned)ntasks && lds_load(ctl + XC_MDONE0 + (c % XS_R)) == c + 1u) ++c;
                if (c > c0) {
                    __hip_atomic_fetch_max(ctl + XC_CDONE, c, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                    atomicMax(progc + slice, c);
                }
            }
            XS_STAMP(t, 12);
        }
        if (A.active_total) {
            // st_exact is per lane (16-lane groups count their own survivors); everything else is wave-uniform
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) st_exact += __shfl_xor(st_exact, o, 64);
            for (int e = lane; e < LSQ_WALK_TRACE; e += 64)
                if (trace[e]) atomicAdd(A.active_total + 4 + e, (unsigned long long)trace[e]);
            if (lane == 0) {
                if (st_nodes) atomicAdd(A.active_total + 0, (unsigned long long)st_nodes);
                if (st_tasks) atomicAdd(A.active_total + 3, (unsigned long long)st_tasks);
                if (st_amb) atomicAdd(A.active_total + 4 + LSQ_WALK_TRACE, (unsigned long long)st_amb);
                if (st_exact) atomicAdd(A.active_total + 4 + LSQ_WALK_TRACE + 1, (unsigned long long)st_exact);
                if (st_f32) atomicAdd(A.active_total + 4 + LSQ_WALK_TRACE + 2, (unsigned long long)st_f32);
            }
        }
    }
}

template <int M, int SLQ, int NMERGE = 2>
int launch_xs_t(hipStream_t s, const XsArgs &A0, const WalkNodes &nodes, int64_t n, DevBuf *part, DevBuf *syncb) {
    using C = XsCfg<M, SLQ, NMERGE>;
    XsArgs A = A0;
    int64_t per_group = (n + C::NG - 1) / C::NG;
    per_group = (per_group + 63) / 64 * 64;
    if (per_group > 65536) { lsq_set_error("icm_xs: %lld vectors per group exceed the 16-bit list entries", (long long)per_group); return LSQ_EINVAL; }
    int Q = (int)((per_group + C::LCAP - 1) / C::LCAP);
    if (Q < 2 && per_group >= 512) Q = 2;
    int clen = (int)((per_group + Q - 1) / Q);
    clen = (clen + 3) / 4 * 4;
    A.per_group = (int)per_group; A.clen = clen; A.Q = Q;
    LSQ_TRY(part->ensure(sizeof(unsigned long long) * (size_t)C::NG * XS_R * C::G * (size_t)clen));
    LSQ_TRY(syncb->ensure(sizeof(XsSync)));
    A.part = part->as<unsigned long long>();
    A.sync = syncb->as<XsSync>();
    LSQ_HIP(hipMemsetAsync(A.sync, 0, sizeof(XsSync), s));
    static LdsOptIn optin;
    LSQ_TRY(optin_lds(optin, &icm_xs_kernel<M, SLQ, NMERGE>, C::LDS_BYTES));
    hipLaunchKernelGGL((icm_xs_kernel<M, SLQ, NMERGE>), dim3(256), dim3(C::NT), C::LDS_BYTES, s, A, nodes);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

}  // namespace

#ifdef LSQ_TUNING
// tools only: device buffer of 1024 x 16 u64 that block (group 0, slice 0) of every icm_xs_kernel launch fills with clock stamps (the last launch stays)
extern "C" __attribute__((visibility("default"))) int lsq_tuning_set_xs_debug(void *buf) {
    LSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_xs_dbg), &buf, sizeof(buf)));
    return LSQ_OK;
}
#endif

// Does schedule 7 apply to (n, m)?  (the caller also requires a 256-CU device)
bool lsq_icm_xs_applies(int64_t n, int m) {
    if (m < 2 || m > LSQ_MAX_M) return false;
    const int ng = m <= 8 ? 32 : 16;
    return (n + ng - 1) / ng <= 65536 - 64;
}

// One launch = the node updates order[0 .. nnodes) (<= LSQ_WALK_MAX_NODES) of every vector.  *gate_word receives the device address of the
// launch's gate: 1 after the launch = it ran; 2 = the start barrier said no and NOTHING was touched (the caller's icm_walkq_kernel launch,
// predicated on that word, does the work).  gate_word[1] (the next word) != 0: a wait gave up mid-way -- results invalid, the caller reports
// an error at the end of the call.
int lsq_launch_icm_xs(hipStream_t s, const float *U, const uint16_t *Uq, const uint16_t *Tq, const float *T, uint8_t *rec, unsigned short *valid,
                      int64_t n, int m, const int32_t *order, int nnodes, int pos0, int use_skip, unsigned long long *active_total,
                      const uint8_t *ref_rec, const unsigned short *ref_valid, const lsq_q16_params *P, const unsigned short *qflag,
                      DevBuf *part, DevBuf *syncb, unsigned *err, const unsigned **gate_word) {
    if (n <= 0 || nnodes <= 0) return LSQ_OK;
    if (nnodes > LSQ_WALK_MAX_NODES || !lsq_icm_xs_applies(n, m)) { lsq_set_error("lsq_launch_icm_xs: shape not supported"); return LSQ_EINVAL; }
    WalkNodes nodes;
    nodes.count = nnodes;
    nodes.pos0 = pos0;
    for (int t = 0; t < nnodes; ++t) {
        const int j = order[t];
        if (j < 0 || j >= m) { lsq_set_error("node %d out of range 0..%d", j, m - 1); return LSQ_EINVAL; }
        nodes.j[t] = (uint8_t)j;
    }
    XsArgs A;
    A.U = U; A.Uq = Uq; A.Tq = Tq; A.T = T; A.rec = rec; A.valid = valid;
    const int skip = (use_skip && valid) ? 1 : 0;
    A.ref_rec = skip ? ref_rec : nullptr; A.ref_valid = skip ? ref_valid : nullptr;
    A.P = P; A.qflag = qflag; A.part = nullptr; A.sync = nullptr; A.active_total = active_total; A.err = err;
    A.n = n; A.per_group = 0; A.clen = 0; A.Q = 0; A.use_skip = skip; A.SLF = lsq_walk_slice_width(m);
#ifdef LSQ_TUNING
    if (m == 8 && LSQ_KNOB("LSQ_XS_NM", 2) != 2) {      // tools only: other splits of the 16 waves
        const int nm = LSQ_KNOB("LSQ_XS_NM", 2);
        if (nm == 3) LSQ_TRY((launch_xs_t<8, 32, 3>(s, A, nodes, n, part, syncb)));
        else if (nm == 4) LSQ_TRY((launch_xs_t<8, 32, 4>(s, A, nodes, n, part, syncb)));
        else LSQ_TRY((launch_xs_t<8, 32, 6>(s, A, nodes, n, part, syncb)));
        if (gate_word) *gate_word = &syncb->as<XsSync>()->gate;
        return LSQ_OK;
    }
#endif
    switch (m) {
#define LSQ_XS_CASE(MM, SLL) case MM: LSQ_TRY((launch_xs_t<MM, SLL>(s, A, nodes, n, part, syncb))); break;
        LSQ_XS_CASE(2, 32) LSQ_XS_CASE(3, 32) LSQ_XS_CASE(4, 32) LSQ_XS_CASE(5, 32) LSQ_XS_CASE(6, 32) LSQ_XS_CASE(7, 32) LSQ_XS_CASE(8, 32)
        LSQ_XS_CASE(9, 16) LSQ_XS_CASE(10, 16) LSQ_XS_CASE(11, 16) LSQ_XS_CASE(12, 16) LSQ_XS_CASE(13, 16) LSQ_XS_CASE(14, 16) LSQ_XS_CASE(15, 16) LSQ_XS_CASE(16, 16)
#undef LSQ_XS_CASE
    }
    if (gate_word) *gate_word = &syncb->as<XsSync>()->gate;
    return LSQ_OK;
}
