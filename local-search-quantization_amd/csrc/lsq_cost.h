// lsq_cost.h -- the perturbation and the quarter-wave cost + accept routine, shared by the cost kernels (lsq_icm.hip) and by the filtered walk
// (lsq_icmq.hip: a block that has finished the node updates of its vectors runs their cost + accept + next perturbation itself -- the per-ILS-iteration
// tail of VERDICT r4, next #1b).  Device code only; every translation unit gets its own copy (anonymous namespace).
#pragma once

#include "lsq_wave.h"

namespace {

// ---- perturbation (cudautils.cu:27-80 / encode_icm.jl:55-70) ------------------------------------
// npert distinct positions (selection sampling, ascending) of the record w get uniform codes; Philox keyed by (seed, global index, ILS iteration)
__device__ inline bool perturb_record(uint64_t (&w)[2], int m, int npert, uint64_t seed, uint32_t it, uint64_t gi) {
    // Same stream as orc_perturb (word p decides position p, word 16 + p is its value), drawn BLOCKWISE: the selection words of positions 4 g .. 4 g + 3
    // are Philox block g, their value words block 4 + g -- both computed unconditionally, once per group.  (Round 4 drew every value word with its own
    // Philox call inside the data-dependent branch: up to 2 + m blocks per wave, ~80 quarter-rate multiplies each -- a fifth of the cost pass.)
    int need = npert < m ? npert : m;
    bool changed = false;
    for (int g = 0; 4 * g < m && need > 0; ++g) {
        const lsq_u32x4 sel = lsq_rng_block(seed, gi, it, LSQ_DOM_PERTURB, (uint32_t)g);
        const lsq_u32x4 vals = lsq_rng_block(seed, gi, it, LSQ_DOM_PERTURB, (uint32_t)(4 + g));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pp = 4 * g + e;
            if (pp < m && need > 0 && lsq_mulhi32(sel.v[e], (uint32_t)(m - pp)) < (uint32_t)need) {
                const uint64_t val = lsq_mulhi32(vals.v[e], LSQ_H);
                const int sh = 8 * (pp & 7);
                changed |= (((w[pp >> 3] >> sh) & 0xffull) != val);
                w[pp >> 3] = (w[pp >> 3] & ~(0xffull << sh)) | (val << sh);
                --need;
            }
        }
    }
    return changed;
}

// The cost kernels can perturb for the NEXT ILS iteration on their way out (lsq_perturb_next::on): the lane that looked at vector i knows its
// final record (accepted candidate or current one) and validity word, so the separate pass over the records and its launch disappear.
template <int CS>
__device__ inline void perturb_next_store(const lsq_perturb_next &pn, int64_t i, const uint32_t (&fin)[CS / 4], unsigned short vfin) {
    uint64_t w[2];
    w[0] = (uint64_t)fin[0] | ((uint64_t)fin[1] << 32);
    w[1] = (CS == 16) ? ((uint64_t)fin[CS / 4 - 2] | ((uint64_t)fin[CS / 4 - 1] << 32)) : 0ull;
#ifdef LSQ_TUNING
    if (pn.abl & 16) return;
    const bool changed = (pn.abl & 1) ? false : perturb_record(w, pn.m, pn.npert, pn.seed, pn.it, pn.goff + (uint64_t)i);
#else
    const bool changed = perturb_record(w, pn.m, pn.npert, pn.seed, pn.it, pn.goff + (uint64_t)i);
#endif
    uint64_t *q = reinterpret_cast<uint64_t *>(pn.dst + i * CS);
    q[0] = w[0];
    if (CS == 16) q[1] = w[1];
    if (pn.vdst) pn.vdst[i] = changed ? (unsigned short)0 : vfin;      // a changed code invalidates every node
}

// d % 4 == 0: a QUARTER wave (one 16-lane DPP row) per vector, 16-byte loads.  Lane l' (0..15) of a row owns dimensions t = 64 q + 4 l' .. + 3
// for q = 0, 1, ..: the four residues 4 l' .. 4 l' + 3 (mod 64) of the canonical 64 strided partial sums, each accumulated in ascending t, so the
// reduction order is exactly oracle cost_one()'s: levels 1 and 2 in-lane ((p0 + p1) + (p2 + p3)), levels 4 .. 32 across the 16 lanes of the row
// (DPP).
//
// Round 5 rewrite (same arithmetic, same results).  Measured on the round-4 kernel: the pass took 212-340 us where its own memory traffic, replayed by
// tools/ubench_cost.hip, needs 100-280 us -- the rest was structure: (i) the batch preamble (candidate record, current record, cost, two validity
// words) compiled into FIVE dependent round trips (a load, a wait, a branch, the next load); (ii) every accepted vector stored its record / cost and
// re-LOADED its validity word inside the round (a wait for a scattered load in front of the next round's gathers); (iii) one 64-float step of d per
// wait.  Now: MODE / HASV are template parameters, so the preamble is one group of unconditional loads; a round only computes -- its four costs
// travel to the lanes that own the vectors (v_readlane / select) and the batch ends with lane-parallel, coalesced stores of whatever was
// accepted; NQ steps of d (two for d >= 128: 18 sixteen-byte loads per lane) are in flight per wait.
template <int M, int MODE, int HASV, int NQ, bool EXTCNT = false>
__device__ inline void cost4_body(const float *__restrict__ X, const float *__restrict__ K,
                                  const uint8_t *rec, uint8_t *cur, float *__restrict__ prev,
                                  unsigned long long *__restrict__ counters, int64_t lo, int64_t n, int64_t w, int64_t nwaves, int d,
                                  const unsigned short *vnew, unsigned short *vcur, const lsq_perturb_next &pn, unsigned *cnt_ext = nullptr) {      // EXTCNT: the caller's two LDS words instead of a static pair
    // vectors [lo, n): wave w of nwaves takes the 64-vector batches lo + 64 (w + q nwaves).  The cost kernels pass lo = 0 and their grid-wide wave
    // index; the filtered walk passes a block's own range and the wave's index inside the block.
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int RW = CS / 4;
    const int lane = threadIdx.x & 63;
    const int qtr = lane >> 4, lp = lane & 15;
    unsigned n_eq = 0, n_lt = 0;
    // 64 consecutive vectors per wave batch.  Accept mode: a vector whose candidate record equals its current record has, bit for bit, the cost it
    // already has (same codes, same arithmetic) -- counted as "equal" (encode_icm_cuda.jl:199-204) without touching X or the codebooks (a NaN cost is
    // never "equal" in the reference's comparison: those are evaluated).  The others are taken four at a time, one per row.
    for (int64_t base = lo + w * 64; base < n; base += nwaves * 64) {
        const int64_t il = base + lane;
        const bool livel = il < n;
        const int64_t ic = livel ? il : n - 1;
        uint32_t rn[RW], cw[RW];
        float pl = 0.0f;
        unsigned short vc = 0, vn = 0;
        {   // one group of independent loads (no control flow between them)
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rec + ic * CS), *cp = reinterpret_cast<const uint32_t *>(cur + ic * CS);
#pragma unroll
            for (int q = 0; q < RW; ++q) rn[q] = rp[q];
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < RW; ++q) cw[q] = cp[q];
                pl = prev[ic];
            }
            if (HASV) vc = vcur[ic];
            if (HASV && MODE == 1) vn = vnew[ic];
        }
        bool same = (MODE == 1);
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            if (MODE == 0) cw[q] = rn[q];
            same = same && (rn[q] == cw[q]);
        }
        const bool skip = livel && same && (pl == pl);
        const unsigned nskip = (unsigned)__popcll(__ballot(skip));
        if (lane == 0) n_eq += nskip;
        unsigned short vfin = vc;                                                   // the vector's validity word after this kernel (for the fused perturbation)
        const bool merge_v = HASV && MODE == 1 && same && livel;                    // same tuple: what the sweeps learnt about it is kept
        if (merge_v) vfin = (unsigned short)(vfin | vn);
        uint64_t accepted = 0;                                                      // bit l: the candidate of vector base + l replaced the current record
        float newp = 0.0f;                                                          // the evaluated cost of this lane's vector
        uint64_t todo = __ballot(livel && !skip);
#ifdef LSQ_TUNING
        if (pn.abl & 8) todo = 0;
#endif
        // Rounds of four vectors.  The x row comes from HBM, the codeword rows from L2: waiting for both in the same round leaves the L1 idle for the
        // length of an HBM round trip, so the x loads run ONE STEP AHEAD -- those of the next 64 NQ dimensions (or of the next round's vectors) are
        // issued behind the current step's codeword loads and are still in flight while it is summed.
        int nsidx[4];
        bool nhv[4];
        auto pick = [&]() {
#pragma unroll
            for (int v = 0; v < 4; ++v) {                                           // fewer than four left: the spare rows repeat the first, unwritten
                nhv[v] = todo != 0;
                nsidx[v] = nhv[v] ? __builtin_ctzll(todo) : (v ? nsidx[0] : 0);
                if (nhv[v]) todo &= todo - 1;
            }
        };
        auto xrow = [&]() -> const float * {
            const int mys = qtr == 0 ? nsidx[0] : qtr == 1 ? nsidx[1] : qtr == 2 ? nsidx[2] : nsidx[3];
            return X + (base + mys) * (int64_t)d;                                   // an empty pick points at the batch's first vector: a harmless read
        };
        f32x4 xn[NQ];
        auto xload = [&](const float *x, int c0) {
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                const int t = c0 + 64 * g + 4 * lp;
                xn[g] = *reinterpret_cast<const f32x4 *>(x + (t < d ? t : 0));
            }
        };
        pick();
        const float *xnext = xrow();
        if (nhv[0]) xload(xnext, 0);
        while (nhv[0]) {
            int sidx[4];
            bool hv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { sidx[v] = nsidx[v]; hv[v] = nhv[v]; }
            const float *x = xnext;
            pick();                                                                 // the NEXT round's vectors (none: nhv[0] = false)
            xnext = xrow();
            const bool live = qtr == 0 ? hv[0] : qtr == 1 ? hv[1] : qtr == 2 ? hv[2] : hv[3];
            uint32_t r[RW];
            float pc;
            {
                uint32_t rv4[4][RW];
                float pc4[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int q = 0; q < RW; ++q) rv4[v][q] = (uint32_t)__builtin_amdgcn_readlane((int)rn[q], sidx[v]);
                    pc4[v] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl), sidx[v]));
                }
#pragma unroll
                for (int q = 0; q < RW; ++q) r[q] = qtr == 0 ? rv4[0][q] : qtr == 1 ? rv4[1][q] : qtr == 2 ? rv4[2][q] : rv4[3][q];
                pc = qtr == 0 ? pc4[0] : qtr == 1 ? pc4[1] : qtr == 2 ? pc4[2] : pc4[3];
            }
            uint32_t kb[M];                                                          // element offsets into K (m h d < 2^32: checked by the launcher): one register per row
#pragma unroll
            for (int k = 0; k < M; ++k) kb[k] = ((uint32_t)(k * LSQ_H) + ((r[k >> 2] >> (8 * (k & 3))) & 0xffu)) * (uint32_t)d;
            f32x4 p = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < d; c0 += 64 * NQ) {
                f32x4 xv[NQ], kv[NQ][M];
                bool ok[NQ];
#pragma unroll
                for (int g = 0; g < NQ; ++g) {                                       // every codeword load of the NQ steps is issued before the first add
                    const int t = c0 + 64 * g + 4 * lp;
                    ok[g] = t < d;
                    const uint32_t u = ok[g] ? (uint32_t)t : 0u;
                    xv[g] = xn[g];
#pragma unroll
                    for (int k = 0; k < M; ++k) kv[g][k] = *reinterpret_cast<const f32x4 *>(K + (size_t)(kb[k] + u));
                }
                if (c0 + 64 * NQ < d) xload(x, c0 + 64 * NQ);                        // ... and behind them the x of the next step
                else xload(xnext, 0);
#pragma unroll
                for (int g = 0; g < NQ; ++g) {                                       // step by step: residues 4 l' .. 4 l' + 3 accumulate in ascending t
                    f32x4 cb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < M; ++k) cb = cb + kv[g][k];                  // k ascending from 0 (utils.jl:238-244)
                    const f32x4 rr = cb - xv[g];
                    const f32x4 sq = rr * rr;                                        // never fused (-ffp-contract=off)
                    p.x = p.x + (ok[g] ? sq.x : 0.0f);
                    p.y = p.y + (ok[g] ? sq.y : 0.0f);
                    p.z = p.z + (ok[g] ? sq.z : 0.0f);
                    p.w = p.w + (ok[g] ? sq.w : 0.0f);
                }
            }
            float v = (p.x + p.y) + (p.z + p.w);                                    // tree levels 1 and 2
            v = v + dpp_self<DPP_XOR1, 0xf>(v);                                     // levels 4, 8, 16, 32: within the row
            v = v + dpp_self<DPP_XOR2, 0xf>(v);
            v = v + dpp_self<DPP_HALF_MIRROR, 0xf>(v);
            v = v + dpp_self<DPP_MIRROR, 0xf>(v);
            const float cost = v;                                                   // every lane of the row holds its vector's cost
            // the four costs go to the lanes that own the vectors: nothing is stored inside the round
#pragma unroll
            for (int v2 = 0; v2 < 4; ++v2) {
                const float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cost), 16 * v2));
                if (hv[v2] && lane == sidx[v2]) newp = cv;
            }
            if (MODE == 1) {
                const bool eq = live && (cost == pc), lt = live && (cost < pc);     // strict improvement only (encode_icm.jl:183-186)
                const uint64_t bl = __ballot(lt && lp == 0);                        // bits 0, 16, 32, 48: the four rows' vectors
                const unsigned ne = (unsigned)__popcll(__ballot(eq && lp == 0)), nl = (unsigned)__popcll(bl);
#pragma unroll
                for (int v2 = 0; v2 < 4; ++v2)
                    if ((bl >> (16 * v2)) & 1ull) accepted |= 1ull << sidx[v2];
                if (lane == 0) { n_eq += ne; n_lt += nl; }
            }
        }
        // lane-parallel epilogue of the batch: coalesced stores of what changed
        const bool acc = (accepted >> lane) & 1ull;
        if (MODE == 0) {
            if (livel) prev[il] = newp;
        } else if (acc) {
            prev[il] = newp;
            uint32_t *qd = reinterpret_cast<uint32_t *>(cur + il * CS);
#pragma unroll
            for (int q = 0; q < RW; ++q) qd[q] = rn[q];
        }
        if (HASV && MODE == 1 && (acc || merge_v)) vcur[il] = acc ? vn : vfin;
        if (pn.on && livel) {
            uint32_t fin[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) fin[q] = acc ? rn[q] : cw[q];
            perturb_next_store<CS>(pn, il, fin, acc ? vn : vfin);
        }
    }
    if (MODE == 1) {                                                                 // one pair of device atomics per block
        unsigned *cnt_s;
        if constexpr (EXTCNT) cnt_s = cnt_ext;
        else { __shared__ unsigned cnt_own[2]; cnt_s = cnt_own; }
        if (threadIdx.x == 0) { cnt_s[0] = 0u; cnt_s[1] = 0u; }
        __syncthreads();
        if (lane == 0 && n_eq) atomicAdd(&cnt_s[0], n_eq);
        if (lane == 0 && n_lt) atomicAdd(&cnt_s[1], n_lt);
        __syncthreads();
        if (threadIdx.x == 0 && cnt_s[0]) atomicAdd(&counters[0], (unsigned long long)cnt_s[0]);
        if (threadIdx.x == 0 && cnt_s[1]) atomicAdd(&counters[1], (unsigned long long)cnt_s[1]);
    }
}


}  // namespace
