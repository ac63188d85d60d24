// lsq_adc.hip -- ADC linear scan ON THE DEVICE (gfx950): the search step right after the encode path ("HIP scan later").
//
// Replaces linscan_aqd_query_extra_byte of the reference (src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:14-104, bound at
// src/linscan/Linscan.jl:63-69) with the same results bit for bit -- distances, 1-based ids and their order including ties -- as the
// reference build in oracle/_ref and as this library's host scan (lsq_linscan.hip).  Same arithmetic, different machinery:
//
//   table[e]  = ((0 - (2 q_0) c_e0) - (2 q_1) c_e1) - ...        f32, k ascending, multiply and subtract rounded separately (:45-47)
//   dist(i)   = (((0 + table[0 h + b_i0]) + table[1 h + b_i1]) + ...) + dbnorms[i]                                      (:66-71)
//   result    = the nn smallest (dist, id) pairs in lexicographic order                                                (:75,84-87)
//
// The reference materialises 10^7 (dist, id) pairs per query and partial_sorts them.  Here:
//   LUT     adc_lut_kernel: the tables of a tile of QT queries, stored transposed ([entry][query]) so that one 16-byte LDS read serves one
//           code for four queries.
//   SCAN    adc_scan_kernel: a block keeps the tables of its QT queries in LDS (m KiB per query: 128 KiB at m = 8, QT = 16) and walks a
//           range of codes; a lane owns (one code, four queries): m ds_read_b128 + 4m adds per code.  This is the whole cost of the
//           search -- nq n m table lookups, bound by the LDS gather rate -- and the only part that touches the database (m + 4 bytes per code
//           and query TILE; the codes of 10^6 vectors are 8 MB and stay in L2).
//   SELECT  no distance is ever written to memory unless it can be among the nn smallest: a strided SAMPLE of the database (16 384 codes)
//           gives every query a threshold tau_q (an order statistic of its sample distances chosen ~6 sigma above the nn / n quantile); the
//           scan appends (dist, id) pairs with dist <= tau_q to a per-query candidate list (a few nn entries), which is sorted as 64-bit keys
//           (order-preserving distance bits << 32 | id: the lexicographic pair order) and its first nn entries are the answer.  The
//           threshold is a heuristic, the answer is not: a query whose list holds fewer than nn entries or overflows (sorted or clustered
//           databases, massive ties) is redone by the exhaustive road -- every distance written, sorted -- and small databases take that
//           road directly.  rocPRIM's segmented radix sort (hipcub) does the sorting.
// NaN distances (the reference's partial_sort has no defined order for them) sort after everything else here.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "lsq_internal.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ADC_SAMPLE = 16384;            // sample size of the threshold estimate
constexpr int ADC_SMALL_N = 65536;           // databases up to this size: exhaustive road (every distance written and sorted)
constexpr int ADC_SCAN_THREADS = 1024;
constexpr int ADC_KC = 1024;                 // LUT build: dimensions staged per pass

__device__ inline uint32_t adc_key(float v) {                    // order-preserving: a < b  <=>  key(a) < key(b);  NaN last
    if (v != v) return 0xffffffffu;
    const uint32_t b = __float_as_uint(v);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ inline float adc_unkey(uint32_t k) {
    if (k == 0xffffffffu) return __uint_as_float(0x7fc00000u);
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// LUT[tile][e][s] = table entry e = j h + a of query qid(tile * QT + s);  qsel (optional) lists the query ids of this batch
template <int QT>
__global__ __launch_bounds__(256) void adc_lut_kernel(const float *__restrict__ Q, const float *__restrict__ K, const int *__restrict__ qsel, int q0,
                                                      int nqb, int d, int entries, float *__restrict__ LUT) {
    extern __shared__ __attribute__((aligned(16))) float q2[];      // [kc][QT]: 2 q (exact): one k = QT consecutive floats, read as broadcast 16-byte words
    const int tile = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    float acc[QT];
#pragma unroll
    for (int s = 0; s < QT; ++s) acc[s] = 0.0f;
    const float *c = K + (int64_t)(e < entries ? e : 0) * d;
    const bool vec = (d & 3) == 0 && (((uintptr_t)K & 15) == 0);
    auto step = [&](int k, float ck) {
#pragma unroll
        for (int s = 0; s < QT; s += 4) {
            const f32x4 qv = *reinterpret_cast<const f32x4 *>(q2 + k * QT + s);
            acc[s] = acc[s] - qv.x * ck;                            // product rounded, then the subtraction (no FMA: contraction is off)
            acc[s + 1] = acc[s + 1] - qv.y * ck;
            acc[s + 2] = acc[s + 2] - qv.z * ck;
            acc[s + 3] = acc[s + 3] - qv.w * ck;
        }
    };
    for (int k0 = 0; k0 < d; k0 += ADC_KC) {
        const int kc = d - k0 < ADC_KC ? d - k0 : ADC_KC;
        __syncthreads();
        for (int t = threadIdx.x; t < QT * kc; t += 256) {
            const int s = t / kc, k = t % kc, slot = tile * QT + s;      // consecutive threads read consecutive dimensions of one query
            float v = 0.0f;
            if (slot < nqb) v = Q[(int64_t)(qsel ? qsel[slot] : q0 + slot) * d + k0 + k];
            q2[k * QT + s] = 2 * v;
        }
        __syncthreads();
        if (vec) {
            for (int k = 0; k < kc; k += 4) {                       // d % 4 == 0 and ADC_KC % 4 == 0: whole quads
                const f32x4 cv = *reinterpret_cast<const f32x4 *>(c + k0 + k);
                step(k, cv.x); step(k + 1, cv.y); step(k + 2, cv.z); step(k + 3, cv.w);
            }
        } else {
            for (int k = 0; k < kc; ++k) step(k, c[k0 + k]);
        }
    }
    if (e < entries) {
        float *o = LUT + ((int64_t)tile * entries + e) * QT;
#pragma unroll
        for (int s = 0; s < QT; s += 4) *reinterpret_cast<f32x4 *>(o + s) = (f32x4){acc[s], acc[s + 1], acc[s + 2], acc[s + 3]};
    }
}

constexpr int ADC_STAGE = 64;                // staged (dist, id) pairs per wave
struct AdcStage {
    uint64_t rec[ADC_STAGE];
    unsigned slot[ADC_STAGE];                // query of the tile (0 .. QT-1)
    unsigned hist[16], base[16];
    unsigned n, pad[3];
};

// MODE 0: append (key << idbits | id) of every distance <= tau to the query's candidate list;  MODE 1: write every (key << idbits | id) of the strided
// subset i = s * stride, s < ns, to out[slot * ns + s];  MODE 2: the same subset, keys only (u32).  MW > 0: m = 4 MW, codes read as dwords;
// MW = 0: any m, byte reads.
// A lane owns one code and four queries per step.  The codes of the NEXT batch of U steps are requested before the current batch is walked: a
// code's bytes come from L2 (~1 us away), its 16-byte table reads from LDS (~0.1 us), and without the prefetch the walk waits for L2 once per
// code (measured: 7 of 28 TB/s of LDS gathers).
template <int QT, int MODE, int MW>
__global__ __launch_bounds__(ADC_SCAN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void adc_scan_kernel(const float *__restrict__ LUT, const uint8_t *__restrict__ codes,
                                                                    const float *__restrict__ dbnorms, int n, int m, int nqb, int stride, int ns,
                                                                    int per_block, const uint32_t *__restrict__ tau, unsigned *__restrict__ count,
                                                                    int cap, uint64_t *__restrict__ out, int idbits) {
    extern __shared__ __attribute__((aligned(16))) float lut[];      // [m * 256][QT], then the emission staging of MODE 0 (AdcStage)
    constexpr int NQ = QT / 4, CPW = 64 / NQ;                        // query quads, codes per wave step
    constexpr int U = MW == 0 ? 1 : (MW <= 2 ? 4 : 2);               // steps per batch
    constexpr int G = (MW == 1 || MW == 2) ? 2 : 1;                  // steps whose table reads are in flight together (<= 64 registers of them)
    constexpr int CW = MW > 0 ? MW : 1;
    const int tile = blockIdx.x;
    const int entries = m * LSQ_H;
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(LUT + (int64_t)tile * entries * QT);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lut);
        for (int t = threadIdx.x; t < entries * QT / 4; t += ADC_SCAN_THREADS) dst[t] = src[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qq = lane % NQ, cs = lane / NQ;
    const int slot0 = tile * QT + 4 * qq;                            // the lane's four queries
    float tf[4] = {0.0f, 0.0f, 0.0f, 0.0f};                          // thresholds as floats: emit unless dist > tau (a NaN on either side emits)
    if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) tf[c] = slot0 + c < nqb ? adc_unkey(tau[slot0 + c]) : -__builtin_inff();
    }
    const int total = MODE == 0 ? n : ns;
    const int first = blockIdx.y * per_block;
    const int last = first + per_block < total ? first + per_block : total;
    const f32x4 *lut4 = reinterpret_cast<const f32x4 *>(lut);
    constexpr int NW = ADC_SCAN_THREADS / 64;
    // MODE 0: a pair that passes the threshold is parked in the wave's LDS staging list; a full list is flushed with ONE global atomic per query
    // (16 lanes, one wait) instead of one returning atomic -- a round trip to L2 that stalls the wave -- per pair (measured: 0.45 ms per 1000
    // candidates per query at 10^4 queries).
    AdcStage *stage = reinterpret_cast<AdcStage *>(lut + (size_t)entries * QT) + wave;
    if (MODE == 0 && lane == 0) stage->n = 0;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned cnt = stage->n < (unsigned)ADC_STAGE ? stage->n : (unsigned)ADC_STAGE;
        if (lane < QT) stage->hist[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool have = (unsigned)lane < cnt;
        const uint64_t rec = have ? stage->rec[lane] : 0ull;
        const unsigned q = have ? stage->slot[lane] : 0u;
        const unsigned rank = have ? atomicAdd(&stage->hist[q], 1u) : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < QT) {
            const unsigned c = stage->hist[lane];
            stage->base[lane] = c ? atomicAdd(&count[tile * QT + lane], c) : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (have) {
            const unsigned at = stage->base[q] + rank;
            if (at < (unsigned)cap) out[(int64_t)(tile * QT + (int)q) * cap + at] = rec;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) stage->n = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    struct Batch { uint32_t w[U][CW]; float nrm[U]; };
    auto fetch = [&](Batch &bt, int s0) {                            // codes s0 + u * CPW + cs, u < U (dead ones read the block's first code)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u * CPW + cs;
            const int64_t i = (int64_t)(s < last ? s : first) * (MODE == 0 ? 1 : stride);
            if (MW > 0) {
                const uint32_t *cp = reinterpret_cast<const uint32_t *>(codes + i * (4 * CW));
#pragma unroll
                for (int t = 0; t < CW; ++t) bt.w[u][t] = cp[t];
            }
            bt.nrm[u] = dbnorms[i];
        }
    };
    Batch cur, nxt;
    int s0 = first + wave * (U * CPW);
    if (s0 < last) fetch(cur, s0);
    for (; s0 < last; s0 += NW * U * CPW) {
        const int sn = s0 + NW * U * CPW;
        if (sn < last) fetch(nxt, sn);
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += G) {
            // all table reads of G steps are requested before the first is consumed (the LDS round trip is paid once per group, not once per read)
            f32x4 v[G][MW > 0 ? 4 * CW : 1];
            f32x4 dist[G];
            if (MW > 0) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int j = 0; j < 4 * CW; ++j) {
                        const uint32_t b = (cur.w[u0 + g][j >> 2] >> (8 * (j & 3))) & 0xffu;
                        v[g][j] = lut4[(j * LSQ_H + (int)b) * NQ + qq];
                    }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s = s0 + (u0 + g) * CPW + cs;
                f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                if (MW > 0) {
#pragma unroll
                    for (int j = 0; j < 4 * CW; ++j) acc = acc + v[g][j];
                } else {
                    const int64_t i = (int64_t)(s < last ? s : first) * (MODE == 0 ? 1 : stride);
                    const uint8_t *cp = codes + i * m;
                    for (int j = 0; j < m; ++j) acc = acc + lut4[(j * LSQ_H + (int)cp[j]) * NQ + qq];
                }
                const float nrm = cur.nrm[u0 + g];
                dist[g] = acc + (f32x4){nrm, nrm, nrm, nrm};
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s = s0 + (u0 + g) * CPW + cs;
                const bool live = s < last;
                const int64_t i = (int64_t)(live ? s : first) * (MODE == 0 ? 1 : stride);
                const float dv[4] = {dist[g].x, dist[g].y, dist[g].z, dist[g].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // MODE 0: one test per query; a wave enters the body for ~16 % of its steps per query (list of 2.8 nn entries, nn = 1000, n = 10^6)
                    const bool hit = MODE == 0 ? (live && !(dv[c] > tf[c])) : live;
                    if (!hit) continue;
                    const int slot = slot0 + c;
                    if (slot >= nqb) continue;
                    const uint32_t key = adc_key(dv[c]);
                    const uint64_t rec = ((uint64_t)key << idbits) | (uint64_t)(i + 1);      // ids are 1-BASED (:75)
                    if (MODE == 1) {
                        out[(int64_t)slot * ns + s] = rec;
                    } else if (MODE == 2) {
                        reinterpret_cast<uint32_t *>(out)[(int64_t)slot * ns + s] = key;
                    } else {
                        const unsigned pos = atomicAdd(&stage->n, 1u);
                        if (pos < (unsigned)ADC_STAGE) {
                            stage->rec[pos] = rec;
                            stage->slot[pos] = (unsigned)(4 * qq + c);
                        } else {                                     // more than a list's worth in one group: the direct road
                            const unsigned at = atomicAdd(&count[slot], 1u);
                            if (at < (unsigned)cap) out[(int64_t)slot * cap + at] = rec;
                        }
                    }
                }
            }
        }
        if (MODE == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (__builtin_amdgcn_readfirstlane((int)stage->n) >= ADC_STAGE / 2) flush();      // wave-uniform
        }
        if (sn < last) cur = nxt;
    }
    if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (__builtin_amdgcn_readfirstlane((int)stage->n) > 0) flush();
    }
}

// tau[slot] = the r-th smallest of the query's ADC_SAMPLE sample keys, one block per query, the keys held in registers (64 per thread: one pass over
// memory).  Radix select on (key - lo) with 256 bins per round, lo = the sample minimum at first and the range shrinking 256-fold per round: bins on
// the sample's own scale spread the keys (the top byte of a float key is the same for all of them -- 16 384 LDS atomics on one word per round
// otherwise: 0.64 ms per 10^4 queries, measured).
__global__ __launch_bounds__(256) void adc_rank_select_kernel(const uint32_t *__restrict__ keys, int r, uint32_t *__restrict__ tau) {
    constexpr int PER = ADC_SAMPLE / 256;
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_digit, sel_rank, wmin[4], wmax[4];
    const uint32_t *kq = keys + (int64_t)blockIdx.x * ADC_SAMPLE;
    uint32_t k[PER];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        k[t] = kq[t * 256 + threadIdx.x];
        mn = k[t] < mn ? k[t] : mn;
        mx = k[t] > mx ? k[t] : mx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) { wmin[threadIdx.x >> 6] = mn; wmax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    uint32_t lo = wmin[0], hi = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) { lo = wmin[w] < lo ? wmin[w] : lo; hi = wmax[w] > hi ? wmax[w] : hi; }
    int shift = 0;                                                   // smallest shift with (hi - lo) >> shift < 256
    while (shift < 24 && ((hi - lo) >> shift) >= 256u) ++shift;
    unsigned rank = (unsigned)r;                                     // 1-based rank among the keys >= lo
    for (;;) {
        hist[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const uint32_t off = k[t] - lo;                         // keys below lo wrap to huge offsets: outside every bin
            if (k[t] >= lo && (off >> shift) < 256u) atomicAdd(&hist[off >> shift], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                     // one wave: inclusive scan of the 256 bins, four per lane
            const unsigned h0 = hist[4 * threadIdx.x], h1 = hist[4 * threadIdx.x + 1], h2 = hist[4 * threadIdx.x + 2], h3 = hist[4 * threadIdx.x + 3];
            unsigned sum = h0 + h1 + h2 + h3, incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, 64);
                if ((int)threadIdx.x >= off) incl += o;
            }
            const unsigned before = incl - sum;
            if (before < rank && rank <= incl) {                    // exactly one lane (rank <= number of keys in the binned range)
                unsigned c = before, d = 0;
                if (rank > c + h0) { c += h0; d = 1; if (rank > c + h1) { c += h1; d = 2; if (rank > c + h2) { c += h2; d = 3; } } }
                sel_digit = 4 * threadIdx.x + d;
                sel_rank = rank - c;
            }
        }
        __syncthreads();
        lo += sel_digit << shift;
        rank = sel_rank;
        __syncthreads();
        if (shift == 0) break;
        shift = shift > 8 ? shift - 8 : 0;
    }
    if (threadIdx.x == 0) tau[blockIdx.x] = lo;
}

// segment [begin, end) of every query's list;  count == nullptr: full segments of `cap` entries.  fail[slot] = 1: too few or too many candidates
__global__ void adc_segments_kernel(const unsigned *__restrict__ count, int nqb, int cap, int nn, int *__restrict__ begin, int *__restrict__ end,
                                    int *__restrict__ fail) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nqb) return;
    const unsigned c = count ? count[slot] : (unsigned)cap;
    const bool bad = c > (unsigned)cap || c < (unsigned)nn;
    begin[slot] = slot * cap;
    end[slot] = slot * cap + (bad ? 0 : (int)c);
    if (fail) fail[slot] = bad ? 1 : 0;
}

__global__ void adc_gather_kernel(const uint64_t *__restrict__ sorted, const int *__restrict__ fail, const int *__restrict__ qsel, int q0, int nqb,
                                  int cap, int nn, float *__restrict__ dists, int *__restrict__ idx, int idbits) {
    const int slot = blockIdx.y;
    if (fail && fail[slot]) return;
    const int64_t q = qsel ? qsel[slot] : q0 + slot;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nn; r += gridDim.x * blockDim.x) {
        const uint64_t rec = sorted[(int64_t)slot * cap + r];
        dists[q * nn + r] = adc_unkey((uint32_t)(rec >> idbits));
        idx[q * nn + r] = (int)(uint32_t)(rec & ((1ull << idbits) - 1ull));
    }
}


}  // namespace

struct lsq_adc_state {
    DevBuf lut, keys_a, keys_b, tau, count, seg, fail, qsel, tmp;
    DevBuf h_codes, h_q, h_k, h_norms, h_dists, h_idx;      // staging of the host-buffer entry point
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool attr_set = false;
};

void lsq_adc_free(lsq_adc_state *st) {
    if (!st) return;
    DevBuf *all[] = {&st->lut, &st->keys_a, &st->keys_b, &st->tau, &st->count, &st->seg, &st->fail, &st->qsel, &st->tmp,
                  &st->h_codes, &st->h_q, &st->h_k, &st->h_norms, &st->h_dists, &st->h_idx};
    for (DevBuf *b : all) b->release();
    for (hipEvent_t e : st->ev) if (e) (void)hipEventDestroy(e);
    delete st;
}

namespace {

template <int QT, int MODE>
int launch_scan(hipStream_t s, const float *LUT, const uint8_t *codes, const float *dbnorms, int n, int m, int nqb, int stride, int ns,
                const uint32_t *tau, unsigned *count, int cap, uint64_t *out, int idbits) {
    const int tiles = (nqb + QT - 1) / QT;
    const int total = MODE == 0 ? n : ns;
    if (tiles == 0 || total == 0) return LSQ_OK;
    // enough blocks to fill the chip a few times over, ranges long enough that the table load (m KiB per query) is amortised
    int ranges = (2048 + tiles - 1) / tiles;
    const int max_ranges = (total + 4095) / 4096;
    if (ranges > max_ranges) ranges = max_ranges;
    if (ranges < 1) ranges = 1;
    if (ranges > 65535) ranges = 65535;
    int per_block = (total + ranges - 1) / ranges;
    per_block = (per_block + 255) & ~255;
    ranges = (total + per_block - 1) / per_block;
    const size_t lds = sizeof(float) * (size_t)m * LSQ_H * QT + (MODE == 0 ? sizeof(AdcStage) * (ADC_SCAN_THREADS / 64) : 0);
    const bool words = (m % 4 == 0) && (((uintptr_t)codes & 3) == 0);
    const dim3 grid((unsigned)tiles, (unsigned)ranges), block(ADC_SCAN_THREADS);
#define ADC_LAUNCH(MWV)                                                                                                                     \
    do {                                                                                                                                    \
        LSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&adc_scan_kernel<QT, MODE, MWV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                                                             \
        hipLaunchKernelGGL((adc_scan_kernel<QT, MODE, MWV>), grid, block, lds, s, LUT, codes, dbnorms, n, m, nqb, stride, ns, per_block, tau,  \
                           count, cap, out, idbits);                                                                                                \
    } while (0)
    if (words && m == 4) ADC_LAUNCH(1);
    else if (words && m == 8) ADC_LAUNCH(2);
    else if (words && m == 12) ADC_LAUNCH(3);
    else if (words && m == 16) ADC_LAUNCH(4);
    else ADC_LAUNCH(0);
#undef ADC_LAUNCH
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

template <int QT>
int launch_lut(hipStream_t s, const float *Q, const float *K, const int *qsel, int q0, int nqb, int d, int m, float *LUT) {
    const int tiles = (nqb + QT - 1) / QT, entries = m * LSQ_H;
    if (tiles == 0) return LSQ_OK;
    const int kc = d < ADC_KC ? d : ADC_KC;
    const size_t lds = sizeof(float) * (size_t)QT * kc;
    LSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&adc_lut_kernel<QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((adc_lut_kernel<QT>), dim3((unsigned)((entries + 255) / 256), (unsigned)tiles), dim3(256), lds, s, Q, K, qsel, q0, nqb, d,
                       entries, LUT);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int sort_segments(lsq_adc_state *st, hipStream_t s, const uint64_t *in, uint64_t *out, int64_t items, int segs, const int *begin, const int *end,
                  int end_bit) {
    size_t bytes = 0;
    LSQ_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, bytes, in, out, (int)items, segs, begin, end, 0, end_bit, s));
    LSQ_TRY(st->tmp.ensure(bytes > 0 ? bytes : 16));
    LSQ_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(st->tmp.p, bytes, in, out, (int)items, segs, begin, end, 0, end_bit, s));
    return LSQ_OK;
}

struct Plan { int ns, stride, r, cap; bool exhaustive; };

// Threshold rank and list capacity.  The nn-th neighbour of a query has, among the ns sampled codes, a rank that is Binomial(ns, nn / n) ~ Poisson(ks),
// ks = nn ns / n, on exchangeable data; the list misses it iff that rank reaches r.  r = ks + 5.6 sqrt(ks) + 6 puts this below ~1e-8 per query (small
// ks: the Poisson tail, e.g. ks = 16.4, r = 45: 3e-9; large ks: 5.6 sigma), and the list then holds r n / ns entries on average (2.7 nn at
// nn = 1000, n = 10^6).  The capacity leaves the same room on the other side.  Anything else -- sorted or clustered databases, ties -- is the exhaustive
// road's job: the rank is a bet on speed, never on the answer.
Plan make_plan(int n, int nn, int force_exhaustive) {
    Plan p;
    p.exhaustive = force_exhaustive || n <= ADC_SMALL_N;
    p.ns = ADC_SAMPLE; p.stride = n / ADC_SAMPLE; p.r = 0; p.cap = n;
    if (p.exhaustive) return p;
    const double ks = (double)nn * p.ns / (double)n;
    const double r = ks + 5.6 * sqrt(ks) + 6.0;
    const double cap = (r + 8.0 * sqrt(r) + 32.0) * ((double)n / p.ns) + nn;
    if (r >= 0.25 * p.ns || cap >= 0.5 * n) { p.exhaustive = true; return p; }
    p.r = (int)ceil(r);
    p.cap = ((int)cap + 63) & ~63;
    return p;
}

template <int QT>
int run_batch(lsq_adc_state *st, hipStream_t s, float *dists, int *idx, const uint8_t *codes, const float *Q, const float *K, const float *dbnorms,
              const int *qsel, int q0, int nqb, int n, int m, int d, int nn, const Plan &P, int *h_fail, lsq_linscan_stats *stats, bool timed) {
    const int entries = m * LSQ_H, tiles = (nqb + QT - 1) / QT;
    int idbits = 1;                                   // ids 1 .. n: the records are (distance key << idbits | id), sorted on their 32 + idbits bits
    while (idbits < 32 && ((uint64_t)n >> idbits) != 0) ++idbits;
    LSQ_TRY(st->lut.ensure(sizeof(float) * (size_t)tiles * entries * QT));
    LSQ_TRY(st->seg.ensure(sizeof(int) * 2 * (size_t)nqb));
    LSQ_TRY(st->fail.ensure(sizeof(int) * (size_t)nqb));
    int *begin = st->seg.as<int>(), *end = begin + nqb;
    if (timed) LSQ_HIP(hipEventRecord(st->ev[0], s));
    LSQ_TRY(launch_lut<QT>(s, Q, K, qsel, q0, nqb, d, m, st->lut.as<float>()));
    if (timed) LSQ_HIP(hipEventRecord(st->ev[1], s));
    const unsigned gblocks = (unsigned)((nqb + 255) / 256);
    int seg_len;
    if (P.exhaustive) {
        seg_len = n;
        LSQ_TRY(st->keys_a.ensure(sizeof(uint64_t) * (size_t)nqb * n));
        LSQ_TRY(st->keys_b.ensure(sizeof(uint64_t) * (size_t)nqb * n));
        if (timed) { LSQ_HIP(hipEventRecord(st->ev[2], s)); }
        LSQ_TRY((launch_scan<QT, 1>(s, st->lut.as<float>(), codes, dbnorms, n, m, nqb, 1, n, nullptr, nullptr, n, st->keys_a.as<uint64_t>(), idbits)));
        if (timed) LSQ_HIP(hipEventRecord(st->ev[3], s));
        hipLaunchKernelGGL(adc_segments_kernel, dim3(gblocks), dim3(256), 0, s, (const unsigned *)nullptr, nqb, n, nn, begin, end, (int *)nullptr);
        LSQ_TRY(sort_segments(st, s, st->keys_a.as<uint64_t>(), st->keys_b.as<uint64_t>(), (int64_t)nqb * n, nqb, begin, end, 32 + idbits));
        hipLaunchKernelGGL(adc_gather_kernel, dim3((unsigned)((nn + 255) / 256 < 64 ? (nn + 255) / 256 : 64), (unsigned)nqb), dim3(256), 0, s,
                           st->keys_b.as<uint64_t>(), (const int *)nullptr, qsel, q0, nqb, n, nn, dists, idx, idbits);
        if (stats) stats->candidates += (int64_t)nqb * n;
    } else {
        seg_len = P.cap;
        const size_t items = (size_t)nqb * (size_t)(P.cap > P.ns ? P.cap : P.ns);
        LSQ_TRY(st->keys_a.ensure(sizeof(uint64_t) * items));
        LSQ_TRY(st->keys_b.ensure(sizeof(uint64_t) * items));
        LSQ_TRY(st->tau.ensure(sizeof(uint32_t) * (size_t)nqb));
        LSQ_TRY(st->count.ensure(sizeof(unsigned) * (size_t)nqb));
        // sample -> thresholds
        LSQ_TRY((launch_scan<QT, 2>(s, st->lut.as<float>(), codes, dbnorms, n, m, nqb, P.stride, P.ns, nullptr, nullptr, P.ns, st->keys_b.as<uint64_t>(), idbits)));
        hipLaunchKernelGGL(adc_rank_select_kernel, dim3((unsigned)nqb), dim3(256), 0, s, st->keys_b.as<uint32_t>(), P.r, st->tau.as<uint32_t>());
        LSQ_HIP(hipMemsetAsync(st->count.p, 0, sizeof(unsigned) * (size_t)nqb, s));
        if (timed) LSQ_HIP(hipEventRecord(st->ev[2], s));
        // the scan proper
        LSQ_TRY((launch_scan<QT, 0>(s, st->lut.as<float>(), codes, dbnorms, n, m, nqb, 1, n, st->tau.as<uint32_t>(), st->count.as<unsigned>(), P.cap,
                                    st->keys_a.as<uint64_t>(), idbits)));
        if (timed) LSQ_HIP(hipEventRecord(st->ev[3], s));
        hipLaunchKernelGGL(adc_segments_kernel, dim3(gblocks), dim3(256), 0, s, st->count.as<unsigned>(), nqb, P.cap, nn, begin, end, st->fail.as<int>());
        LSQ_TRY(sort_segments(st, s, st->keys_a.as<uint64_t>(), st->keys_b.as<uint64_t>(), (int64_t)nqb * P.cap, nqb, begin, end, 32 + idbits));
        hipLaunchKernelGGL(adc_gather_kernel, dim3((unsigned)((nn + 255) / 256 < 64 ? (nn + 255) / 256 : 64), (unsigned)nqb), dim3(256), 0, s,
                           st->keys_b.as<uint64_t>(), st->fail.as<int>(), qsel, q0, nqb, P.cap, nn, dists, idx, idbits);
        LSQ_HIP(hipMemcpyAsync(h_fail, st->fail.p, sizeof(int) * (size_t)nqb, hipMemcpyDeviceToHost, s));
        if (stats) {
            std::vector<unsigned> hc((size_t)nqb);
            LSQ_HIP(hipMemcpyAsync(hc.data(), st->count.p, sizeof(unsigned) * (size_t)nqb, hipMemcpyDeviceToHost, s));
            LSQ_HIP(hipStreamSynchronize(s));
            for (unsigned c : hc) stats->candidates += c;
        }
    }
    (void)seg_len;
    LSQ_HIP(hipGetLastError());
    if (timed) {
        LSQ_HIP(hipEventRecord(st->ev[4], s));
        LSQ_HIP(hipEventSynchronize(st->ev[4]));
        float a = 0, b = 0, c = 0, e = 0;
        LSQ_HIP(hipEventElapsedTime(&a, st->ev[0], st->ev[1]));
        LSQ_HIP(hipEventElapsedTime(&b, st->ev[1], st->ev[2]));
        LSQ_HIP(hipEventElapsedTime(&c, st->ev[2], st->ev[3]));
        LSQ_HIP(hipEventElapsedTime(&e, st->ev[3], st->ev[4]));
        stats->lut_ms += a; stats->sample_ms += b; stats->scan_ms += c; stats->select_ms += e;
    } else {
        LSQ_HIP(hipStreamSynchronize(s));
    }
    return LSQ_OK;
}

}  // namespace

// All pointers are device pointers.  force_exhaustive: test hook (every query by the exhaustive road).
int lsq_adc_search(hipStream_t s, lsq_adc_state **pst, float *dists, int *idx, const uint8_t *codes, const float *Q, const float *K,
                   const float *dbnorms, int nq, int n, int m, int d, int nn, int force_exhaustive, int sample_override, lsq_linscan_stats *stats,
                   int timed) {
    if (!*pst) *pst = new lsq_adc_state();
    lsq_adc_state *st = *pst;
    if (timed) for (hipEvent_t &e : st->ev) if (!e) LSQ_HIP(hipEventCreate(&e));
    Plan P = make_plan(n, nn, force_exhaustive);
    if (!P.exhaustive && sample_override > 0) {      // test hook: a deliberately wrong threshold rank (forces the fallback road)
        P.r = sample_override < P.ns ? sample_override : P.ns;
    }
    const bool wide = m <= 8;                        // 16 queries per block up to m = 8 (128 KiB of tables), 8 above
    const int64_t per_query = P.exhaustive ? n : (P.cap > P.ns ? P.cap : P.ns);
    // batches of queries: two key buffers of per_query entries each, at most ~2 GiB apiece, and hipcub counts items in int
    int64_t qb = (int64_t)(1u << 28) / per_query;
    if (qb < 1) qb = 1;
    if (qb > 16384) qb = 16384;
    if (qb > 16) qb &= ~(int64_t)15;
    std::vector<int> h_fail;
    std::vector<int> failed;
    for (int q0 = 0; q0 < nq; q0 += (int)qb) {
        const int nqb = (int)std::min<int64_t>(qb, nq - q0);
        h_fail.assign((size_t)nqb, 0);
        if (wide) LSQ_TRY(run_batch<16>(st, s, dists, idx, codes, Q, K, dbnorms, nullptr, q0, nqb, n, m, d, nn, P, h_fail.data(), stats, timed != 0));
        else LSQ_TRY(run_batch<8>(st, s, dists, idx, codes, Q, K, dbnorms, nullptr, q0, nqb, n, m, d, nn, P, h_fail.data(), stats, timed != 0));
        if (!P.exhaustive) {
            LSQ_HIP(hipStreamSynchronize(s));
            for (int t = 0; t < nqb; ++t) if (h_fail[(size_t)t]) failed.push_back(q0 + t);
        }
        if (stats) stats->batches += 1;
    }
    if (!failed.empty()) {
        // the exhaustive road for the queries whose candidate list came out short or overflowed
        Plan E = P;
        E.exhaustive = true; E.cap = n;
        int64_t fb = (int64_t)(1u << 28) / n;
        if (fb < 1) fb = 1;
        if (fb > 16) fb &= ~(int64_t)15;
        LSQ_TRY(st->qsel.ensure(sizeof(int) * failed.size()));
        LSQ_HIP(hipMemcpyAsync(st->qsel.p, failed.data(), sizeof(int) * failed.size(), hipMemcpyHostToDevice, s));
        for (size_t f0 = 0; f0 < failed.size(); f0 += (size_t)fb) {
            const int nqb = (int)std::min<size_t>((size_t)fb, failed.size() - f0);
            const int *qs = st->qsel.as<int>() + f0;
            if (wide) LSQ_TRY(run_batch<16>(st, s, dists, idx, codes, Q, K, dbnorms, qs, 0, nqb, n, m, d, nn, E, nullptr, stats, timed != 0));
            else LSQ_TRY(run_batch<8>(st, s, dists, idx, codes, Q, K, dbnorms, qs, 0, nqb, n, m, d, nn, E, nullptr, stats, timed != 0));
            if (stats) stats->batches += 1;
        }
        if (stats) stats->fallback_queries += (int64_t)failed.size();
    }
    if (stats) { stats->queries += nq; stats->codes = n; stats->exhaustive = P.exhaustive ? 1 : 0; stats->threshold_rank = P.r; stats->list_capacity = P.cap; }
    LSQ_HIP(hipStreamSynchronize(s));
    return LSQ_OK;
}

// host-buffer entry: upload, search, download
int lsq_adc_search_host(hipStream_t s, lsq_adc_state **pst, float *dists, int *idx, const unsigned char *codes, const float *Q, const float *K,
                        const float *dbnorms, int nq, int n, int m, int d, int nn, int force_exhaustive, int sample_override, lsq_linscan_stats *stats,
                        int timed) {
    if (!*pst) *pst = new lsq_adc_state();
    lsq_adc_state *st = *pst;
    LSQ_TRY(st->h_codes.ensure((size_t)n * m + 16));
    LSQ_TRY(st->h_q.ensure(sizeof(float) * (size_t)nq * d));
    LSQ_TRY(st->h_k.ensure(sizeof(float) * (size_t)m * LSQ_H * d));
    LSQ_TRY(st->h_norms.ensure(sizeof(float) * (size_t)n));
    LSQ_TRY(st->h_dists.ensure(sizeof(float) * (size_t)nq * nn));
    LSQ_TRY(st->h_idx.ensure(sizeof(int) * (size_t)nq * nn));
    LSQ_HIP(hipMemcpyAsync(st->h_codes.p, codes, (size_t)n * m, hipMemcpyHostToDevice, s));
    LSQ_HIP(hipMemcpyAsync(st->h_q.p, Q, sizeof(float) * (size_t)nq * d, hipMemcpyHostToDevice, s));
    LSQ_HIP(hipMemcpyAsync(st->h_k.p, K, sizeof(float) * (size_t)m * LSQ_H * d, hipMemcpyHostToDevice, s));
    LSQ_HIP(hipMemcpyAsync(st->h_norms.p, dbnorms, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    LSQ_TRY(lsq_adc_search(s, pst, st->h_dists.as<float>(), st->h_idx.as<int>(), st->h_codes.as<uint8_t>(), st->h_q.as<float>(), st->h_k.as<float>(),
                           st->h_norms.as<float>(), nq, n, m, d, nn, force_exhaustive, sample_override, stats, timed));
    LSQ_HIP(hipMemcpyAsync(dists, st->h_dists.p, sizeof(float) * (size_t)nq * nn, hipMemcpyDeviceToHost, s));
    LSQ_HIP(hipMemcpyAsync(idx, st->h_idx.p, sizeof(int) * (size_t)nq * nn, hipMemcpyDeviceToHost, s));
    LSQ_HIP(hipStreamSynchronize(s));
    return LSQ_OK;
}
