// lsq_internal.h -- shared internals of liblsq_mi355x.so (gfx950 only; not a public header).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "../../include/lsq_mi355x.h"

#define LSQ_H 256          // candidates per codebook: one wave x float4 per lane
#define LSQ_MAX_M 16
#define LSQ_WALK_TRACE 64        // per-position (sweep * m + rank in the node order, mod 64) recomputed node updates
#define LSQ_WALK_COUNTERS (4 + LSQ_WALK_TRACE + 3)      // device counters of the walk kernel: [0] node updates recomputed, [1..3] staged / light / filtered block-node-updates, [4..67] trace, [68] node updates the filter refined exactly, [69] exact candidate evaluations of those, [70] node updates sent to f32 (outside the sampled level range)

// ---- tuning knobs -------------------------------------------------------------------------
// Tuning knobs (environment variables) exist in the tuning build only; the shipped library uses the measured defaults.
#ifdef LSQ_TUNING
#define LSQ_KNOB(name, def) ([] { static const int v_ = [] { const char *e_ = getenv(name); return e_ ? atoi(e_) : (def); }(); return v_; }())
#else
#define LSQ_KNOB(name, def) (def)
#endif

// ---- error plumbing ---------------------------------------------------------------------
void lsq_set_error(const char *fmt, ...);

#define LSQ_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            lsq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? LSQ_ENOMEM : LSQ_EHIP;                        \
        }                                                                                      \
    } while (0)

#define LSQ_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != LSQ_OK) return rc_; \
    } while (0)

// ---- a device buffer that only grows (owned by a context or by one of its sub-states) --------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool view = false;          // a window into another allocation (the context's block of small per-call words): never freed here; outgrown -> an allocation of its own
    int ensure(size_t bytes) {
        if (bytes <= cap) return LSQ_OK;
        if (view) { p = nullptr; cap = 0; view = false; }
        if (p) { LSQ_HIP(hipFree(p)); p = nullptr; cap = 0; }
        if (bytes == 0) return LSQ_OK;
        LSQ_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return LSQ_OK;
    }
    void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
    void window(void *base, size_t off, size_t bytes) { release(); p = static_cast<char *>(base) + off; cap = bytes; view = true; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- Philox4x32-10 (Random123; Salmon et al. SC'11), shared by host and device -----------
// Stream layout (build-defined, mirrored by oracle/lsq_oracle.c):
//   counter = (idx_lo, idx_hi, it, (domain << 16) | (word >> 2)),  key = (seed_lo, seed_hi),
//   word w of the stream = output[w & 3].
enum { LSQ_DOM_PERTURB = 1, LSQ_DOM_PERM = 2, LSQ_DOM_INIT = 3, LSQ_DOM_DATA = 4, LSQ_DOM_CODEBOOK = 5 };

struct lsq_u32x4 { uint32_t v[4]; };

__host__ __device__ inline lsq_u32x4 lsq_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    lsq_u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

__host__ __device__ inline lsq_u32x4 lsq_rng_block(uint64_t seed, uint64_t idx, uint32_t it, uint32_t domain, uint32_t block) {
    return lsq_philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), it, (domain << 16) | block,
                             (uint32_t)seed, (uint32_t)(seed >> 32));
}

__host__ __device__ inline uint32_t lsq_rng_word(uint64_t seed, uint64_t idx, uint32_t it, uint32_t domain, uint32_t w) {
    return lsq_rng_block(seed, idx, it, domain, w >> 2).v[w & 3];
}

__host__ __device__ inline uint32_t lsq_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// ---- 16-bit filtered walk: quantisation parameters (device memory, one per resident chunk) ---------------------------------
// Every term of a conditioned sum s[a] = U_j[a] + SUM_k T_jk[b_k][a] is ALSO held as a 16-bit level on ONE common step D_j per node:
//   U level = rint((u + g_j[a] + sigma_ij - loU) / D),  table level = rint((t - g_jk[a] - min of the centred row) / D)  (shifts shared by all
//   candidates of a node update, and per-candidate shifts moved from the tables to the unary: lsq_icmq.hip),  so that the sum of the m levels Q[a] fits 16 bits and s[a] = C_i + D * Q[a] up to (0.5 + 2^-5) D per term.  Rigorous
//   consequence (see icm_walkq_kernel): the exact fp32 argmin lies among the candidates with Q <= Qmin + window.  ok = 0 (non-finite or
//   degenerate bounds) sends the chunk to the fp32 walk instead.
struct lsq_q16_node {
    float loU, invD, D;        // U levels start at loU: the sampled range of the SHIFTED U_j widened by 1/8; tables use the exact range of their centred rows
    float hiq;                 // largest U level of the widened sampled range: a vector with a level outside [0, hiq] is flagged (beyond it the sum of
                               // the m levels could pass 65535 and wrap)
    int window;                // levels
    double slack;              // bound of |fp32 conditioned sum - real sum| + per-term level error, in s units: m (0.5 + 2^-5) D + eps
};
struct lsq_q16_params {
    int ok;
    int oor;                   // values the GEMM epilogue found outside the sampled level range (their vectors are flagged and take the f32 path)
    int nflag;                 // (vector, node) pairs flagged in this chunk: above 1 / filter_fallback_div of all pairs the host sends the whole chunk to the f32 walk
    lsq_q16_node node[LSQ_MAX_M];
};

// one kernel of each translation unit of the encode path: lsq_create asks for its attributes, which makes the runtime load that unit's code object
// THEN (HIP loads a unit's device code at its first use: ~10 ms that the first encode of a process would otherwise pay)
const void *lsq_probe_kernel_gemm();
const void *lsq_probe_kernel_icm();
const void *lsq_probe_kernel_icmq();

// ---- kernel launchers (implemented in the .hip files) ------------------------------------
// All pointers are device pointers; all launch on `s` and return immediately.

// D[off(r,c)] = chain_t( A[r][t] * (alpha * Bm[c][t]) ) (+ addv[c]);  r<M, c<N, t<Kd.
// off(r,c) = (c / h) * plane_stride + (c % h) + r * row_stride            (slice == 0, row-major planes)
//          = (c / h) * plane_stride + ((c % h) / slice) * (Mtot * slice) + (c % h) % slice + r * slice   (slice-major)
// Chain = k-ascending fmaf from +0.  The launch covers output rows [rbase, rbase + M) of a Mtot-row result (A points at
// row rbase): row r above counts from rbase -- lets the caller build the unaries panel by panel under the H2D copies.
// Optional second output (Dq != nullptr): the same values as 16-bit fixed-point levels in slice-major u16 planes of slice width slice_q,
// Dq[(c / h) * Mtot * h + ((c % h) / slice_q) * Mtot * slice_q + r * slice_q + (c % h) % slice_q] = rint((v + colshift[c] + sigma[r][c / h] - qp->node[c / h].loU) * invD).
int lsq_launch_chain_gemm(hipStream_t s, const float *A, const float *Bm, const float *addv, float alpha,
                          int64_t M, int N, int Kd, int h, int64_t plane_stride, int64_t row_stride, float *D, int slice,
                          int64_t Mtot, int64_t rbase, uint16_t *Dq = nullptr, int slice_q = 0, struct lsq_q16_params *qp = nullptr,
                          int64_t lda = 0, unsigned short *qflag = nullptr, unsigned *qrange = nullptr, int rts = 1, const float *sigma = nullptr, const float *colshift = nullptr);
// sigma (optional, [Mtot][N / h] floats) / colshift (optional, [N] floats): per-(row, plane) and per-column shifts added to a value before its LEVEL is
// taken (and in the range-only pass); the f32 output D is not shifted.  A shift common to all candidates of a node update cannot change its argmin (lsq_icmq.hip).
// lda: row stride of A in floats (0 = Kd).  rts: range-only pass over every rts-th 128-row panel of A.  qflag [Mtot] u16 (Mtot even-padded): bit j raised when a value of
// plane j fell outside the level range (Dq output).  qrange != nullptr: range-only pass, nothing stored; qrange[2 j], [2 j + 1] = min / max keys.
// sci[r] = chain_t(Kb[r][t]^2)
int lsq_launch_sqnorms(hipStream_t s, const float *Kb, int rows, int d, float *sci);

// Internal code records: cs bytes per vector (8 for m<=8, 16 for m<=16), byte j = code of codebook j.
static inline int lsq_code_stride(int m) { return m <= 8 ? 8 : 16; }

int lsq_launch_codes_expand(hipStream_t s, const uint8_t *tight, int64_t n, int m, uint8_t *rec);      // [n][m] -> [n][cs]
int lsq_launch_codes_compact(hipStream_t s, const uint8_t *rec, int64_t n, int m, uint8_t *tight);     // [n][cs] -> [n][m]
int lsq_launch_codes_from_i16(hipStream_t s, const int16_t *B, int64_t n, int m, int h, uint8_t *rec, int *bad_flag);
int lsq_launch_codes_to_i16(hipStream_t s, const uint8_t *rec, int64_t n, int m, int16_t *B);

// vsrc/vdst (optional): per-vector node-validity masks (bit j = code j is the argmin for the current other
// codes); a perturbation that changes a code clears the whole mask.
int lsq_launch_perturb(hipStream_t s, const uint8_t *src, uint8_t *dst, int64_t n, int m, int npert,
                       uint64_t seed, uint32_t it, uint64_t global_offset, const unsigned short *vsrc, unsigned short *vdst);
// LDS-walk schedule: one block walks all slices of its vector range; Ts = slice-major pair tables
int lsq_walk_slice_width(int m);      // 16 for m <= 8 (8 if LSQ_WALK_SL=8 is set: tuning knob), 8 above
int lsq_launch_tables_to_slices(hipStream_t s, const float *T, float *Ts, int m, int sl);
// valid (optional): validity masks, maintained by the kernel; use_skip: skip vectors whose bit j is set (exact);
// active_total (optional): += number of vectors actually recomputed; ablation != 0: timing-only variants (m = 8), garbage results.
// U is the slice-major unary buffer of ALL nodes; T (optional) the row-major tables for light blocks' L2 gathers; order[nnodes] = node updates run back to back inside the launch
// (a block owns its vectors for the whole launch): 1 entry = one node update, icmiter*m entries = a whole ILS iteration.
void lsq_walk_geometry(int64_t n, int m, int *per_pass, int *npass, int *pp_cap);      // segments of the walk kernel over n vectors
int lsq_launch_icm_wave(hipStream_t s, const float *U, const float *T, uint8_t *rec, unsigned short *valid, int64_t n, int m, const int32_t *order,
                        int nnodes, int pos0, int use_skip, unsigned long long *active_total, const uint8_t *ref_rec, const unsigned short *ref_valid,
                        const int *idle_if_set = nullptr);      // chunks in which every block would be light: a wave owns its vectors through the launch
int lsq_launch_icm_walk(hipStream_t s, const float *U, const float *Ts, const float *T, uint8_t *rec, unsigned short *valid, int64_t n, int m,
                        const int32_t *order, int nnodes, int pos0, int use_skip, unsigned long long *active_total, int ablation, int light,
                        const uint8_t *ref_rec, const unsigned short *ref_valid, const int *idle_if_set = nullptr);
// idle_if_set (optional, device): the launch does nothing when *idle_if_set != 0 (the filtered walk handled it)
// 16-bit filtered walk (lsq_icmq.hip).  lsq_launch_q16_prepare: per chunk, after the pair tables and before the unary GEMM -- bounds,
// parameters P and the 16-bit slice tables Tq [m][256/SLQ][m-1][256][SLQ]; tables_changed = 1 on the first chunk of a call.
// bad (1 int), trange (3 m m floats), qrange (2 * 16 + 2 u32): scratch.  lsq_launch_icm_walkq: same contract as lsq_launch_icm_walk plus
// Uq (the GEMM's u16 planes), Tq and P; the caller launches it only after reading the chunk's verdict (P->ok, P->nflag) on the host.
int lsq_q16_slice_width(int m);
int lsq_launch_q16_prepare(hipStream_t s, const float *X, int64_t n, int d, const float *K, const float *sci, const float *T, int m, uint16_t *Tq,
                           int *bad, float *trange, unsigned *qrange, unsigned short *qflag, lsq_q16_params *P, int tables_changed,
                           float *rowmin, float *means, float *sigma, float *colmean, float *colshift, const float *Xsample = nullptr,
                           int64_t nsample_rows = 0, float *sigma_sample = nullptr);
// Xsample (optional; the host-buffer pipeline): the rows the strided sample pass would read -- every rts-th 128-row panel, lsq_q16_sample_rows -- already
// compacted on the device; the level parameters come from them alone (max |sigma| widened x2) and X itself is not touched: its sigma are computed panel by
// panel as the panels land (lsq_launch_unary_shift_panel: vectors beyond the assumed |sigma| bound are flagged for the f32 routine).
int lsq_q16_sample_rows(int64_t n, int d, int64_t *rts_out);      // -> number of 128-row sample panels; *rts_out = the panel stride
int lsq_launch_unary_shift_panel(hipStream_t s, const float *Xp, int64_t rows, int d, int m, const float *means, float *sigma_p, unsigned *qrange,
                                 unsigned short *qflag, int64_t row0, lsq_q16_params *P);
// rowmin [m*m*256], means [m*d], colmean [m*m*256], colshift [m*256] (per call), sigma [n*m] (per chunk); trange: 3 floats per pair table
int lsq_launch_icm_walkq(hipStream_t s, const float *U, const uint16_t *Uq, const uint16_t *Tq, const float *T, uint8_t *rec, unsigned short *valid,
                         int64_t n, int m, const int32_t *order, int nnodes, int pos0, int use_skip, unsigned long long *active_total, int light,
                         const uint8_t *ref_rec, const unsigned short *ref_valid, const lsq_q16_params *P, const unsigned short *qflag,
                         const unsigned *gate = nullptr);
// option "async": the chunk's road decided on the device (lsq_icmq.hip): road[0] = 2 filtered walk / 0 f32 walk, road[1] = chunks handed over
int lsq_launch_q16_road(hipStream_t s, const lsq_q16_params *P, unsigned *road, int64_t pairs, int64_t fallback_div);
int lsq_launch_q16_probe(hipStream_t s, const unsigned long long *probe, unsigned long long *totals, unsigned *road, int64_t probe_div);
// gate (optional, device; option "async"): the launch runs only when *gate == 2 (the road word names the filtered walk)
// ref_rec / ref_valid (optional, read-only): the vectors' current records and their validity masks; a candidate that becomes
// equal to its current record inherits those bits (exact: validity depends on the code tuple only)
// light: blocks with <= light active vectors gather table columns from L2 instead of staging slices (-1 = default 256)
// cost of `rec`; mode 0: prev[i] = cost.  mode 1 (accept): if cost < prev[i] { cur[i] = rec[i]; prev[i] = cost }
// and counters[0] += (#cost == prev), counters[1] += (#cost < prev)   (counters: 2 x uint64 on device)
// perturbation for the NEXT ILS iteration fused into the cost kernel's exit (on = 0: none): dst / vdst receive the perturbed copy of every vector's
// final record / validity word (dst may be the candidate array the kernel has just judged)
struct lsq_perturb_next { int on; int m, npert; uint32_t it; uint64_t seed, goff; uint8_t *dst; unsigned short *vdst; int abl; };      // abl: timing-only ablations of the cost kernel (tuning build; 0 in the shipped library)
int lsq_launch_cost(hipStream_t s, const float *X, const float *K, const uint8_t *rec, uint8_t *cur, float *prev,
                    unsigned long long *counters, int64_t n, int d, int m, int mode,
                    const unsigned short *vnew, unsigned short *vcur,
                    const lsq_perturb_next *next = nullptr);      // accept also copies the validity mask
// *sum += SUM_i v[i]  (f64)
int lsq_launch_sum_f64(hipStream_t s, const float *v, int64_t n, double *sum);

int lsq_launch_synth_data_u8(hipStream_t s, uint64_t seed, uint64_t global_offset, int64_t n, int d, float *X);
int lsq_launch_randinit(hipStream_t s, uint64_t seed, uint64_t global_offset, int64_t n, int m, int h, uint8_t *tight);
int lsq_launch_synth_codebooks(hipStream_t s, uint64_t seed, int m, int h, int d, float *K);

// ---- the initialisers' two data-parallel kernels (lsq_init.hip) ---------------------------------------------------------------
// U: row-major f32 unary planes [m][n][256] (the unary GEMM with slice = 0), T: the pair tables of prepare_tables; codes: tight [n][m] u8, 0-based.
int lsq_launch_viterbi(hipStream_t s, const float *U, const float *T, int64_t n, int m, uint8_t *codes);            // encode_chain.jl:2-89
int lsq_launch_unary_argmin(hipStream_t s, const float *U, int64_t n, int m, uint8_t *codes, float *minval);     // PQ.jl:12-41, kmeans.jl:6-75; minval optional [n][m]

// ---- device ADC scan (lsq_adc.hip) ----------------------------------------------------------------------------------------------------
struct lsq_adc_state;      // buffers of the scan, owned by the context
void lsq_adc_free(lsq_adc_state *st);
// device pointers; force_exhaustive / rank_override: test hooks (options "linscan_exhaustive", "linscan_rank")
int lsq_adc_search(hipStream_t s, lsq_adc_state **st, float *dists, int *idx, const uint8_t *codes, const float *Q, const float *K, const float *dbnorms,
                   int nq, int n, int m, int d, int nn, int force_exhaustive, int rank_override, lsq_linscan_stats *stats, int timed);
int lsq_adc_search_host(hipStream_t s, lsq_adc_state **st, float *dists, int *idx, const unsigned char *codes, const float *Q, const float *K,
                        const float *dbnorms, int nq, int n, int m, int d, int nn, int force_exhaustive, int rank_override, lsq_linscan_stats *stats,
                        int timed);

// ---- quantize_norms on the device (lsq_norms.hip): codes [n][stride] u8 0-based; any of the four outputs may be null ----------------------
int lsq_launch_quantize_norms(hipStream_t s, const uint8_t *codes, int stride, const float *K, const float *cb, int ncb, int64_t n, int d, int m,
                              uint8_t *idx0, int16_t *idx1, float *dbnorms, float *norms);

// ---- LSQR codebook update on the device (lsq_lsqr.hip) ---------------------------------------------------------------------------------------
struct lsq_lsqr_state;
void lsq_lsqr_free(lsq_lsqr_state *st);
int lsq_lsqr_update_codebooks(hipStream_t s, lsq_lsqr_state **st, const float *dX, const uint8_t *dcodes, int d, int64_t n, int m, float *dK, int *iters_out);
