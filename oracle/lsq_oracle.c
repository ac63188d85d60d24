/*
 * oracle/lsq_oracle.c -- CPU restatement of the reference's LSQ encoding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * local-search-quantization_amd/, its C-ABI library, bench.py's GPU leg) may
 * import, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg use it -- as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED (encode path): the reference (una-dinosauria/local-search-
 * quantization) ships no tests, golden vectors or fixtures, is Julia-0.6-only
 * (Julia is not installed here) and is not bit-reproducible against itself
 * (BLAS-defined GEMM summation order, @simd reassociation, Mersenne-Twister /
 * curand(clock()) seeding).  This file restates the algorithm line by line from the
 * cited reference sources and FREEZES the choices the reference leaves open:
 *
 *   [build-defined 1] every dot product is a k-ascending fmaf chain from +0
 *                     (bitwise what gfx950's fp32 MFMA computes);
 *   [build-defined 2] the cost reduction order: 64 strided partial sums
 *                     (t = l + 64*q, q ascending) then a balanced pairwise tree;
 *   [build-defined 3] the RNG: Philox4x32-10 (Salmon et al., SC'11; the published
 *                     Random123 algorithm, checked against its known-answer vectors
 *                     in tests/) keyed by (seed), counter = (global vector index,
 *                     ILS iteration, domain|block) -- independent of sharding;
 *   [build-defined 4] the objective is accumulated in float64.
 *
 * Everything else is the reference's CPU-path behaviour
 * (src/encodings/encode_icm.jl, src/utils.jl); each function cites the lines it
 * follows.  The search half (linear scan) IS pinned: see oracle/Makefile (`_ref`).
 *
 * Layouts at this boundary are the Julia column-major buffers read in place:
 *   X  d x n  f32      -> X[i*d + t]
 *   K  = hcat(C...) d x (m*h) f32 -> K[(j*h + a)*d + t]
 *   B  m x n  Int16, 1-based      -> B[i*m + j]
 * Internal codes are uint8, 0-based, [n][m].
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: no implicit FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EXPORT __attribute__((visibility("default")))

enum { ORC_DOM_PERTURB = 1, ORC_DOM_PERM = 2, ORC_DOM_INIT = 3, ORC_DOM_DATA = 4, ORC_DOM_CODEBOOK = 5 };

/* ------------------------------------------------------------------ RNG ---- */

/* Philox4x32-10, Random123 (Salmon, Moraes, Dror, Shaw; SC'11).  Third-party
 * published algorithm; known-answer vectors are checked in tests/test_oracle_rng.py. */
ORC_EXPORT void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 32-bit random word number `w` of the stream (seed; idx, it, domain). */
ORC_EXPORT uint32_t orc_rng_word(uint64_t seed, uint64_t idx, uint32_t it, uint32_t domain, uint32_t w) {
    uint32_t ctr[4] = { (uint32_t)idx, (uint32_t)(idx >> 32), it, (domain << 16) | (w >> 2) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    return out[w & 3];
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* Node visiting order for ILS iteration `it`.
 * Reference: encode_icm.jl:46-49 (`to_look = randperm(m)` once per call, shared by
 * every vector); encode_icm_cuda.jl:141-144 (once per ILS iteration).  The RNG is
 * build-defined: Fisher-Yates driven by Philox words of domain PERM. */
ORC_EXPORT void orc_perm(uint64_t seed, uint32_t it, int m, int randord, int32_t *order) {
    for (int p = 0; p < m; ++p) order[p] = p;
    if (!randord) return;
    for (int p = m - 1; p >= 1; --p) {
        uint32_t r = orc_rng_word(seed, 0, it, ORC_DOM_PERM, (uint32_t)(m - 1 - p));
        int q = (int)mulhi32(r, (uint32_t)(p + 1));
        int32_t tmp = order[p]; order[p] = order[q]; order[q] = tmp;
    }
}

/* Perturb `npert` distinct positions of one code, ascending selection scan.
 * Reference: encode_icm.jl:55-70 (sample(1:m,npert,replace=false,ordered=true);
 * rand(1:h)) and the selection-sampling scan of cudautils.cu:48-70 (take position p
 * with probability n_needed/n_left).  Integer form: take p iff
 * mulhi32(r_p, left) < need;  value = mulhi32(r'_p, h).  Word p selects, word 16+p
 * is the value (fixed word positions so a lane-parallel GPU version is trivial). */
ORC_EXPORT void orc_perturb(uint64_t seed, uint64_t gidx, uint32_t it, int m, int h, int npert, uint8_t *code) {
    int need = npert < m ? npert : m;
    for (int p = 0; p < m && need > 0; ++p) {
        uint32_t left = (uint32_t)(m - p);
        uint32_t r = orc_rng_word(seed, gidx, it, ORC_DOM_PERTURB, (uint32_t)p);
        if (mulhi32(r, left) < (uint32_t)need) {
            uint32_t rv = orc_rng_word(seed, gidx, it, ORC_DOM_PERTURB, (uint32_t)(16 + p));
            code[p] = (uint8_t)mulhi32(rv, (uint32_t)h);
            --need;
        }
    }
}

/* randinit: initializations.jl:2-8 (`rand(1:h, m, n)`); 1-based Int16 out, [n][m]. */
ORC_EXPORT void orc_randinit(uint64_t seed, uint64_t global_offset, long n, int m, int h, int16_t *B) {
    for (long i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j)
            B[i * m + j] = (int16_t)(1 + mulhi32(orc_rng_word(seed, global_offset + (uint64_t)i, 0, ORC_DOM_INIT, (uint32_t)j), (uint32_t)h));
}

/* Synthetic SIFT-like data: X[i][t] = float(uniform int 0..255). */
ORC_EXPORT void orc_synth_data_u8(uint64_t seed, uint64_t global_offset, long n, int d, float *X) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i)
        for (int t = 0; t < d; ++t)
            X[i * (long)d + t] = (float)(orc_rng_word(seed, global_offset + (uint64_t)i, (uint32_t)(t >> 10), ORC_DOM_DATA, (uint32_t)(t & 1023)) >> 24);
}

/* -------------------------------------------------------------- tables ---- */

/* sci[j*h+a] = ||c_{j,a}||^2  -- utils.jl:109 `diag(C[i]'*C[i])`; chain order build-defined. */
ORC_EXPORT void orc_sqnorms(const float *K, int mh, int d, float *sci) {
    for (int r = 0; r < mh; ++r) {
        float acc = 0.0f;
        for (int t = 0; t < d; ++t) acc = fmaf(K[(long)r * d + t], K[(long)r * d + t], acc);
        sci[r] = acc;
    }
}

/* T[((j*m + k)*h + b)*h + a] = 2<c_{j,a}, c_{k,b}>  -- utils.jl:125-144
 * (`binaries[idx] = 2*C[i]'*C[j]`) plus the transposed copies of
 * encode_icm.jl:25-28, stored as one table per ORDERED pair so that the column added
 * when codebook k holds code b while node j is minimised (encode_icm.jl:84-101) is
 * the contiguous run T[j][k][b][0..h).  Diagonal blocks (j==k) are filled too but
 * never read.  chain(2*c_ja[t] * c_kb[t]) is bitwise symmetric in (j,a)<->(k,b). */
ORC_EXPORT void orc_tables(const float *K, int m, int h, int d, float *T) {
    /* Kt[j][t][a] so the inner loop runs over a */
    float *Kt = (float *)malloc(sizeof(float) * (size_t)m * h * d);
    for (int j = 0; j < m; ++j)
        for (int a = 0; a < h; ++a)
            for (int t = 0; t < d; ++t)
                Kt[((size_t)j * d + t) * h + a] = K[((size_t)j * h + a) * d + t];
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < m; ++j)
        for (int k = 0; k < m; ++k) {
            const float *Ktj = Kt + (size_t)j * d * h;
            for (int b = 0; b < h; ++b) {
                float *row = T + (((size_t)j * m + k) * h + b) * h;
                const float *ckb = K + ((size_t)k * h + b) * d;
                for (int a = 0; a < h; ++a) row[a] = 0.0f;
                for (int t = 0; t < d; ++t) {
                    const float y = ckb[t];
                    const float *kt = Ktj + (size_t)t * h;
                    for (int a = 0; a < h; ++a) row[a] = fmaf(2.0f * kt[a], y, row[a]);
                }
            }
        }
    free(Kt);
}

/* Unaries of ONE vector: u[j*h + a] = chain(-2 c_{j,a}[t] * x[t]) + sci[j*h+a]
 * -- utils.jl:108 (`-2*C[i]'*X`) then :112-118 (`ui[k,j] += sci[k]`: one rounded add).
 * Kt is [m][d][h]. */
static void unaries_one(const float *x, const float *Kt, const float *sci, int d, int m, int h, float *u) {
    for (int j = 0; j < m; ++j) {
        float *uj = u + (size_t)j * h;
        const float *Ktj = Kt + (size_t)j * d * h;
        for (int a = 0; a < h; ++a) uj[a] = 0.0f;
        for (int t = 0; t < d; ++t) {
            const float xv = x[t];
            const float *kt = Ktj + (size_t)t * h;
            for (int a = 0; a < h; ++a) uj[a] = fmaf(-2.0f * kt[a], xv, uj[a]);
        }
        for (int a = 0; a < h; ++a) uj[a] = uj[a] + sci[(size_t)j * h + a];
    }
}

static float *make_Kt(const float *K, int m, int h, int d) {
    float *Kt = (float *)malloc(sizeof(float) * (size_t)m * h * d);
    for (int j = 0; j < m; ++j)
        for (int a = 0; a < h; ++a)
            for (int t = 0; t < d; ++t)
                Kt[((size_t)j * d + t) * h + a] = K[((size_t)j * h + a) * d + t];
    return Kt;
}

/* U[(j*n + i)*h + a]  (the reference's `unaries[j]`, an h x n column-major matrix). */
ORC_EXPORT void orc_unaries(const float *X, const float *K, long n, int d, int m, int h, float *U) {
    float *Kt = make_Kt(K, m, h, d);
    float *sci = (float *)malloc(sizeof(float) * (size_t)m * h);
    orc_sqnorms(K, m * h, d, sci);
#pragma omp parallel
    {
        float *u = (float *)malloc(sizeof(float) * (size_t)m * h);
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i) {
            unaries_one(X + (size_t)i * d, Kt, sci, d, m, h, u);
            for (int j = 0; j < m; ++j)
                memcpy(U + ((size_t)j * n + i) * h, u + (size_t)j * h, sizeof(float) * h);
        }
        free(u);
    }
    free(sci);
    free(Kt);
}

/* ---------------------------------------------------------------- cost ---- */

/* veccost -- utils.jl:225-254: CBi = sum_k C_k[:,B[k,i]] (k ascending, from 0),
 * cost = sum_t (CBi[t] - x[t])^2.  The reference's @simd reduction order is
 * compiler-defined; [build-defined 2]: lane partials then a pairwise tree. */
static float cost_one(const float *x, const float *K, const uint8_t *code, int d, int m, int h) {
    float p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0f;
    for (int t = 0; t < d; ++t) {
        float cb = 0.0f;
        for (int k = 0; k < m; ++k) cb = cb + K[((size_t)k * h + code[k]) * d + t];
        float r = cb - x[t];
        float sq = r * r;              /* -ffp-contract=off: never fused into the add */
        p[t & 63] = p[t & 63] + sq;    /* t = l + 64 q, q ascending */
    }
    for (int s = 1; s < 64; s <<= 1)
        for (int l = 0; l < 64; l += 2 * s) p[l] = p[l] + p[l + s];
    return p[0];
}

ORC_EXPORT void orc_veccost(const float *X, const float *K, const uint8_t *codes, long n, int d, int m, int h, float *cost) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) cost[i] = cost_one(X + (size_t)i * d, K, codes + (size_t)i * m, d, m, h);
}

/* ------------------------------------------------------------ ICM node ---- */

/* One ICM node update for one vector -- encode_icm.jl:76-119.
 *   s = U_j (copy, :78-81); for k ascending, k != j: s += T[j][k][code_k] (:84-101,
 *   plain f32 adds); new code = first index of the minimum under a strict `<` scan
 *   from index 0 (:105-119). */
static int node_update(const float *uj, const float *T, const uint8_t *code, int j, int m, int h, float *s) {
    for (int a = 0; a < h; ++a) s[a] = uj[a];
    for (int k = 0; k < m; ++k) {
        if (k == j) continue;
        const float *col = T + (((size_t)j * m + k) * h + code[k]) * h;
        for (int a = 0; a < h; ++a) s[a] = s[a] + col[a];
    }
    float minv = s[0];
    int mini = 0;
    for (int a = 1; a < h; ++a)
        if (s[a] < minv) { minv = s[a]; mini = a; }
    return mini;
}

ORC_EXPORT int orc_icm_node(const float *uj, const float *T, const uint8_t *code, int j, int m, int h) {
    float s[1024];
    if (h > 1024) return -1;
    return node_update(uj, T, code, j, m, h, s);
}

/* ------------------------------------------------ whole call, per-vector ---- */

/* The full `encode_icm_cuda`-shaped call (encode_icm_cuda.jl:253-296, :22-234) with
 * the reference CPU path's semantics for every ILS iteration (encode_icm.jl:131-189):
 *   prev = cost(B);  N = B;  perturb N (:55-70);  J sweeps over `order` (:72-125);
 *   new = cost(N);   B = N where new < prev strictly (:178-186);
 *   snapshot + objective at the iterations listed in ilsiters
 *   (encode_icm_cuda.jl:211-222; objective = qerror, utils.jl:257-285, in f64).
 * Vectors never interact, so running all node updates of one vector back to back is
 * exactly equivalent to the reference's whole-array sweeps.
 *
 * stats (optional, 2*I doubles): per ILS iteration, #equal and #better
 * (encode_icm.jl:180-184).  Returns 0, or <0 on bad arguments. */
ORC_EXPORT int orc_encode_icm(const float *X, const int16_t *B, const float *K, int d, long n, int m, int h,
                              const int64_t *ilsiters, int nr, int icmiter, int npert, int randord,
                              uint64_t seed, uint64_t global_offset, int16_t *Bs, float *objs, double *stats) {
    if (d < 1 || n < 0 || m < 1 || m > 16 || h < 1 || h > 256 || nr < 1 || icmiter < 0 || npert < 0) return -1;
    long I = 0;
    for (int r = 0; r < nr; ++r) { if (ilsiters[r] < 1) return -2; if (ilsiters[r] > I) I = ilsiters[r]; }
    for (long q = 0; q < n * m; ++q) if (B[q] < 1 || B[q] > h) return -3;

    float *Kt = make_Kt(K, m, h, d);
    float *sci = (float *)malloc(sizeof(float) * (size_t)m * h);
    float *T = (float *)malloc(sizeof(float) * (size_t)m * m * h * h);
    orc_sqnorms(K, m * h, d, sci);
    orc_tables(K, m, h, d, T);
    int32_t *orders = (int32_t *)malloc(sizeof(int32_t) * (size_t)I * m);
    for (long it = 0; it < I; ++it) orc_perm(seed, (uint32_t)it, m, randord, orders + it * m);
    double *objsum = (double *)calloc((size_t)nr, sizeof(double));
    if (stats) memset(stats, 0, sizeof(double) * 2 * (size_t)I);

#pragma omp parallel
    {
        float *u = (float *)malloc(sizeof(float) * (size_t)m * h);
        float *s = (float *)malloc(sizeof(float) * (size_t)h);
        double *lobj = (double *)calloc((size_t)nr, sizeof(double));
        double *lst = (double *)calloc(2 * (size_t)I, sizeof(double));
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i) {
            const float *x = X + (size_t)i * d;
            uint8_t cur[16], nw[16];
            for (int j = 0; j < m; ++j) cur[j] = (uint8_t)(B[i * m + j] - 1);
            unaries_one(x, Kt, sci, d, m, h, u);
            float prev = cost_one(x, K, cur, d, m, h);
            for (long it = 0; it < I; ++it) {
                memcpy(nw, cur, (size_t)m);
                orc_perturb(seed, global_offset + (uint64_t)i, (uint32_t)it, m, h, npert, nw);
                const int32_t *order = orders + it * m;
                for (int sw = 0; sw < icmiter; ++sw)
                    for (int q = 0; q < m; ++q) {
                        int j = order[q];
                        nw[j] = (uint8_t)node_update(u + (size_t)j * h, T, nw, j, m, h, s);
                    }
                float nc = cost_one(x, K, nw, d, m, h);
                if (nc == prev) lst[2 * it] += 1.0;
                if (nc < prev) { lst[2 * it + 1] += 1.0; memcpy(cur, nw, (size_t)m); prev = nc; }
                for (int r = 0; r < nr; ++r)
                    if (ilsiters[r] == it + 1) {
                        for (int j = 0; j < m; ++j) Bs[((size_t)r * n + i) * m + j] = (int16_t)(cur[j] + 1);
                        lobj[r] += (double)prev;
                    }
            }
        }
#pragma omp critical
        {
            for (int r = 0; r < nr; ++r) objsum[r] += lobj[r];
            if (stats) for (long q = 0; q < 2 * I; ++q) stats[q] += lst[q];
        }
        free(u); free(s); free(lobj); free(lst);
    }
    /* note: the f64 accumulation order differs between thread counts (1e-16 rel.);
     * the objective is compared at 1e-5 relative (north_star). */
    for (int r = 0; r < nr; ++r) objs[r] = (float)(n > 0 ? objsum[r] / (double)n : 0.0);
    free(objsum); free(orders); free(T); free(sci); free(Kt);
    return 0;
}

/* ------------------------------------------------ the worker, per-vector ---- */

/* `encode_icm_fully!` (encode_icm.jl:4-127): perturbation (:55-70), then `niter` sweeps over the node order (:72-125),
 * IN PLACE on B and WITHOUT the accept test (that is encoding_icm's, :178-186).  Every node update goes through
 * node_update() above -- the function all the other entry points of this file use.  With npert = 0 and randord = 0 the
 * call draws no random number: that is the deterministic call tools/make_reference_fixture.jl records from the reference
 * itself, and tests/test_reference_fixture.py replays it through THIS function to pin the oracle's encoder.
 * margins (optional, n floats): per vector, the smallest (second-best - best) conditioned value met in any of its node
 * updates (+inf when h == 1) -- lets a checker tell decisions that survive a different GEMM summation order from near-ties. */
ORC_EXPORT int orc_encode_icm_fully(const float *X, int16_t *B, const float *K, int d, long n, int m, int h, int niter,
                                    int randord, int npert, uint64_t seed, uint32_t it, uint64_t global_offset, float *margins) {
    if (d < 1 || n < 0 || m < 1 || m > 16 || h < 1 || h > 256 || niter < 0 || npert < 0) return -1;
    for (long q = 0; q < n * m; ++q) if (B[q] < 1 || B[q] > h) return -3;
    float *Kt = make_Kt(K, m, h, d);
    float *sci = (float *)malloc(sizeof(float) * (size_t)m * h);
    float *T = (float *)malloc(sizeof(float) * (size_t)m * m * h * h);
    orc_sqnorms(K, m * h, d, sci);
    orc_tables(K, m, h, d, T);
    int32_t order[16];
    orc_perm(seed, it, m, randord, order);
#pragma omp parallel
    {
        float *u = (float *)malloc(sizeof(float) * (size_t)m * h);
        float *s = (float *)malloc(sizeof(float) * (size_t)h);
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i) {
            uint8_t nw[16];
            for (int j = 0; j < m; ++j) nw[j] = (uint8_t)(B[i * m + j] - 1);
            unaries_one(X + (size_t)i * d, Kt, sci, d, m, h, u);
            orc_perturb(seed, global_offset + (uint64_t)i, it, m, h, npert, nw);
            float margin = INFINITY;
            for (int sw = 0; sw < niter; ++sw)
                for (int q = 0; q < m; ++q) {
                    const int j = order[q];
                    const int best = node_update(u + (size_t)j * h, T, nw, j, m, h, s);
                    nw[j] = (uint8_t)best;
                    for (int a = 0; a < h; ++a)
                        if (a != best && !(s[a] - s[best] >= margin)) margin = s[a] - s[best];
                }
            for (int j = 0; j < m; ++j) B[i * m + j] = (int16_t)(nw[j] + 1);
            if (margins) margins[i] = margin;
        }
        free(u); free(s);
    }
    free(T); free(sci); free(Kt);
    return 0;
}

/* --------------------------------- one ILS iteration, structure-faithful ---- */

/* `encoding_icm` (encode_icm.jl:131-189) with the reference's own loop nest kept:
 * per worker shard (splitarray, utils.jl:152-177; one OpenMP thread stands in for
 * one `julia -p` worker process) the unaries of the whole shard are materialised
 * (get_unaries, utils.jl:94-122) -- recomputed on every call exactly as the
 * reference does -- and each node update is a whole-array sweep over `ub`
 * (encode_icm.jl:72-125: copy, m-1 read-modify-write passes, argmin pass).
 * Bit-identical results to orc_encode_icm; this is the honest CPU baseline that
 * bench.py times ("port").  `it` = 0-based ILS iteration index (RNG counter). */
ORC_EXPORT int orc_encoding_icm_faithful(const float *X, const int16_t *oldB, const float *K, int d, long n, int m, int h,
                                         int niter, int randord, int npert, uint64_t seed, uint32_t it,
                                         uint64_t global_offset, int nworkers, int16_t *outB) {
    if (d < 1 || n < 0 || m < 1 || m > 16 || h < 1 || h > 256 || nworkers < 1) return -1;
    /* get_binaries (every call, as the reference does) */
    float *T = (float *)malloc(sizeof(float) * (size_t)m * m * h * h);
    orc_tables(K, m, h, d, T);
    float *Kt = make_Kt(K, m, h, d);
    float *sci = (float *)malloc(sizeof(float) * (size_t)m * h);
    orc_sqnorms(K, m * h, d, sci);
    int32_t order[16];
    orc_perm(seed, it, m, randord, order);

    uint8_t *cur = (uint8_t *)malloc((size_t)n * m);
    uint8_t *nw = (uint8_t *)malloc((size_t)n * m);
    for (long q = 0; q < n * m; ++q) { cur[q] = (uint8_t)(oldB[q] - 1); nw[q] = cur[q]; }
    float *prevcost = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float *newcost = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    orc_veccost(X, K, cur, n, d, m, h, prevcost);                       /* :149 */

    long perpart = n / nworkers, xtra = n % nworkers;                    /* splitarray */
#pragma omp parallel for schedule(static, 1) num_threads(nworkers)
    for (int w = 0; w < nworkers; ++w) {
        long lo = w < xtra ? w * (perpart + 1) : xtra * (perpart + 1) + (w - xtra) * perpart;
        long nw_n = w < xtra ? perpart + 1 : perpart;
        if (nw_n == 0) continue;
        /* get_unaries for the shard: m tables of h x n_w */
        float *U = (float *)malloc(sizeof(float) * (size_t)m * h * nw_n);
        float *u = (float *)malloc(sizeof(float) * (size_t)m * h);
        for (long l = 0; l < nw_n; ++l) {
            unaries_one(X + (size_t)(lo + l) * d, Kt, sci, d, m, h, u);
            for (int j = 0; j < m; ++j) memcpy(U + ((size_t)j * nw_n + l) * h, u + (size_t)j * h, sizeof(float) * h);
        }
        free(u);
        float *ub = (float *)malloc(sizeof(float) * (size_t)h * nw_n);
        for (long l = 0; l < nw_n; ++l)                                   /* :55-70 */
            orc_perturb(seed, global_offset + (uint64_t)(lo + l), it, m, h, npert, nw + (size_t)(lo + l) * m);
        for (int sw = 0; sw < niter; ++sw)                                /* :72 */
            for (int q = 0; q < m; ++q) {
                int j = order[q];
                memcpy(ub, U + (size_t)j * nw_n * h, sizeof(float) * (size_t)h * nw_n);   /* :78-81 */
                for (int k = 0; k < m; ++k) {                             /* :84 */
                    if (k == j) continue;
                    const float *bb = T + ((size_t)j * m + k) * h * h;
                    for (long l = 0; l < nw_n; ++l) {                     /* :96-101 */
                        const float *col = bb + (size_t)nw[(size_t)(lo + l) * m + k] * h;
                        float *ul = ub + (size_t)l * h;
                        for (int a = 0; a < h; ++a) ul[a] = ul[a] + col[a];
                    }
                }
                for (long l = 0; l < nw_n; ++l) {                         /* :105-119 */
                    const float *ul = ub + (size_t)l * h;
                    float minv = ul[0]; int mini = 0;
                    for (int a = 1; a < h; ++a) if (ul[a] < minv) { minv = ul[a]; mini = a; }
                    nw[(size_t)(lo + l) * m + j] = (uint8_t)mini;
                }
            }
        free(ub); free(U);
    }
    orc_veccost(X, K, nw, n, d, m, h, newcost);                         /* :178 */
    for (long i = 0; i < n; ++i) {                                        /* :183-186 */
        const uint8_t *src = (newcost[i] < prevcost[i]) ? nw + (size_t)i * m : cur + (size_t)i * m;
        for (int j = 0; j < m; ++j) outB[i * m + j] = (int16_t)(src[j] + 1);
    }
    free(prevcost); free(newcost); free(cur); free(nw); free(sci); free(Kt); free(T);
    return 0;
}

/* qerror -- utils.jl:257-285 (mean squared reconstruction error; f64 accumulation). */
ORC_EXPORT double orc_qerror(const float *X, const int16_t *B, const float *K, int d, long n, int m, int h) {
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (long i = 0; i < n; ++i) {
        uint8_t c[16];
        for (int j = 0; j < m; ++j) c[j] = (uint8_t)(B[i * m + j] - 1);
        acc += (double)cost_one(X + (size_t)i * d, K, c, d, m, h);
    }
    return n > 0 ? acc / (double)n : 0.0;
}

ORC_EXPORT int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
