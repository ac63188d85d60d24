/*
 * lsq_mi355x.h -- C-ABI of liblsq_mi355x.so: the MI355X (gfx950) LSQ encoding engine.
 *
 * Drop-in boundary for the ILS/ICM encoding hot path of
 * una-dinosauria/local-search-quantization.  Every entry point cites the reference
 * interface it replaces (paths relative to the reference repository root).
 *
 * Conventions (the reference's own, src/linscan/Linscan.jl:63-69 style):
 *   - extern "C", plain pointers + integer sizes, no torch/STL types;
 *   - HOST buffers are the Julia column-major arrays read in place:
 *       RX/X  d x n  Float32          -> x_i[t]        at  X[i*d + t]
 *       K     d x (m*h) Float32 = hcat(C...) (encode_icm_cuda.jl:80, Linscan.jl:68)
 *                                      -> c_{j,a}[t]    at  K[(j*h + a)*d + t]
 *       B     m x n  Int16, 1-BASED   -> code (i,j)    at  B[i*m + j]
 *   - DEVICE buffers (the *_dev entry points) use the same X / K layouts and
 *     uint8 0-BASED codes [n][m] (the layout the reference hands to search,
 *     demos/demo_lsq_gpu.jl:67);
 *   - the caller allocates every output (encode_icm_cuda.jl:40-41,275-279) and lends
 *     pointers for the duration of the call; nothing is retained;
 *   - every function returns 0 on success, a negative LSQ_E* code otherwise;
 *     lsq_last_error() returns a thread-local message (the reference has no error
 *     convention at all: encode_icm_cuda.jl:264 "TODO check that splits >= 1");
 *   - calls are blocking unless stated; one host thread per lsq_ctx.
 *
 * Supported shapes: h == 256 (the reference GPU path hard-codes it:
 * src/encodings/cuda/cudautils.cu:38,58,155,245), 1 <= m <= 16 (cudautils.cu:38),
 * any d >= 1, any n >= 0.
 *
 * RNG (build-defined; the reference seeds curand with clock(), cudautils.cu:21):
 * Philox4x32-10 keyed by `seed`, counter = (global vector index, ILS iteration,
 * domain|block).  `global_offset` is the global index of the first vector of the
 * buffer, so results do not depend on nsplits / #GPUs / chunking.
 */
#ifndef LSQ_MI355X_H
#define LSQ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSQ_VERSION 600

#if defined(__GNUC__)
#define LSQ_API __attribute__((visibility("default")))
#else
#define LSQ_API
#endif

enum {
    LSQ_OK = 0,
    LSQ_EINVAL = -1,    /* bad argument (shape, range, null pointer)      */
    LSQ_EHIP = -2,      /* a HIP runtime call failed (message has details) */
    LSQ_ENOMEM = -3,    /* device or host allocation failed                */
    LSQ_ECODE = -4,     /* an input code is outside 1..h                   */
    LSQ_ENODEV = -5     /* no usable gfx950 device                         */
};

typedef struct lsq_ctx lsq_ctx;

/* Accumulated device time per kernel class since the last lsq_reset_timings (hipEvent pairs on
 * the context's stream), only collected while option "profile" is 1. */
typedef struct lsq_timings {
    double tables_ms;        /* sqnorms + pairwise tables (get_binaries)            */
    double unaries_ms;       /* unary build (get_unaries)                            */
    double perturb_ms;       /* perturbation kernels                                 */
    double icm_ms;           /* ICM node-update kernels (the dominant kernel)        */
    double cost_ms;          /* cost + accept (+ objective) kernels                  */
    double other_ms;         /* layout conversion, snapshots                         */
    int64_t icm_launches;    /* number of ICM kernel launches inside icm_ms           */
    int64_t icm_node_updates;/* vector x node updates executed inside icm_ms          */
    /* which path the ICM blocks took (counted per block and node update, on the device; since v200): */
    int64_t staged_blocks;   /* table slices staged through LDS, the block walked all of them (team size 1)  */
    int64_t light_blocks;    /* few active vectors: table columns gathered from L2, one wave per vector       */
    int64_t filtered_blocks; /* 16-bit filtered walk (exact refinement of ambiguous vectors), LDS-staged slices       */
    int64_t filter_refined;  /* node updates the 16-bit filter could not decide: every candidate within the window evaluated exactly */
    int64_t filter_exact;    /* ... number of exact f32 candidate evaluations that took                               */
    int64_t filter_f32;      /* node updates sent to the f32 path because a unary fell outside the sampled level range  */
    int64_t filter_fallback_chunks; /* resident chunks the filter handed to the f32 walk: non-finite / degenerate value ranges, more than 1/64 of the
                                * (vector, node) pairs outside the sampled level range, or a first ILS iteration in which the filter decided too
                                * little (since v300)                                                                   */
    int64_t xs_launches;     /* reserved, always 0 (counted launches of round 4's schedule 7, which left the tree in v600; kept for the layout) */
    int64_t xs_fallback_launches; /* reserved, always 0 (as above)                                                      */
    int64_t table_reuses;    /* host-buffer calls that found their codebooks unchanged since the previous one: no upload of K, no table rebuild (since v500) */
} lsq_timings;      /* fields are only ever APPENDED: a caller built against an older header passes its own sizeof to lsq_get_timings_sized */

LSQ_API const char *lsq_last_error(void);
LSQ_API int lsq_version(void);
LSQ_API int lsq_device_count(int *count);

/* Context = device + stream + workspace.  Replaces the per-call CuContext / module load /
 * destroy! of encode_icm_cuda.jl:59-64,226-228. */
LSQ_API int lsq_create(lsq_ctx **ctx, int device);
LSQ_API int lsq_destroy(lsq_ctx *ctx);
/* Launch on the caller's hipStream_t (e.g. torch's current stream).  NULL = HIP's default (null)
 * stream; option "own_stream" switches back to the context's private non-blocking stream. */
LSQ_API int lsq_set_stream(lsq_ctx *ctx, void *hip_stream);
/* Options: "chunk" (vectors per resident chunk, default 1015808 = 256 blocks x 3968 vectors: one pass of the walk kernel per block), "profile" (0/1), "own_stream",
 *   "schedule" -- how the ICM node updates run; all give bit-identical codes:
 *        6 (default) 16-bit FILTERED walk, one launch per ILS iteration: every term of a conditioned sum is also held as a 16-bit level
 *                    on a common step (u16 unary planes written by the unary GEMM's epilogue, u16 pair-table slices staged in LDS); a
 *                    1024-thread block owns <= 4096 vectors, sums the levels with packed 16-bit adds, tracks the two smallest keys and
 *                    decides a node update on the levels alone when the runner-up lies outside a rigorous error window; otherwise every
 *                    candidate inside the window is evaluated exactly in f32.  Half the HBM / LDS bytes of schedule 4.  Chunks below
 *                    "q16_min" vectors, and chunks whose data are not finite (decided on the device), run as schedule 4.
 *        4           f32 walk, one launch per ILS iteration: the block runs the icmiter x m node updates back to back, walking all LDS-staged
 *                    f32 table slices for each; f32 unaries streamed slice-major from HBM;
 *        3           the f32 walk, one launch per node update;
 *   "q16_min" (default 65536): schedule 6 applies to chunks with at least this many vectors (below, every block is "light": nothing to filter);
 *   "per_node" (0/1, default 0): schedule 6 with one launch per node update (profiling: per-sweep timings and counters);
 *   "light" (default 160 in the filtered walk, 256 in the f32 walk: the measured crossovers): a block with at most this many active vectors
 *        gathers f32 table columns straight from L2 (one wave per vector, two vectors in flight) instead of staging slices through LDS;
 *        0 = always stage.  Same codes.
 *   "wave_max" (default 64): a chunk with at most this many vectors per block of the walk kernel (n <= 256 x 64) -- all of them light -- runs
 *        icm_wave_kernel instead: a wave owns its two vectors through every node update of the launch (records and validity words in registers,
 *        no compaction, no barriers); 0 = never.  Same codes.
 *   "fallback" (0/1, default 1): a candidate whose codes become equal to the vector's current codes inherits the current state's validity
 *        bits (validity depends on the code tuple only): exact, ~12 % fewer node updates.
 *   "skip" (0/1, default 1): a node whose conditioning codes did not change since it was last minimised is not recomputed (exact memoisation --
 *        same codes, fewer bytes).
 *   "filter_fallback_div" (default 64): schedule 6 hands a resident chunk to the f32 walk when more than 1 / div of its (vector, node) pairs
 *        have a unary outside the sampled 16-bit level range (each such pair takes the one-wave-per-vector f32 routine, ~5x the cost of a filtered update);
 *        0 = never.  Same codes.
 *   "filter_probe_div" (default 8): after the FIRST ILS iteration of a resident chunk schedule 6 reads that iteration's counters and runs the
 *        remaining iterations as schedule 4 when more than 1 / div of the recomputed node updates needed the exact refinement or the f32 routine
 *        (a level step blown up by a few extreme values: scale-mixture / heavy-tailed data); 0 = never.  Same codes.  A call of ONE ILS iteration
 *        (lsq_encoding_icm chained by a trainer) splits its launch after the first SWEEP and probes that; nothing is remembered between calls.
 *   "async" (0/1, default 0): lsq_encode_icm_dev without any host synchronisation (see there).  Which entry points block: every entry point that
 *        takes or returns HOST buffers waits for its results; lsq_encode_icm_dev waits (per chunk: verdict, probe; at the end: sums) unless "async" = 1;
 *        lsq_linscan_dev, lsq_quantize_norms_dev and lsq_update_codebooks_dev read small control words back (threshold lists, convergence counter) and wait.
 *   "ils_counter": the next iteration index used by lsq_encoding_icm / lsq_encode_icm_fully when called with it = LSQ_IT_AUTO
 *        (starts at 0, advances by one per such call).
 *   (liblsq_mi355x_tuning.so only) "ablation": timing-only kernel variants whose results are garbage. */
LSQ_API int lsq_set_option(lsq_ctx *ctx, const char *key, int64_t value);
LSQ_API int lsq_get_timings(lsq_ctx *ctx, lsq_timings *out);      /* writes the v400 layout only (everything before table_reuses: a caller built against any header since v400 is never overrun); the fields appended since come through: */
/* ... the size-checked form: at most `bytes` bytes of the structure are written (the fields a caller compiled against an older header knows about);
 * compare lsq_version() with LSQ_VERSION at load time to learn which fields the library fills. */
LSQ_API int lsq_get_timings_sized(lsq_ctx *ctx, void *out, size_t bytes);
/* Node updates actually recomputed (not memoised) per POSITION in an ILS iteration's node sequence, position = sweep * m + rank in
 * the visiting order (mod 64), summed over ILS iterations, chunks and calls since the last lsq_reset_timings: the device-side
 * counterpart of the reference's per-iteration "% equal / % better" prints, for the sweeps.  out[count], count <= 64. */
LSQ_API int lsq_get_walk_trace(lsq_ctx *ctx, int64_t *out, int count);
LSQ_API int lsq_reset_timings(lsq_ctx *ctx);
LSQ_API int lsq_synchronize(lsq_ctx *ctx);

/* ---- single-process multi-GPU (the `ngpus` of SURVEY 8(b); what a Julia master process calls on a multi-GPU node) --------
 * One context and one host thread per listed device; the n vectors are split with `splitarray` (src/utils.jl:152-177,
 * the reference's own sharding of encode_icm.jl:165-173) and every shard is encoded with its global offset, so the result
 * is bit-identical to the one-device call for any device list (RNG keyed by the global vector index).  No data-path
 * collective: K is uploaded to every device, objective sums and counters are added on the host.  `devices` may repeat an
 * ordinal (two shards time-share one GPU).  Options apply to every context.  One process per GPU with torch.distributed /
 * RCCL (local-search-quantization_amd/distributed.py) is the other, multi-process way to use several GPUs. */
typedef struct lsq_multi lsq_multi;
LSQ_API int lsq_multi_create(lsq_multi **out, const int *devices, int ndev);
LSQ_API int lsq_multi_destroy(lsq_multi *mg);
LSQ_API int lsq_multi_set_option(lsq_multi *mg, const char *key, int64_t value);
/* Same arguments, layouts and outputs as lsq_encode_icm (below), minus nsplits. */
LSQ_API int lsq_multi_encode_icm(lsq_multi *mg, const float *RX, const int16_t *B, const float *K, int d, int64_t n, int m, int h,
                                 const int64_t *ilsiters, int nr, int icmiter, int npert, int randord, uint64_t seed,
                                 uint64_t global_offset, int verbose, int16_t *Bs, float *objs);

/* ---- (1) the whole call ------------------------------------------------------------------
 * Replaces encode_icm_cuda(RX, B, C, ilsiters, icmiter, npert, randord, nsplits, V)
 *   -> (Bs, objs)      src/encodings/encode_icm_cuda.jl:253-296 (and _single, :22-234);
 * call site demos/demo_lsq_gpu.jl:50.  Runs max(ilsiters) ILS iterations; snapshot k holds the
 * codes and the objective (qerror, src/utils.jl:257-285) after ilsiters[k] iterations.
 * Semantics of each ILS iteration follow the reference CPU path (src/encodings/encode_icm.jl:
 * 131-189): perturb, icmiter sweeps, accept iff strictly better.
 *   Bs   : nr x (m x n Int16, 1-based), caller-allocated;   objs : nr Float32.
 * nsplits is accepted for signature compatibility (the reference needs it for 12 GB GPUs,
 * demos/demo_lsq_gpu.jl:49); results do not depend on it. */
LSQ_API int lsq_encode_icm(lsq_ctx *ctx, const float *RX, const int16_t *B, const float *K,
                   int d, int64_t n, int m, int h,
                   const int64_t *ilsiters, int nr, int icmiter, int npert, int randord,
                   int nsplits, uint64_t seed, uint64_t global_offset, int verbose,
                   int16_t *Bs, float *objs);

/* Same call on DEVICE-resident buffers: launches go to the context's stream.  BLOCKING by default: the host waits once per resident chunk for the
 * chunk's three-word verdict after the unary GEMM, once more for the probe after the first ILS iteration, and at the end for the read-back of the
 * objective sums and counters.  Option "async" = 1 removes every one of those waits: the verdict and the probe are taken by one-thread kernels, BOTH walk
 * kernels are enqueued each ILS iteration (the one the device word does not name returns at once: ~3 us per launch), and obj_sums / stats are written in
 * stream order -- they must then be DEVICE pointers or PINNED host memory, and are valid once the stream has been synchronised by the caller.  With "async"
 * the call contains no host synchronisation and no allocation once the context's work buffers have the size of the shape (i.e. from the second call of
 * that shape on): it can be captured into a hipGraph.  Walk statistics of async calls are folded into lsq_get_timings at its next call (which waits).
 * dB0 / dBs: uint8 0-based [n][m] (dBs: nr of them).
 * obj_sums (host, nr doubles) receives SUM_i cost_i (not the mean) so that a multi-GPU caller
 * can add shards; objs = obj_sums / n_total.  stats (host, optional, 2*max(ilsiters) int64):
 * per ILS iteration the number of vectors whose new cost was == / < the previous one
 * (the two counters the reference prints, encode_icm.jl:180-184). */
LSQ_API int lsq_encode_icm_dev(lsq_ctx *ctx, const float *dX, const uint8_t *dB0, const float *dK,
                       int d, int64_t n, int m, int h,
                       const int64_t *ilsiters, int nr, int icmiter, int npert, int randord,
                       uint64_t seed, uint64_t global_offset,
                       uint8_t *dBs, double *obj_sums, int64_t *stats);

/* ---- (2) the CPU-path shaped entry points -------------------------------------------------
 * encoding_icm(X, oldB, C, niter, randord, npert, V) -> B     src/encodings/encode_icm.jl:131-189
 * ONE ILS iteration with the accept rule.  `it` = the iteration's 0-based index: it keys the perturbation and the node
 * order (the reference draws them from Julia's global RNG, so every call differs).  The reference's callers keep no
 * such count (demos/demo_lsq.jl:48-51: `for i = 1:ilsiter; B = encoding_icm(...); end`), so an unchanged caller passes
 * it = LSQ_IT_AUTO: the CONTEXT then counts -- the k-th such call on a context uses it = k-1 (option "ils_counter" sets
 * the next value; lsq_encode_icm_fully shares the counter) -- and `ilsiter` chained calls give exactly the codes of
 * lsq_encode_icm(ilsiters = [ilsiter]) on a fresh context.  A fixed `it` on every call would re-draw the SAME
 * perturbation each time and the ILS would silently stop exploring. */
#define LSQ_IT_AUTO 0xFFFFFFFFu
LSQ_API int lsq_encoding_icm(lsq_ctx *ctx, const float *X, const int16_t *oldB, const float *K,
                     int d, int64_t n, int m, int h, int niter, int randord, int npert,
                     uint64_t seed, uint32_t it, uint64_t global_offset, int16_t *outB);

/* encode_icm_fully!(B, X, C, binaries, cbi, niter, randord, npert, IDX, V)
 *   src/encodings/encode_icm.jl:4-127 -- the worker: perturb + niter sweeps, in place, NO accept
 * test.  This is the hook the authors left commented at encode_icm.jl:163 (`encode_icm_cpp!`).
 * `binaries`/`cbi` are rebuilt on the device from K (cheaper than shipping them);
 * idx_first = first(IDX), 1-based global column of X[:,1] (keys the RNG). */
LSQ_API int lsq_encode_icm_fully(lsq_ctx *ctx, int16_t *B, const float *X, const float *K,
                         int d, int64_t n, int m, int h, int niter, int randord, int npert,
                         int64_t idx_first, uint64_t seed, uint32_t it);

/* ---- (3) the numeric helpers the path is made of (host buffers) ---------------------------
 * get_unaries(X, C)   src/utils.jl:94-122  -> U [m][n][h]  (unaries[j] is h x n column-major) */
LSQ_API int lsq_get_unaries(lsq_ctx *ctx, const float *X, const float *K, int d, int64_t n, int m, int h, float *U);
/* get_binaries(C)     src/utils.jl:125-144 (+ the transposes of encode_icm.jl:25-28)
 *   -> T [m][m][h][h], T[j][k][b][a] = 2<c_{j,a}, c_{k,b}>; binaries[idx(i<j)][a + b*h] = T[i][j][b][a] */
LSQ_API int lsq_get_binaries(lsq_ctx *ctx, const float *K, int d, int m, int h, float *T);
/* veccost(X, B, C)    src/utils.jl:225-254  -> cost [n] */
LSQ_API int lsq_veccost(lsq_ctx *ctx, const float *X, const int16_t *B, const float *K,
                int d, int64_t n, int m, int h, float *cost);
/* qerror(X, B, C)     src/utils.jl:257-285  -> mean squared error (f64 accumulation) */
LSQ_API int lsq_qerror(lsq_ctx *ctx, const float *X, const int16_t *B, const float *K,
               int d, int64_t n, int m, int h, double *out);
/* perturb kernel      src/encodings/cuda/cudautils.cu:27-80 (host buffers, in place) */
LSQ_API int lsq_perturb(lsq_ctx *ctx, int16_t *B, int64_t n, int m, int h, int npert,
                uint64_t seed, uint32_t it, uint64_t global_offset);
/* randinit(n, m, h)   src/initializations.jl:2-8   (host function, Philox-keyed) */
LSQ_API int lsq_randinit(uint64_t seed, uint64_t global_offset, int64_t n, int m, int h, int16_t *B);
/* node visiting order of ILS iteration `it` (randperm of encode_icm.jl:46-49), 0-based */
LSQ_API int lsq_node_order(uint64_t seed, uint32_t it, int m, int randord, int32_t *order);
/* splitarray(1:n, nparts)  src/utils.jl:152-177 -> part's [start, start+len) , 0-based */
LSQ_API int lsq_splitarray(int64_t n, int nparts, int part, int64_t *start, int64_t *len);

/* ---- (3b) search side of the path (host code) --------------------------------
 * linscan_aqd_query_extra_byte(dists, idx, codes, queries, codebooks, dbnorms, nqueries, ncodes, m, h, d, nn)
 *   src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:97-104, ccall at src/linscan/Linscan.jl:63-69.
 * Same argument list plus `nthreads` (0 = all cores).  codes: m x n uint8 0-based; queries: d x nq;
 * codebooks = hcat(C...); dbnorms: n.  Outputs (caller-allocated): dists nn x nq f32 ascending,
 * idx nn x nq int32, 1-BASED.  Distances are bit-identical to the reference build (same f32
 * operation order); ties are ordered by id, as the reference's pair sort does.  Requires nn <= n. */
LSQ_API int lsq_linscan_aqd_query_extra_byte(float *dists, int *idx, const unsigned char *codes, const float *queries,
                                     const float *codebooks, const float *dbnorms, int nqueries, int ncodes,
                                     int m, int h, int d, int nn, int nthreads);

/* The same scan ON THE DEVICE (SURVEY 8(f)-1 "HIP scan later"; csrc/lsq_adc.hip).  Same argument list behind a context; same results bit for
 * bit -- distances, 1-based ids, order including ties -- as the host function above and as the reference build.  Requires h == 256,
 * 1 <= m <= 16, 1 <= nn <= ncodes.  NaN distances (undefined order in the reference's partial_sort) sort last.
 *   lsq_linscan      host buffers (uploaded, searched, downloaded);
 *   lsq_linscan_dev  device buffers: codes [ncodes][m] uint8 0-based, queries [nq][d], codebooks [m*h][d], dbnorms [ncodes]; outputs [nq][nn]. */
LSQ_API int lsq_linscan(lsq_ctx *ctx, float *dists, int *idx, const unsigned char *codes, const float *queries, const float *codebooks,
                        const float *dbnorms, int nqueries, int ncodes, int m, int h, int d, int nn);
LSQ_API int lsq_linscan_dev(lsq_ctx *ctx, float *d_dists, int *d_idx, const uint8_t *d_codes, const float *d_queries, const float *d_codebooks,
                            const float *d_dbnorms, int nqueries, int ncodes, int m, int h, int d, int nn);
/* The same over a database sharded across the devices of an lsq_multi (splitarray shards, one host thread per device, host merge of the per-device
 * lists by (distance, id)): the result of ONE scan of the whole database, ties across shards included. */
LSQ_API int lsq_multi_linscan(lsq_multi *mg, float *dists, int *idx, const unsigned char *codes, const float *queries, const float *codebooks,
                              const float *dbnorms, int nqueries, int ncodes, int m, int h, int d, int nn);
/* What the device scan did since the last lsq_reset_timings (times only with option "profile" = 1). */
typedef struct lsq_linscan_stats {
    int64_t queries, codes;          /* queries searched (accumulated); database size of the last call */
    int64_t candidates;              /* (dist, id) pairs written to memory: the lists the selection sorted */
    int64_t fallback_queries;        /* queries redone by the exhaustive road (candidate list short or overflowing) */
    int64_t batches;
    int64_t exhaustive;              /* last call: 1 = every distance written and sorted (small database / large nn / option) */
    int64_t threshold_rank, list_capacity;      /* last call: sample rank of the threshold, entries per candidate list */
    double lut_ms, sample_ms, scan_ms, select_ms;
} lsq_linscan_stats;
LSQ_API int lsq_get_linscan_stats(lsq_ctx *ctx, lsq_linscan_stats *out);

/* quantize_norms(B, C, cbnorms) -> dbnormsB      src/utils.jl:6-31: per database vector the squared norm of its reconstruction
 * (f32; codebooks, then dimensions, ascending) and the 1-based index of the nearest of the `ncb` (<= 256) scalar centroids, first minimum of
 * (norm - cbnorms[j])^2 like findmin.  `dbnorms` (optional) receives cbnorms[index] -- what the scan consumes (demos/demo_lsq_gpu.jl:57-60);
 * `norms` (optional) the unquantised norms (the input of the reference's norm k-means, src/lsq/LSQ.jl:80-84).  PARITY UNPINNED: the reference's
 * norm loop is `@simd`; this is the sequential order.  The _dev variant takes device pointers and uint8 0-BASED codes [n][m], index output 0-based. */
LSQ_API int lsq_quantize_norms(lsq_ctx *ctx, const int16_t *B, const float *K, const float *cbnorms, int ncb, int d, int64_t n, int m, int h,
                               int16_t *idx_out, float *dbnorms, float *norms);
LSQ_API int lsq_quantize_norms_dev(lsq_ctx *ctx, const uint8_t *d_codes, const float *d_K, const float *d_cbnorms, int ncb, int d, int64_t n,
                                   int m, int h, uint8_t *d_idx_out, float *d_dbnorms, float *d_norms);

/* update_codebooks(X, B, h) -> C      src/codebook_update.jl:52-86 (host code; north_star keeps it on the host).
 * K[t, :] = lsqr(sparsify_codes(B, h), X[t, :]) for every dimension t (LSQR of Paige & Saunders, Float32,
 * x0 = 0, atol = btol = sqrt(eps(Float32)), conlim = 1e8 -- IterativeSolvers' defaults; PARITY UNPINNED, the
 * reference neither vendors nor pins IterativeSolvers).  X d x n, B m x n Int16 1-based; K_out d x (m*h)
 * = hcat(C...) caller-allocated.  nthreads 0 = all cores (dimensions are independent). */
LSQ_API int lsq_update_codebooks(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K_out);
/* The reference's other solver, codebook_upd_method = "lsmr" (src/codebook_update.jl:18-21 -> IterativeSolvers.lsmr): LSMR of Fong & Saunders with the same
 * operator, tolerances and threading as lsq_update_codebooks (host code; since v500). */
LSQ_API int lsq_update_codebooks_lsmr(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K_out);

/* The same update ON THE DEVICE (csrc/lsq_lsqr.hip): all d systems advanced together, the same LSQR restatement (Float32 recurrences, the long sums
 * in double, IterativeSolvers' default stopping rules), the rows sorted by code once per call so that S'u is added in the host's order without atomics.
 * The same bits on every call; the same bits as lsq_update_codebooks on every tested problem (its norms are added in another order: agreement is
 * required to 1e-5 only).  _gpu: host buffers, Julia layout as above (X d x n, B m x n Int16 1-based, K_out d x (m*h));  _dev: device buffers,
 * codes [n][m] uint8 0-BASED.  h must be 256.  iterations (optional): LSQR iterations LAUNCHED = the slowest dimension's count rounded up to the host's
 * next look at the convergence counter (every 4 iterations for the first 8, every 2 after: up to 3 more than needed; a converged system is frozen, so the
 * extra launches change nothing but this number); `maxiter` bounds launched iterations likewise. */
LSQ_API int lsq_update_codebooks_gpu(lsq_ctx *ctx, const float *X, const int16_t *B, int d, int64_t n, int m, int h, float *K_out, int *iterations);
LSQ_API int lsq_update_codebooks_dev(lsq_ctx *ctx, const float *d_X, const uint8_t *d_codes, int d, int64_t n, int m, int h, float *d_K_out,
                                     int *iterations);

/* ---- the initialisers' two data-parallel steps ON THE DEVICE (csrc/lsq_init.hip; since v600) -----------------------------------
 * encoding_viterbi(X, C) -> B      src/encodings/encode_chain.jl:92-123 (worker encode_viterbi! :2-89): the exact MAP codes of a CHAIN -- unaries
 * (get_unaries, utils.jl:94-122) plus the binaries 2 C_i' C_{i+1} of consecutive codebooks only (:103-106) -- by dynamic programming: m - 1 min-plus
 * steps over a 256 x 256 table per vector, first minimum on ties (the reference's strict-'<' scans, :58-66,74), then the trace (:77-83).  K = hcat(C...)
 * as everywhere (chain codebooks are zero outside the dimensions they cover: codebook_update.jl:88-102); 2 <= m <= 16, h must be 256.
 * Host form: X d x n, B m x n Int16 1-BASED (the reference's return value).  _dev: device pointers, codes [n][m] uint8 0-BASED. */
LSQ_API int lsq_encode_viterbi(lsq_ctx *ctx, const float *X, const float *K, int d, int64_t n, int m, int h, int16_t *B);
LSQ_API int lsq_encode_viterbi_dev(lsq_ctx *ctx, const float *d_X, const float *d_K, int d, int64_t n, int m, int h, uint8_t *d_B);
/* quantize_pq(X, C) / the assignment step of the k-means behind train_pq and train_opq      src/pq/PQ.jl:12-41, src/opq/kmeans.jl:6-75 (update_assignments!),
 * src/opq/OPQ.jl:60-66,88-91: per codebook j INDEPENDENTLY the first argmin_a of ||c_ja||^2 - 2 <x, c_ja> (= the sub-space distance minus ||x_sub||^2 when
 * codebook j is zero outside its sub-space -- PQ / OPQ codebooks padded to d rows -- and the plain nearest codeword of full-dimensional codebooks otherwise).
 * minval (optional, [n][m] floats, vector-major): the minimum itself; add ||x_sub||^2 for the squared distance the reference's `costs` hold.
 * PARITY UNPINNED as every initialiser: the reference takes pairwise(SqEuclidean()) from Distances.jl and argmin from its own scan; near-ties may differ.
 * Host form: B m x n Int16 1-based; _dev: codes [n][m] uint8 0-based.  h must be 256. */
LSQ_API int lsq_assign_codewords(lsq_ctx *ctx, const float *X, const float *K, int d, int64_t n, int m, int h, int16_t *B, float *minval);
LSQ_API int lsq_assign_codewords_dev(lsq_ctx *ctx, const float *d_X, const float *d_K, int d, int64_t n, int m, int h, uint8_t *d_B, float *d_minval);

/* ---- (4) device-side generators used by the benchmark harness -----------------------------
 * X[i][t] = float(uniform integer 0..255) (SIFT-like);  codes uniform 0..h-1 (randinit);
 * codebooks: K[j][a][:] = scale * x_{pick(j,a)} for a Philox-picked synthetic vector. */
LSQ_API int lsq_synth_data_u8_dev(lsq_ctx *ctx, uint64_t seed, uint64_t global_offset, int64_t n, int d, float *dX);
LSQ_API int lsq_randinit_dev(lsq_ctx *ctx, uint64_t seed, uint64_t global_offset, int64_t n, int m, int h, uint8_t *dB);
LSQ_API int lsq_synth_codebooks_dev(lsq_ctx *ctx, uint64_t seed, int m, int h, int d, float *dK);

#ifdef __cplusplus
}
#endif
#endif /* LSQ_MI355X_H */
