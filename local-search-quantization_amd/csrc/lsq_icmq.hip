// lsq_icmq.hip -- the 16-bit FILTERED ICM node update (gfx950): same codes as icm_walk_kernel, half the bytes.
//
// A node update needs argmin_a s[a],  s[a] = ((U_j[a] + T_jk1[b_k1][a]) + T_jk2[b_k2][a]) + ...  (plain f32 adds, ascending k,
// lowest index on ties: reference src/encodings/encode_icm.jl:84-119).  icm_walk_kernel evaluates all 256 candidates in f32: 1 KiB
// of unaries from HBM and (m-1) KiB of table rows from LDS per vector and node -- both at their practical limits
// (profiles/r02a_*: 5.7 TB/s of HBM, LDS 52 % busy).  Almost all of that precision is wasted: on SIFT-like data the best and the
// second-best candidate differ by ~6000 while the values span ~10^6.  So:
//
//   FILTER   every term is also stored as a 16-bit level on ONE common step D_j per node (lsq_q16_params): U by the GEMM epilogue
//            (Uq, slice-major u16), the pair tables by tables_to_q16_slices_kernel (Tq).  The levels of a candidate are summed as
//            packed pairs, Q[a] = qU[a] + SUM_k qT_k[a] < 65536 by construction (lsq_q16_node::hiq), so plain 32-bit three-operand
//            adds are exact: 512 B of unaries and (m-1) x 512 B of table rows per vector and node.  A wave tracks the two smallest keys.
//   SHIFTS   a term that is the same for all 256 candidates of a node update cannot change its argmin, so it needs no level range; and a term that
//            depends on the candidate alone may sit in the unary as well as in a table.  The levels are taken of
//                U_j[a] + sigma_ij + SUM_k g_jk[a]      sigma_ij = 2 <x_i, mean codeword of codebook j>  (unary_shift_kernel),
//                T_jk[b][a] - g_jk[a] - min_a(...)      g_jk[a] = mean_b T_jk[b][a]  (table_colmean_kernel, table_range_kernel):
//            the unary against the residual of x after the other codebooks' mean codewords, the tables without their additive row + column
//            structure.  The ranges the common step must cover shrink to 40 % on SIFT-like data (D 11.3 -> 7.4 -> 4.5 at m = 8, 5.9 -> 3.9 -> 2.2 at
//            m = 16), and the ambiguous node updates with them (1.7 % -> 1.2 % -> 0.8 %; m = 16: 3.2 % -> 2.2 % -> 1.4 %).
//   BOUND    |C_i + D Q[a] - s_f32[a]| <= slack := m (0.5 + 2^-5) D + eps_f32  for every candidate, C_i the same for all of them (rounding of
//            each level, of the shifts, of the f32 chain).  Hence the exact argmin a* obeys  Q[a*] <= Qmin + window,  window = floor(2 slack / D) + 1.
//   REFINE   second - best > window: the best key IS the exact argmin (>= 98 % of the node updates).  Otherwise (q16_refine) every
//            candidate whose level sum lies within the window of the best -- the SURVIVORS: they provably contain the exact argmin
//            and all its exact ties -- is evaluated EXACTLY (the f32 unaries the GEMM also wrote, the f32 tables, canonical order)
//            and the lexicographic (value, index) minimum is taken: the reference's strict-< scan.  Vectors with a unary outside
//            the sampled level range are flagged by the GEMM and take the full-f32 routine (one wave per vector).
// Non-finite inputs / degenerate ranges (params.ok = 0) or too many flagged vectors: the HOST reads the chunk's verdict after the unary GEMM
// and gives the chunk to icm_walk_kernel instead (one round trip per resident chunk).  Parity: every test that compares icm_walk_kernel with the oracle also runs this kernel.
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "lsq_q16.h"
#ifdef LSQ_TUNING
__device__ unsigned long long *g_walkq_dbg = nullptr;      // [launch slot][block][16] timestamps (tools only)
__device__ unsigned int g_walkq_dbg_slot = 0;
__device__ unsigned long long *g_walkq_dbg_cur = nullptr;   // block 0's record of the running launch (for q16_refine's stamps)
__device__ unsigned long long *g_walkq_blk = nullptr;       // [block][2] start / end clock of every block of the latest launch (tools only: launch tails)
#define LSQ_WALKQ_DBG_WORDS (24 + 2 * 65)                  // 24 phase stamps of the launch's last node, then start clock / active count of every node + the end clock
#define DBG_STAMP(k) do { if (dbgp && threadIdx.x == 0) dbgp[k] = wall_clock64(); } while (0)
#else
#define DBG_STAMP(k) do {} while (0)
#endif

namespace {

// ---- parameters -----------------------------------------------------------------------------------------------------------------
// Shifts that are the same for every candidate of a node update cannot change its argmin, so they need no level range.  A table row T[j][k][b][:]
// enters a conditioned sum as a whole (the row of the code b that codebook k holds): its minimum is such a shift.  And a shift g_jk[a] that depends on
// the CANDIDATE alone can be moved from the table to the unary: s[a] = (U[a] + SUM_k g_jk[a]) + SUM_k (T_jk[b_k][a] - g_jk[a]).  With g = the column
// means the tables lose their additive row + column structure (T_jk[b][a] = 2 <c_a, c_b>: what is left is the interaction of the two deviations from
// the codebook means) and the unary becomes ||c_a||^2 - 2 <c_a, x - the other codebooks' mean codewords>, against the residual instead of x: both
// ranges shrink, the common step D by another 40 % (7.3 -> 4.5 at cfg2, 3.9 -> 2.2 at m = 16) on top of the row / sigma shifts.
// table_colmean_kernel: g[(j*m + k)*256 + a] = mean_b T[j][k][b][a] (double sum, rounded once).
__global__ __launch_bounds__(256) void table_colmean_kernel(const float *__restrict__ T, int m, float *__restrict__ colmean) {
    const int jk = blockIdx.x, j = jk / m, k = jk % m, a = threadIdx.x;
    if (j == k) { colmean[(int64_t)jk * LSQ_H + a] = 0.0f; return; }
    const float *p = T + (int64_t)jk * LSQ_H * LSQ_H;
    double acc = 0.0;
    for (int b = 0; b < LSQ_H; ++b) acc += (double)p[(int64_t)b * LSQ_H + a];      // coalesced over a
    colmean[(int64_t)jk * LSQ_H + a] = (float)(acc * (1.0 / LSQ_H));
}
// colshift[j*256 + a] = fl32( SUM_{k != j} g_jk[a] ) (double sum of the f32 means: ONE rounding, bounded in q16_params_kernel)
__global__ __launch_bounds__(256) void unary_colshift_kernel(const float *__restrict__ colmean, int m, float *__restrict__ colshift) {
    const int j = blockIdx.x, a = threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < m; ++k)
        if (k != j) acc += (double)colmean[((int64_t)j * m + k) * LSQ_H + a];
    colshift[j * LSQ_H + a] = (float)acc;
}
// Per off-diagonal pair table (65536 floats), on the column-centred entries c = fl(t - g[a]):  rowmin[(j*m + k)*256 + b] = min_a c;
// range[(j*m + k)*3] = max_b (max_a c - min_a c) -- the range the levels need --;  range[.. + 1] = max |t|, range[.. + 2] = max |g| (for the f32 rounding
// terms of the bound).
__global__ __launch_bounds__(256) void table_range_kernel(const float *__restrict__ T, int m, const float *__restrict__ colmean, float *__restrict__ range,
                                                          float *__restrict__ rowmin, int *__restrict__ bad) {
    const int jk = blockIdx.x, j = jk / m, k = jk % m;
    if (j == k) return;
    const float *p = T + (int64_t)jk * LSQ_H * LSQ_H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 g = *reinterpret_cast<const f32x4 *>(colmean + (int64_t)jk * LSQ_H + 4 * lane);
    float rng = 0.0f, mag = 0.0f;
    float gmag = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
    bool nonfinite = false;
    for (int b = wave; b < LSQ_H; b += 4) {                       // one wave per row: 64 lanes x 16 B = the row's 1 KiB in one load
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + (int64_t)b * LSQ_H + 4 * lane);
        nonfinite = nonfinite || !(fabsf(v.x) <= 3.0e38f) || !(fabsf(v.y) <= 3.0e38f) || !(fabsf(v.z) <= 3.0e38f) || !(fabsf(v.w) <= 3.0e38f);
        const f32x4 c = v - g;
        float lo = fminf(fminf(c.x, c.y), fminf(c.z, c.w)), hi = fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w));
        float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); am = fmaxf(am, __shfl_xor(am, off, 64));
        }
        if (lane == 0) rowmin[(int64_t)jk * LSQ_H + b] = lo;
        rng = fmaxf(rng, hi - lo);                                // rounded up below (the params kernel works in double and adds its own margin)
        mag = fmaxf(mag, am);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmag = fmaxf(gmag, __shfl_xor(gmag, off, 64));
    __shared__ float srng[4], smag[4];
    if (__ballot(nonfinite) != 0ull && lane == 0) atomicExch(bad, 1);
    if (lane == 0) { srng[wave] = rng; smag[wave] = mag; }
    __syncthreads();
    if (threadIdx.x == 0) {
        range[jk * 3 + 0] = fmaxf(fmaxf(srng[0], srng[1]), fmaxf(srng[2], srng[3]));
        range[jk * 3 + 1] = fmaxf(fmaxf(smag[0], smag[1]), fmaxf(smag[2], smag[3]));
        range[jk * 3 + 2] = gmag;
    }
}

// The same holds for the unaries: adding sigma_ij to every candidate of (vector i, node j) leaves the argmin alone.  With sigma_ij = 2 <x_i, r_j>,
// r_j = the mean codeword of codebook j, the shifted unary ||c||^2 - 2 <x, c - r_j> no longer carries the component every candidate shares (large on
// SIFT-like non-negative data: it scales with ||x||), and the range the levels must cover shrinks: together with the centred table rows the common
// step D falls by a third (11.3 -> 7.4 at cfg2), and so does the number of ambiguous node updates.  sigma only has to be THE SAME number wherever it
// is used (range pass and level epilogue read it from memory); its own arithmetic is free.
__global__ __launch_bounds__(256) void codebook_means_kernel(const float *__restrict__ K, int m, int d, float *__restrict__ R) {
    // block (j, 64 dimensions): wave w sums codewords w, w + 4, ... (coalesced 256-byte reads), the four partial sums meet in LDS
    const int j = blockIdx.y, t = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float acc = 0.0f;
    if (t < d)
        for (int a = wave; a < LSQ_H; a += 4) acc += K[((int64_t)j * LSQ_H + a) * d + t];
    part[wave][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wave == 0 && t < d) R[(int64_t)j * d + t] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) * (1.0f / LSQ_H);
}

// sigma[i][j] = 2 <x_i, r_j>  (one 16-lane DPP row per vector, grid-stride);  sigmax[0] = bits of max |sigma| over the chunk (non-negative floats
// order like uints).  One atomic per wave at most, and only when the wave's maximum beats the word as L2 holds it (an L1-cached read would stay at
// the initial 0 and send every wave's atomic to the same word: 2.5 ms of serialisation per 10^6 vectors, measured).
// bound (optional; with qflag, P): the level parameters were derived from a SAMPLE's max |sigma| (the host-buffer pipeline: the sample is uploaded ahead
// of the panels); a vector whose |sigma| exceeds the bound the parameters assumed gets every node flagged -- it takes the f32 routine, like a unary
// outside the sampled level range -- so the window stays rigorous for every vector the filter decides.  flag_row0: the row of sigma[0] in qflag.
// LONGV: the instantiation for d > 256 (its m running sums would cost the short-vector path a quarter of its occupancy: 265 -> 365 us at 10^6 x 128)
template <bool LONGV>
__global__ __launch_bounds__(256) void unary_shift_kernel(const float *__restrict__ X, const float *__restrict__ R, int64_t n, int d, int m,
                                                          float *__restrict__ sigma, unsigned *__restrict__ sigmax,
                                                          const unsigned *__restrict__ bound, unsigned short *__restrict__ qflag,
                                                          int64_t flag_row0, lsq_q16_params *__restrict__ P) {
    const int lane = threadIdx.x & 63, lp = lane & 15;
    const int64_t nrows = (int64_t)gridDim.x * 16;
    const bool vec = (d & 3) == 0 && ((((uintptr_t)X | (uintptr_t)R) & 15) == 0);
    float amax = 0.0f;
    // whole waves stay together (the DPP row sums read neighbour lanes): the loop bound is wave-uniform, dead rows are masked
    const int64_t first = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63)) >> 4;      // the wave's first row
    const float bnd = bound ? __uint_as_float(*bound) : __builtin_inff();
    auto flag_row = [&](int64_t i, bool live, float rmax, bool rnan) {
        if (!qflag || !live || lp != 0 || (!(rmax > bnd) && !rnan)) return;
        const int64_t gi = flag_row0 + i;
        const unsigned bits = ((1u << m) - 1u) << (16 * (int)(gi & 1));
        const unsigned old = atomicOr(reinterpret_cast<unsigned *>(qflag + (gi & ~(int64_t)1)), bits);
        const int fresh = __popc(bits & ~old);
        if (P && fresh) atomicAdd(&P->nflag, fresh);
    };
    for (int64_t base = first; base < n; base += nrows) {
        const int64_t i = base + (lane >> 4);
        const bool live = i < n;
        float rmax = 0.0f;
        bool rnan = false;
        const float *x = X + (live ? i : 0) * (int64_t)d;
        if (!LONGV && vec && d <= 256) {                            // the vector stays in registers (16 floats per lane) while the m means pass by (L1)
            f32x4 xr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                xr[t] = 4 * lp + 64 * t < d ? *reinterpret_cast<const f32x4 *>(x + 4 * lp + 64 * t) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            for (int j = 0; j < m; ++j) {
                const float *r = R + (int64_t)j * d + 4 * lp;
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (4 * lp + 64 * t < d) {
                        const f32x4 rv = *reinterpret_cast<const f32x4 *>(r + 64 * t);
                        acc = fmaf(xr[t].w, rv.w, fmaf(xr[t].z, rv.z, fmaf(xr[t].y, rv.y, fmaf(xr[t].x, rv.x, acc))));      // (sigma is a shift, not a result: any rounding serves)
                    }
                acc = acc + dpp_self<DPP_XOR1, 0xf>(acc);
                acc = acc + dpp_self<DPP_XOR2, 0xf>(acc);
                acc = acc + dpp_self<DPP_HALF_MIRROR, 0xf>(acc);
                acc = acc + dpp_self<DPP_MIRROR, 0xf>(acc);
                const float sg = 2.0f * acc;
                if (live && lp == 0) sigma[i * m + j] = sg;
                if (live) { amax = fmaxf(amax, fabsf(sg)); rmax = fmaxf(rmax, fabsf(sg)); if (!(sg == sg)) rnan = true; }
            }
            flag_row(i, live, rmax, rnan);
            continue;
        }
        if (LONGV && vec) {
            // long vectors: x passes through the registers ONCE, in pieces of 256 floats, against all m means (the m sums live side by side).  Before: x was re-read from L1 / L2 for every node (0.52 ms at 125 000 x 960)
            float accj[LSQ_MAX_M];
#pragma unroll
            for (int j = 0; j < LSQ_MAX_M; ++j) accj[j] = 0.0f;
            for (int c0 = 0; c0 < d; c0 += 256) {
                f32x4 xr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    xr[t] = c0 + 4 * lp + 64 * t < d ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x + c0 + 4 * lp + 64 * t)) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < LSQ_MAX_M; ++j) {
                    if (j < m) {
                        const float *r = R + (int64_t)j * d + c0 + 4 * lp;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (c0 + 4 * lp + 64 * t < d) {
                                const f32x4 rv = *reinterpret_cast<const f32x4 *>(r + 64 * t);
                                accj[j] = fmaf(xr[t].w, rv.w, fmaf(xr[t].z, rv.z, fmaf(xr[t].y, rv.y, fmaf(xr[t].x, rv.x, accj[j]))));
                            }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < LSQ_MAX_M; ++j) {
                if (j < m) {
                    float acc = accj[j];
                    acc = acc + dpp_self<DPP_XOR1, 0xf>(acc);
                    acc = acc + dpp_self<DPP_XOR2, 0xf>(acc);
                    acc = acc + dpp_self<DPP_HALF_MIRROR, 0xf>(acc);
                    acc = acc + dpp_self<DPP_MIRROR, 0xf>(acc);
                    const float sg = 2.0f * acc;
                    if (live && lp == 0) sigma[i * m + j] = sg;
                    if (live) { amax = fmaxf(amax, fabsf(sg)); rmax = fmaxf(rmax, fabsf(sg)); if (!(sg == sg)) rnan = true; }
                }
            }
            flag_row(i, live, rmax, rnan);
            continue;
        }
        for (int j = 0; j < m; ++j) {
            const float *r = R + (int64_t)j * d;
            float acc = 0.0f;
            if (vec) {
                for (int t = 4 * lp; t < d; t += 64) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + t), rv = *reinterpret_cast<const f32x4 *>(r + t);
                    acc += xv.x * rv.x + xv.y * rv.y + xv.z * rv.z + xv.w * rv.w;
                }
            } else {
                for (int t = lp; t < d; t += 16) acc += x[t] * r[t];
            }
            acc = acc + dpp_self<DPP_XOR1, 0xf>(acc);
            acc = acc + dpp_self<DPP_XOR2, 0xf>(acc);
            acc = acc + dpp_self<DPP_HALF_MIRROR, 0xf>(acc);
            acc = acc + dpp_self<DPP_MIRROR, 0xf>(acc);
            const float sg = 2.0f * acc;
            if (live && lp == 0) sigma[i * m + j] = sg;
            if (live) { amax = fmaxf(amax, fabsf(sg)); rmax = fmaxf(rmax, fabsf(sg)); if (!(sg == sg)) rnan = true; }      // amax is NaN-ignoring: a non-finite sigma shows up in the range pass
        }
        flag_row(i, live, rmax, rnan);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0 && __float_as_uint(amax) > __hip_atomic_load(sigmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sigmax, __float_as_uint(amax));
}

__global__ __launch_bounds__(64) void q16_range_init_kernel(unsigned *__restrict__ qrange, int m) {
    const int t = threadIdx.x;
    if (t < 2 * LSQ_MAX_M + 4) qrange[t] = (t < 2 * m && !(t & 1)) ? 0xffffffffu : 0u;
}

static void launch_unary_shift(dim3 grid, hipStream_t s, const float *X, const float *R, int64_t n, int d, int m, float *sigma, unsigned *sigmax,
                               const unsigned *bound = nullptr, unsigned short *qflag = nullptr, int64_t flag_row0 = 0, lsq_q16_params *P = nullptr) {
    if (d > 256) hipLaunchKernelGGL(unary_shift_kernel<true>, grid, dim3(256), 0, s, X, R, n, d, m, sigma, sigmax, bound, qflag, flag_row0, P);
    else hipLaunchKernelGGL(unary_shift_kernel<false>, grid, dim3(256), 0, s, X, R, n, d, m, sigma, sigmax, bound, qflag, flag_row0, P);
}
#define LSQ_SHIFT_LAUNCH(D_, GRID_, S_, ...) launch_unary_shift(GRID_, S_, __VA_ARGS__)

// One block: value ranges -> lsq_q16_params (in double).
//   U_j: the minimum / maximum over a SAMPLE of the chunk's vectors (range-only GEMM pass, qrange = order-preserving keys), widened by 1/8 of
//        the width on both sides.  Vectors the sample did not predict are flagged by the GEMM epilogue (non-finite values included) and
//        take the f32 path, so the range only has to be good, not guaranteed.
//   T_jk: exact range (table_range_kernel).
// ok = 0 -- the whole chunk goes to the f32 walk -- when a pair table or the sample holds a non-finite value, or a range degenerates.
__global__ __launch_bounds__(64) void q16_params_kernel(const float *__restrict__ trange, const int *__restrict__ bad, const unsigned *__restrict__ qrange,
                                                        int m, lsq_q16_params *__restrict__ P, unsigned *__restrict__ sigrange, float sig_scale) {
    if (threadIdx.x != 0) return;
    // max |sigma| of the chunk (bit pattern of a non-negative float) -- or, sig_scale > 1: of a SAMPLE of it, widened; sigrange[1] then tells the
    // panel-wise unary_shift_kernel launches which bound the parameters assumed (vectors beyond it are flagged)
    const float sigb = sigrange ? __uint_as_float(sigrange[0]) * sig_scale + (sig_scale > 1.0f ? 1e-30f : 0.0f) : 0.0f;
    if (sigrange) sigrange[1] = __float_as_uint(sigb);
    const double sigmax = (double)sigb;
    bool ok = (bad[0] == 0) && (qrange[2 * LSQ_MAX_M] == 0);
    for (int j = 0; j < m; ++j) {
        const unsigned kl = qrange[2 * j], kh = qrange[2 * j + 1];
        lsq_q16_node nd;
        nd.loU = 0.0f; nd.invD = 0.0f; nd.D = 0.0f; nd.hiq = 0.0f; nd.window = 65535; nd.slack = 0.0;
        if (kl > kh) { ok = false; P->node[j] = nd; continue; }      // empty sample
        const double sl = (double)__uint_as_float((kl & 0x80000000u) ? (kl ^ 0x80000000u) : ~kl);
        const double sh = (double)__uint_as_float((kh & 0x80000000u) ? (kh ^ 0x80000000u) : ~kh);
        const double mg = 0.125 * (sh - sl) + 1e-30 + fmax(fabs(sl), fabs(sh)) / 1048576.0;
        double loU = sl - mg;
        const double hiU = sh + mg;
        nd.loU = (float)loU;
        if ((double)nd.loU > loU) nd.loU = nextafterf(nd.loU, -__builtin_inff());
        loU = (double)nd.loU;
        // W = the sampled bound of the shifted unary w = (v + G[a]) + sigma_i;  the unshifted |v| <= W + sigmax + Gmax
        const double W = fmax(fabs(loU), fabs(hiU));
        double Gmax = 0.0;                                     // |colshift| <= SUM_k max |g_jk|
        double rsum = hiU - loU, tsum = 0.0, sub = 0.0;
        for (int k = 0; k < m; ++k) {
            if (k == j) continue;
            const double tr = (double)trange[(j * m + k) * 3] * (1.0 + 1e-6), tm = (double)trange[(j * m + k) * 3 + 1], gm = (double)trange[(j * m + k) * 3 + 2];
            rsum += tr;                                        // centred-entry range
            tsum += tm;
            Gmax += gm;
            sub += 2.0 * (tm + gm) + tr;                       // the two f32 subtractions that centre a table entry: t - g[a], then - rowmin[b]
        }
        const double smax = W + sigmax + Gmax + tsum;          // magnitudes along the canonical f32 chain U + T + T + ...
        sub += (W + Gmax + sigmax) * 3.0 + 2.0 * Gmax;         // the unary side: v + G[a], the row's lo_row = loU - sigma, their difference; the f32 rounding of G itself
        const double D = rsum / 65500.0;
        // f32 chain vs real sum: <= m roundings of <= 2^-24 smax (x2 margin);  + every f32 rounding on the way to a level's input, <= 2^-24 of the
        // magnitudes collected in `sub` (x2 margin)
        const double eps = (double)(m + 1) * smax * 5.9604644775390625e-8 * 2.0 + sub * 5.9604644775390625e-8 * 2.0;
        const double slack = (double)m * (0.5 + 1.0 / 32.0) * D + eps + 65535.0 * D * 2.384185791015625e-7;      // + the f32 rounding of D and 1/D over 65535 levels
        nd.D = (float)D;
        nd.invD = (float)(1.0 / D);
        // levels of one sum: U <= hiq, table k <= R_k / D + 1/2  =>  sum <= 65500 + m / 2 + rounding < 65535: packed (and plain 32-bit) adds never carry
        nd.hiq = (float)fmin(floor((hiU - loU) / D), 65535.0);
        nd.slack = slack;
        const double w = 2.0 * slack / D;
        nd.window = (w < 30000.0) ? (int)w + 1 : 65535;
        if (!(D > 1e-30 && D < 1e30 && rsum == rsum && smax < 1e30)) ok = false;
        if (!(nd.invD > 1e-30f && nd.invD < 1e30f)) ok = false;
        P->node[j] = nd;
    }
    P->ok = ok ? 1 : 0;          // P->oor accumulates over the chunks of a call (zeroed by the host at its start)
    P->nflag = 0;                // per chunk: raised by the GEMM epilogue that follows
}

// Tq[j][slice][row q16_row_index(kk, b)][SLQ] (u16)  <-  rint(((T[j][k(kk)][b][slice*SLQ ..] - g_jk[a]) - rowmin[j][k][b]) * invD_j)      (one thread per 8 levels = 16 B)
template <int SLQ>
__global__ __launch_bounds__(256) void tables_to_q16_slices_kernel(const float *__restrict__ T, uint16_t *__restrict__ Tq, int m,
                                                                   const lsq_q16_params *__restrict__ P, const float *__restrict__ rowmin,
                                                                   const float *__restrict__ colmean) {
    constexpr int NS = LSQ_H / SLQ, LPV = SLQ / 8;
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
    const float lo = rowmin[((int64_t)j * m + k) * LSQ_H + b], inv = P->node[j].invD;      // the centred row's own minimum: a shift common to all candidates
    const float *src = T + (((int64_t)j * m + k) * LSQ_H + b) * LSQ_H + slice * SLQ + qq * 8;
    const float *gs = colmean + ((int64_t)j * m + k) * LSQ_H + slice * SLQ + qq * 8;          // the candidates' column means (moved to the unary levels)
    uint32_t w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float q0 = fminf(fmaxf(rintf(((src[2 * t] - gs[2 * t]) - lo) * inv), 0.0f), 65535.0f);
        const float q1 = fminf(fmaxf(rintf(((src[2 * t + 1] - gs[2 * t + 1]) - lo) * inv), 0.0f), 65535.0f);
        w[t] = (uint32_t)q0 | ((uint32_t)q1 << 16);
    }
    const int64_t dst = (((int64_t)j * NS + slice) * (m - 1) * LSQ_H + q16_row_index<SLQ>(m, kk, b)) * LPV + qq;      // (plain geometry: dst == e)
    reinterpret_cast<u32x4 *>(Tq)[dst] = (u32x4){w[0], w[1], w[2], w[3]};
}

// Exact refinement of the block's ambiguous vectors (records arec[0 .. namb): {ci | a1 << 16 | a2 << 24, limit, record words}, written by
// the decide phase): 16 lanes per vector, 4 vectors per wave at a time.  ONE dependent global round trip in the common case:
//   * lane t recomputes the level sums of candidates [16 t, 16 t + 16) from the u16 planes and the u16 slice tables (the codes come from
//     the LDS record, so the table loads do not wait for anything) and keeps the survivors: those within the window of the best key,
//     which provably include the exact argmin and all its exact ties;
//   * IN PARALLEL the exact f32 terms of the two best candidates a1, a2 (known from the keys) are loaded speculatively, one term per lane
//     (lanes 0..7: a1, 8..15: a2; term 0 = unary, 1.. = table entries in ascending k) and summed in canonical order through shuffles;
//   * survivors other than a1 / a2 (rare) take a second trip: q16_exact_value per survivor.
// The lexicographic (value, index) minimum over everything evaluated is the argmin.  Kept lean: it shares the slice walk's 128 VGPRs.
template <int M, int SLQ, int NT>
__device__ inline int q16_refine(const float *__restrict__ U, const uint16_t *__restrict__ Uq, const uint16_t *__restrict__ Tq,
                                 const float *__restrict__ T, uint8_t *__restrict__ rec, unsigned short *__restrict__ valid,
                                 const uint8_t *__restrict__ ref_rec, const unsigned short *__restrict__ ref_valid, int64_t n, int j,
                                 int64_t lo, const unsigned short *list, const uint32_t *arec, int namb, int SLF, int abl, unsigned short *vmir, bool list_hole = false) {      // list_hole: 16 entries in every 128 (WalkqRot::LHOLE)
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int RW = CS / 4;
    constexpr int AREC = 2 + RW;
    if (abl & 4) return 0;
    constexpr int NS = LSQ_H / SLQ;
    constexpr int TAB = (M - 1) * LSQ_H * (SLQ / 8);
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, t16 = lane & 15;
    const int sl = (16 * t16) / SLQ, off = (16 * t16) % SLQ;              // slice / offset (in candidates) of this lane's 16 candidates
    const uint16_t *__restrict__ Uqj = Uq + (int64_t)j * n * LSQ_H;
    const uint16_t *__restrict__ Tqj = Tq + ((int64_t)j * NS + sl) * TAB * 8 + off;
    const float *__restrict__ Tj = T + (int64_t)j * M * LSQ_H * LSQ_H;
    const bool have_ref = ref_rec && ref_valid;
    int nexact = 0;
    for (int r0 = wave * 4; r0 < namb; r0 += NW * 4) {
        const int r = r0 + grp;
        const bool act = r < namb;
        const uint32_t *ar = arec + (act ? r : r0) * AREC;
        const uint32_t key = ar[0], limit = ar[1];
        uint32_t rw[RW];
#pragma unroll
        for (int w2 = 0; w2 < RW; ++w2) rw[w2] = ar[2 + w2];
        const int ci = (int)(key & 0xffffu), a1 = (int)((key >> 16) & 0xffu), a2 = (int)(key >> 24);
        const int64_t i = lo + (list_hole ? list[((ci >> 4) << 7) + (ci & 15)] : list[ci]);
        // ---- the one round trip: unary levels, table levels, bookkeeping (lane 0), speculative exact terms (one per lane)
        const u32x4 *up = reinterpret_cast<const u32x4 *>(Uqj + ((int64_t)sl * n + ((abl & 2) ? (int64_t)0 : i)) * SLQ + off);
        u32x4 s0 = (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, s1 = s0;
        if (!(abl & 8)) { s0 = up[0]; s1 = up[1]; }
        unsigned short vo = 0, rv = 0;
        uint32_t rr[RW];
#pragma unroll
        for (int w2 = 0; w2 < RW; ++w2) rr[w2] = 0;
        if (t16 == 0) {
            if (valid) vo = valid[i];
            if (have_ref) {
#pragma unroll
                for (int w2 = 0; w2 < RW; ++w2) rr[w2] = reinterpret_cast<const uint32_t *>(ref_rec + i * CS)[w2];
                rv = ref_valid[i];
            }
        }
        float term = 0.0f;                                                // lanes 0..M-1: terms of a1, lanes 8..8+M-1: terms of a2 (M <= 8); M > 8: slow path only
        if (M <= 8) {
            const int cand = (t16 < 8) ? a1 : a2, tt = t16 & 7;           // term tt: 0 = unary, q >= 1 = table of the q-th conditioning codebook
            if (abl & 1) term = 1.0f;
            else if (tt == 0) term = U[(int64_t)j * n * LSQ_H + ((int64_t)(cand / SLF) * n + ((abl & 16) ? lo : i)) * SLF + (cand % SLF)];
            else if (tt < M) {
                const int k = (tt - 1) + ((tt - 1) >= j ? 1 : 0);
                const uint32_t bk = (rw[k >> 2] >> (8 * (k & 3))) & 0xffu;
                term = Tj[((int64_t)(k * LSQ_H) + bk) * LSQ_H + cand];
            }
        }
#pragma unroll
        for (int kk = 0; kk < M - 1; ++kk) {
            if (abl & 8) break;
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
            // a1 and a2 are survivors by construction; drop them from the masks and rank them here
            if (a1 / 16 == t16) mask &= ~(1u << (a1 % 16));
            if (a2 / 16 == t16) mask &= ~(1u << (a2 % 16));
            bv = e1; bi = a1;
            if (e2 < bv || (e2 == bv && a2 < bi)) { bv = e2; bi = a2; }
            if (act && t16 == 0) nexact += 2;
        }
        while (mask) {                                                    // third survivors (or every survivor when M > 8): one more trip
            const int a = 16 * t16 + __builtin_ctz(mask);
            mask &= mask - 1;
            const float ev = q16_exact_value<M, RW>(U, T, n, SLF, j, i, rw, a);
            ++nexact;
            if (ev < bv || (ev == bv && a < bi)) { bv = ev; bi = a; }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 16);
            const int oi = __shfl_xor(bi, o, 16);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (act && t16 == 0) {                                            // apply_node_result, its loads done above
            const uint8_t c8 = (uint8_t)bi;
            const uint8_t old = (uint8_t)(rw[j >> 2] >> (8 * (j & 3)));
            if (c8 != old) store_code<CS>(rec, i, j, c8, rw);
            if (valid) {
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
                valid[i] = vm;
                if (vmir) vmir[i - lo] = vm;
            }
        }
    }
    return nexact;
}

// dynamic LDS of icm_walkq_kernel: the arrays (table, keys, active list, validity mirror), then the block's scalars
template <int M, int SLQ, int CPL, int NT, int BPC>
constexpr int walkq_main_bytes() {
    if (walkq_rot(M, SLQ, CPL, NT, BPC)) return WalkqRot<M>::lds_bytes();
    const int pp = WalkqTab<SLQ, CPL>::pp(M, BPC);
    return WalkqTab<SLQ, CPL>::lds_entries(M) * 16 + pp * 8 + pp * 2 + (WalkqTab<SLQ, CPL>::mirror(M, BPC) ? pp * 2 : 0);
}
constexpr int WALKQ_MISC_BYTES = (20 + LSQ_WALK_COUNTERS) * 4;

// ---- the filtered walk ----------------------------------------------------------------------------------------------------------
// Block / pass / node structure, compaction of the active vectors, light blocks and the validity bookkeeping are those of
// icm_walk_kernel; the slice walk runs on 16-bit levels (slices of SLQ = 32 candidates for m <= 8, 16 above: the same 64 / 32-byte
// pieces and the same LDS table footprint as the f32 walk, half as many slices).
template <int M, int SLQ, int CPL, int DEPTH, int NT, int BPC>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 64 * BPC / 4, NT / 64 * BPC / 4))) void icm_walkq_kernel(const float *__restrict__ U, const uint16_t *__restrict__ Uq, const uint16_t *__restrict__ Tq,
                                                       const float *__restrict__ T, uint8_t *__restrict__ rec, unsigned short *__restrict__ valid,
                                                       int64_t n, const WalkNodes nodes, int per_pass, int use_skip, int direct_max,
                                                       unsigned long long *__restrict__ active_total,
                                                       const uint8_t *__restrict__ ref_rec, const unsigned short *__restrict__ ref_valid,
                                                       const lsq_q16_params *__restrict__ P, int SLF, const unsigned short *__restrict__ qflag, int abl,
                                                       const unsigned *__restrict__ gate) {
    constexpr int CS = (M <= 8) ? 8 : 16;
    constexpr int NS = LSQ_H / SLQ;
    using TL = WalkqTab<SLQ, CPL>;
    constexpr int NR = TL::NR;                           // 16-byte reads per lane and table (8 levels each)
    constexpr int LPV = TL::LPV;                         // lanes per vector: CPL levels per lane
    constexpr int EPR = TL::EPR;                         // 16-byte entries per table row (global layout)
    constexpr int VPW = 64 / LPV;
    constexpr int CW = (M - 1 + 3) / 4;
    constexpr int RW = CS / 4;
    constexpr int TAB = (M - 1) * LSQ_H * EPR;           // 16-byte entries of one slice table (global)
    constexpr int LTAB = TL::lds_entries(M);             // ... in LDS (planes, skew)
    using RT = WalkqRot<M>;
    constexpr bool ROT = walkq_rot(M, SLQ, CPL, NT, BPC);   // rotated-rows placement of the slice table (lsq_q16.h): the default geometry up to m = 8
    constexpr int PP = ROT ? RT::pp() : TL::pp(M, BPC);  // BPC = 1: the f32 walk's geometry (4096 up to m = 14); BPC = 2: two 512-thread blocks share a CU
    if (P->ok == 0) return;                              // never launched in that case (the host read the verdict after the GEMM); kept as a guard
    if (gate && *gate != 2u) return;                     // option "async": the chunk's road word (q16_road_kernel) names the f32 walk
#ifdef LSQ_TUNING
    unsigned long long *dbgp = nullptr;
    extern __shared__ u32x4 lds_walkq[];
    // No static __shared__ in this kernel: the dynamic segment must start at LDS address 0 -- the rotated placement builds table addresses byte-wise
    // (v_perm_b32) and has no instruction to spare for a segment base.  The block's few scalars live behind the arrays (walkq_misc_words()).
    constexpr int MAIN_BYTES = walkq_main_bytes<M, SLQ, CPL, NT, BPC>();
    int *misc = reinterpret_cast<int *>(reinterpret_cast<char *>(lds_walkq) + MAIN_BYTES);
    if (g_walkq_dbg) {
        unsigned &dbg_slot_s = reinterpret_cast<unsigned *>(misc)[19];
        if (threadIdx.x == 0) dbg_slot_s = (blockIdx.x == 0) ? atomicAdd(&g_walkq_dbg_slot, 1u) : 0u;
        __syncthreads();
        // only block 0 knows the slot; other blocks use the launch's slot through a second counter-free trick: they record nothing
        if (blockIdx.x == 0 && dbg_slot_s < 4096) dbgp = g_walkq_dbg + (size_t)dbg_slot_s * LSQ_WALKQ_DBG_WORDS;
    }
    DBG_STAMP(0);
    if (blockIdx.x == 0 && threadIdx.x == 0) g_walkq_dbg_cur = dbgp;
    if (g_walkq_blk && threadIdx.x == 0) g_walkq_blk[2 * blockIdx.x] = wall_clock64();
#endif
#ifndef LSQ_TUNING
    extern __shared__ u32x4 lds_walkq[];
    constexpr int MAIN_BYTES = walkq_main_bytes<M, SLQ, CPL, NT, BPC>();
    int *misc = reinterpret_cast<int *>(reinterpret_cast<char *>(lds_walkq) + MAIN_BYTES);      // (see the note in the tuning branch above: no static __shared__ here)
#endif
    if constexpr (walkq_rot(M, SLQ, CPL, NT, BPC)) {
        if ((uint32_t)(uintptr_t)(lds_char *)lds_walkq != 0u) __builtin_trap();      // the rotated placement addresses LDS by number: the segment must start at 0
    }
    u32x4 *tab = lds_walkq;
    constexpr bool HOLE = ROT && RT::HOLE;                                                       // bestA in the free slot of the table's second group (m <= 8)
    constexpr bool LHOLE = ROT && RT::LHOLE;                                                     // the active list in the free slot of the second group (m > 8)
    constexpr bool ROT8 = ROT && M <= 8, ROT16 = ROT && M > 8;
    constexpr int TABE = ROT ? RT::TAB_BYTES / 16 : LTAB;
    uint32_t *bestA = HOLE ? reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lds_walkq) + RT::HOLE_BYTE0)
                           : reinterpret_cast<uint32_t *>(lds_walkq + TABE);                     // [PP] smallest key (Q << 16 | candidate); HOLE: 16 words in every 64 (keyA())
    uint32_t *bestB = HOLE ? reinterpret_cast<uint32_t *>(lds_walkq + TABE) : bestA + PP;        // [PP] second smallest key
    unsigned short *list = LHOLE ? reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(lds_walkq) + RT::HOLE_BYTE0)
                                 : reinterpret_cast<unsigned short *>(bestB + PP);             // [PP] active local indices; LHOLE: 16 entries in every 128 (listp())
    auto listp = [&](int i) -> unsigned short * { return LHOLE ? list + ((i >> 4) << 7) + (i & 15) : list + i; };
    constexpr bool MIRROR = ROT ? RT::mirror() : TL::mirror(M, BPC);
    auto keyA = [&](int ci) -> uint32_t * { return HOLE ? bestA + ((ci >> 4) << 6) + (ci & 15) : bestA + ci; };
    // the f32-path list (16-bit entries) reuses bestA's storage after the decide phase has read the keys
    auto f32slot = [&](int i) -> unsigned short * { return reinterpret_cast<unsigned short *>(keyA(i >> 1)) + (i & 1); };
    unsigned short *vmir = (MIRROR && valid) ? list + PP : nullptr;      // (never with LHOLE)                              // [PP] mirror of valid[lo ..): read by the compaction, written with every store to valid[]
    int *wave_tot = misc;                                // [16]
    int &nact_s = misc[16];
    int &redo_s = misc[17];
    int &f32_s = misc[18];
    // statistics are accumulated per block in LDS and flushed ONCE per launch: per-node device atomics from 256 blocks on the same few
    // words sat in front of every node update's first barrier (5-9 us per node, profiles/r02j_walkq_phases.txt "pre")
    unsigned *stat_s = reinterpret_cast<unsigned *>(misc) + 20;      // [LSQ_WALK_COUNTERS]
    for (int e = threadIdx.x; e < LSQ_WALK_COUNTERS; e += NT) stat_s[e] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int v = lane / LPV, q = lane % LPV;
    constexpr int NW = NT / 64;
    constexpr int EPT = 4096 / NT;
    constexpr int step = NW * VPW;

    struct Item { u32x4 u[NR]; uint32_t r[RW]; };

    auto walk_slices = [&](const int j, const int64_t lo, const int nact, const bool dense) {
#ifndef LSQ_TUNING
        unsigned long long *dbgp = nullptr; (void)dbgp;
#endif
        const uint16_t *__restrict__ Uqj = Uq + (int64_t)j * n * LSQ_H;
        const uint16_t *__restrict__ Tqj = Tq + (int64_t)j * NS * TAB * 8;
        uint32_t sel[CW > 0 ? CW : 1];
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
        constexpr int NST = (TAB + NT - 1) / NT;
        u32x4 nxt[NST > 0 ? NST : 1];
        // rotated rows (lsq_q16.h, WalkqRot): lane constants of this node.  selC0 / selC1 gather the record's code bytes of the lane's t-th read of group 0 / 1
        // (slot (t + v) mod tables-of-the-group, table kk = 4 group + slot, code k = kk + (kk >= j)); base0 / base1 hold the matching slot | lane_q address bytes
        // (byte 3 of base1 = 1: the second group starts at 65536).
        uint32_t selC0 = 0x0c0c0c0cu, selC1 = 0x0c0c0c0cu, base0 = 0u, base1 = 0x01000000u;
        // m > 8: the record's 15 code bytes are first compressed (the node's own code dropped: sel[], uniform) into four words; rotA / rotB (rot1A / rot1B) then pick
        // group 0's (group 1's) bytes in the lane's read order -- the t-th read of a group goes to slot (t + v) mod tables-of-the-group -- and b0a .. b1b hold the
        // matching slot | lane_q address bytes.  Group 1 is addressed as (line + slot + 16 bytes) + 0xfff0: its base 65536 does not fit the 16-bit offset field.
        uint32_t rotA = 0x0c0c0c0cu, rotB = 0x0c0c0c0cu, rot1A = 0x0c0c0c0cu, rot1B = 0x0c0c0c0cu, b0a = 0u, b0b = 0u, b1a = 0u, b1b = 0u;
        if constexpr (ROT16) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t < RT::NT0) {
                    const uint32_t slot = (uint32_t)(t + v) % (uint32_t)RT::NT0;
                    const uint32_t ab = ((slot << 5) | ((uint32_t)q << 4)) << (8 * (t & 3));
                    if (t < 4) { rotA = (rotA & ~(0xffu << (8 * t))) | (slot << (8 * t)); b0a |= ab; }
                    else { rotB = (rotB & ~(0xffu << (8 * (t - 4)))) | (slot << (8 * (t - 4))); b0b |= ab; }
                }
                if (t < RT::NT1) {
                    const uint32_t slot = (uint32_t)(t + v) % (uint32_t)(RT::NT1 ? RT::NT1 : 1);
                    const uint32_t ab = ((slot << 5) + (((uint32_t)q + 1u) << 4)) << (8 * (t & 3));
                    if (t < 4) { rot1A = (rot1A & ~(0xffu << (8 * t))) | (slot << (8 * t)); b1a |= ab; }
                    else { rot1B = (rot1B & ~(0xffu << (8 * (t - 4)))) | (slot << (8 * (t - 4))); b1b |= ab; }
                }
            }
        }
        if constexpr (ROT8) {
#pragma unroll
            for (int t = 0; t < RT::NT0; ++t) {
                const uint32_t slot = (uint32_t)(t + v) % (uint32_t)(RT::NT0 ? RT::NT0 : 1);
                const uint32_t k = slot + (slot >= (uint32_t)j ? 1u : 0u);
                selC0 = (selC0 & ~(0xffu << (8 * t))) | (k << (8 * t));
                base0 |= ((slot << 6) | ((uint32_t)q << 4)) << (8 * t);
            }
#pragma unroll
            for (int t = 0; t < RT::NT1; ++t) {
                const uint32_t slot = (uint32_t)(t + v) % (uint32_t)(RT::NT1 ? RT::NT1 : 1);
                const uint32_t k = 4u + slot + (4u + slot >= (uint32_t)j ? 1u : 0u);
                selC1 = (selC1 & ~(0xffu << (8 * t))) | (k << (8 * t));
                base1 |= ((slot << 6) | ((uint32_t)q << 4)) << (8 * t);
            }
        }
        const uint32_t v4 = (uint32_t)v * 4u;
        constexpr uint32_t LIST_BYTE0 = (uint32_t)(ROT && !LHOLE ? RT::TAB_BYTES + PP * RT::KEY_BYTES : 0);      // byte address of list[] under the rotated placement
        const int limv = q == 0 ? nact - v : -0x7fffffff;      // c0 < limv  <=>  q == 0 and c0 + v < nact: one compare per item
        // staging of a slice: global rows are [group][code][slot] (q16_row_index) -- a straight copy for a group of four tables; a group of nt < 4 tables
        // leaves 4 - nt slots of every 256-byte LDS line free
        constexpr bool GROT = (SLQ == 32 && M <= 8) || (SLQ == 16 && M > 8);          // the global layout of this geometry
        auto rot_entry = [&](auto R_) -> int {
            constexpr int r = decltype(R_)::value;
            constexpr int g = r * NT >= RT::G0_ENTRIES ? 1 : 0, nt = g ? RT::NT1 : RT::NT0, e0 = r * NT - g * RT::G0_ENTRIES;      // (group 0 is a whole number of rounds)
            const int eg = e0 + (int)threadIdx.x;             // entry inside the group
            if constexpr (nt == RT::SPL) return g * 4096 + eg;
            else { const int code = eg / (RT::EPS * (nt ? nt : 1)); return g * 4096 + code * 16 + (eg - code * RT::EPS * nt); }
        };
        auto prefetch_tab = [&](int sl) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(Tqj) + (int64_t)sl * TAB;
#pragma unroll
            for (int r = 0; r < NST; ++r) {
                const int e = (int)threadIdx.x + r * NT;
                if constexpr (GROT && !ROT)                   // (tuning geometries on slices of 32: plain LDS placement from the rotated global rows)
                    nxt[r] = (e < TAB) ? src[q16_row_index<SLQ>(M, e / (LSQ_H * EPR), (e / EPR) % LSQ_H) * EPR + e % EPR] : (u32x4){0u, 0u, 0u, 0u};
                else nxt[r] = (e < TAB) ? src[e] : (u32x4){0u, 0u, 0u, 0u};
            }
        };
        prefetch_tab(0);
        const int ipw = (wave * VPW < nact) ? (nact - wave * VPW + step - 1) / step : 0;
        int ls = 0, lit = 0;
        // the slice plane of the level stream the NEXT loaded item belongs to: advanced when the wave's items wrap around (uniform, scalar)
        const char *ub = reinterpret_cast<const char *>(Uqj + lo * SLQ);
        const char *const rb = reinterpret_cast<const char *>(rec + lo * CS);
        const int64_t plane_bytes = n * (int64_t)(SLQ * 2);
        auto load_next = [&](Item &it) {
            int ci = wave * VPW + lit * step + v;
            ci = ci < nact ? ci : nact - 1;
            uint32_t li = (uint32_t)ci;
            if (!dense) {
                if constexpr (ROT && !LHOLE) li = *reinterpret_cast<lds_u16 *>(LIST_BYTE0 + 2u * (uint32_t)ci);      // (LDS by number: no segment-base add)
                else li = *listp(ci);
            }
#ifdef LSQ_TUNING
            const uint32_t uo = ((abl & 32) ? (li & 63u) : li) * (uint32_t)(SLQ * 2) + (uint32_t)q * 16u;      // ablation: the level stream from L2 instead of HBM
#else
            const uint32_t uo = li * (uint32_t)(SLQ * 2) + (uint32_t)q * 16u;
#endif
#pragma unroll
            for (int r = 0; r < NR; ++r) it.u[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ub + uo + r * (LPV * 16)));
#ifdef LSQ_TUNING
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rb + ((abl & 512) ? (li & 63u) : li) * (uint32_t)CS);      // ablation: records from 64 hot lines
#else
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(rb + li * (uint32_t)CS);
#endif
#pragma unroll
            for (int w = 0; w < RW; ++w) it.r[w] = rp[w];
            if (++lit >= ipw) {
                lit = 0;
                if (++ls < NS) ub += plane_bytes;           // past the last slice the (unused) preloads re-read the last plane
            }
        };
        auto compute = [&](const Item &cur, int slice, int c0) {
            u32x4 s[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) s[r] = cur.u[r];
            // The two 16-bit halves of a word never carry into each other (the sum of the m levels of a candidate stays below 65536: lsq_q16_node::hiq),
            // so plain 32-bit adds are exact on the packed levels -- and v_add3_u32 takes two table rows per instruction where v_pk_add_u16 takes one.
            uint32_t code[M > 1 ? M - 1 : 1];                // (plain placement only)
#pragma unroll
            for (int w = 0; w < (ROT ? 0 : CW); ++w) {
                const uint32_t hiw = (w + 1 < RW) ? cur.r[w + 1] : 0u;
                const uint32_t cw = __builtin_amdgcn_perm(hiw, cur.r[w], sel[w]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * w + t;
                    if (kk < M - 1) {
                        if (t == 0) asm("v_and_b32 %0, 0xff, %1" : "=v"(code[kk]) : "v"(cw));
                        else code[kk] = (cw >> (8 * t)) & 0xffu;
                    }
                }
            }
#ifdef LSQ_TUNING
            if (!ROT && (abl & 1024)) {                      // timing only: rows whose bank position depends on the vector's place in the wave, not on its codes (no conflicts)
#pragma unroll
                for (int kk = 0; kk < M - 1; ++kk) code[kk] = (code[kk] & ~(uint32_t)(64 / (LPV * 16) * 4 - 1)) | (uint32_t)((v + kk) & (256 / (LPV * 16) - 1));
            }
#endif
            if constexpr (ROT16) {
                uint32_t dw[4] = {0u, 0u, 0u, 0u};               // the compressed code bytes kk = 0 .. m - 2
#pragma unroll
                for (int w = 0; w < CW; ++w) dw[w] = __builtin_amdgcn_perm((w + 1 < RW) ? cur.r[w + 1] : 0u, cur.r[w], sel[w]);
                const uint32_t cw4[4] = {__builtin_amdgcn_perm(dw[1], dw[0], rotA), __builtin_amdgcn_perm(dw[1], dw[0], rotB),
                                         __builtin_amdgcn_perm(dw[3], dw[2], rot1A), __builtin_amdgcn_perm(dw[3], dw[2], rot1B)};
                const uint32_t bw4[4] = {b0a, b0b, b1a, b1b};
                // four reads at a time (16 registers in flight), summed before the next four are requested
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int cnt = g4 < 2 ? (RT::NT0 - 4 * g4 < 4 ? RT::NT0 - 4 * g4 : 4) : (RT::NT1 - 4 * (g4 - 2) < 4 ? RT::NT1 - 4 * (g4 - 2) : 4);
                    if (cnt <= 0) continue;
                    u32x4 rd[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (t < cnt) {
                            const uint32_t a = __builtin_amdgcn_perm(cw4[g4], bw4[g4], 0x0c0c0400u + 0x0101u * (uint32_t)t);      // [slot | lane_q] [code] [0] [0]
                            rd[t] = *reinterpret_cast<lds_cu32x4 *>(a + (g4 < 2 ? 0u : 0xfff0u));
                        }
#ifdef LSQ_TUNING
                    if (!(abl & 64))
#endif
                    {
#pragma unroll
                        for (int t = 0; t < 4; t += 2) {
                            if (t + 1 < cnt) {
                                s[0].x = s[0].x + rd[t].x + rd[t + 1].x; s[0].y = s[0].y + rd[t].y + rd[t + 1].y;
                                s[0].z = s[0].z + rd[t].z + rd[t + 1].z; s[0].w = s[0].w + rd[t].w + rd[t + 1].w;
                            } else if (t < cnt) {
                                s[0].x += rd[t].x; s[0].y += rd[t].y; s[0].z += rd[t].z; s[0].w += rd[t].w;
                            }
                        }
                    }
                    if (g4 < 3) asm volatile("" : "+v"(s[0].x), "+v"(s[0].y), "+v"(s[0].z), "+v"(s[0].w) : : "memory");
                }
            } else if constexpr (ROT8) {
                u32x4 rd[M > 1 ? M - 1 : 1];
                const uint32_t c0r = __builtin_amdgcn_perm(cur.r[1], cur.r[0], selC0);
                // LDS addresses as plain numbers (the segment starts at 0, checked at kernel entry): through the segment's symbol every read pays an add of its base
#pragma unroll
                for (int t = 0; t < RT::NT0; ++t)            // address bytes: [slot | lane_q] [code] [0] [0]
                    rd[t] = *reinterpret_cast<lds_cu32x4 *>(__builtin_amdgcn_perm(c0r, base0, 0x0c0c0400u + 0x0101u * (uint32_t)t));
                auto add_rows = [&](int k0, int k1) {
#ifdef LSQ_TUNING
                    if (abl & 64) return;                    // ablation: no table rows
#endif
#pragma unroll
                    for (int kk = k0; kk < k1; kk += 2) {
                        if (kk + 1 < k1) {
                            s[0].x = s[0].x + rd[kk].x + rd[kk + 1 < M - 1 ? kk + 1 : kk].x; s[0].y = s[0].y + rd[kk].y + rd[kk + 1 < M - 1 ? kk + 1 : kk].y;
                            s[0].z = s[0].z + rd[kk].z + rd[kk + 1 < M - 1 ? kk + 1 : kk].z; s[0].w = s[0].w + rd[kk].w + rd[kk + 1 < M - 1 ? kk + 1 : kk].w;
                        } else {
                            s[0].x += rd[kk].x; s[0].y += rd[kk].y; s[0].z += rd[kk].z; s[0].w += rd[kk].w;
                        }
                    }
                };
                if constexpr (RT::NT1 > 0) {
                    // the second group's rows are read only when the first group's are summed: seven reads in flight hold 28 registers, and what the
                    // loop cannot keep is spilled AROUND it -- the compaction and the decide phase then wait for scratch (measured: +20 us per node update)
                    const uint32_t c1r = __builtin_amdgcn_perm(cur.r[1], cur.r[0], selC1);
                    add_rows(0, RT::NT0);
                    asm volatile("" : "+v"(s[0].x), "+v"(s[0].y), "+v"(s[0].z), "+v"(s[0].w) : : "memory");
#pragma unroll
                    for (int t = 0; t < RT::NT1; ++t)        // ... [1] [0]: the second group
                        rd[RT::NT0 + t] = *reinterpret_cast<lds_cu32x4 *>(__builtin_amdgcn_perm(c1r, base1, 0x0c030400u + 0x0101u * (uint32_t)t));
                    add_rows(RT::NT0, M - 1);
                } else add_rows(0, M - 1);
            } else {
#ifdef LSQ_TUNING
            if (!(abl & 64))                                 // ablation: no table rows
#endif
#pragma unroll
            for (int kk = 0; kk < M - 1; kk += 2) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const u32x4 ra = tab[kk * TL::TS_E + r * TL::PLANE_E + (int)code[kk] * LPV + q];
                    if (kk + 1 < M - 1) {
                        const u32x4 rb2 = tab[(kk + 1) * TL::TS_E + r * TL::PLANE_E + (int)code[kk + 1 < M - 1 ? kk + 1 : kk] * LPV + q];
                        s[r].x = s[r].x + ra.x + rb2.x; s[r].y = s[r].y + ra.y + rb2.y;
                        s[r].z = s[r].z + ra.z + rb2.z; s[r].w = s[r].w + ra.w + rb2.w;
                    } else {
                        s[r].x += ra.x; s[r].y += ra.y; s[r].z += ra.z; s[r].w += ra.w;
                    }
                }
            }
            }
#ifdef LSQ_TUNING
            if (abl & 128) {                                 // ablation: no keys / top-2 / LDS atomics
                if ((q == 0) & (c0 + v < nact) & (s[0].x == 0x12345678u)) *keyA(c0 + v) = s[0].y ^ s[NR - 1].z;
                return;
            }
#endif
            // keys (Q << 16 | candidate): the two smallest of the lane's CPL, then of the vector's LPV lanes.  Inside the lane the keys carry the
            // candidate's offset from the lane's first one (inline constants: chunk r starts LPV * 8 r further); the base is added to the two survivors
            uint32_t l0 = 0, h0 = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                constexpr uint32_t HI = 0xffff0000u;
                const uint32_t o = (uint32_t)(r * LPV * 8);
                const uint32_t k0 = (s[r].x << 16) | o, k1 = (s[r].x & HI) | (o + 1u);
                const uint32_t k2 = (s[r].y << 16) | (o + 2u), k3 = (s[r].y & HI) | (o + 3u);
                const uint32_t k4 = (s[r].z << 16) | (o + 4u), k5 = (s[r].z & HI) | (o + 5u);
                const uint32_t k6 = (s[r].w << 16) | (o + 6u), k7 = (s[r].w & HI) | (o + 7u);
                uint32_t la = umin3(k0, k1, k2), ha = umed3(k0, k1, k2);      // two triples + a pair: 6 + 2 x 3 instructions instead of 8 + 3 x 3
                const uint32_t lb = umin3(k3, k4, k5), hb = umed3(k3, k4, k5);
                const uint32_t lc = umin(k6, k7), hc = umax(k6, k7);
                top2_merge(la, ha, lb, hb);
                top2_merge(la, ha, lc, hc);
                if (r == 0) { l0 = la; h0 = ha; } else top2_merge(l0, h0, la, ha);
            }
            const uint32_t base = (uint32_t)(SLQ * slice) + 8u * (uint32_t)q;
            l0 += base; h0 += base;                                          // candidate < 256: never carries into the level
            if (LPV >= 2) top2_merge(l0, h0, dpp_u32<DPP_XOR1>(l0), dpp_u32<DPP_XOR1>(h0));
            if (LPV >= 4) top2_merge(l0, h0, dpp_u32<DPP_XOR2>(l0), dpp_u32<DPP_XOR2>(h0));
            if (c0 < limv) {                                 // the vector's first lane, vector c0 + v inside the active list
                if constexpr (HOLE) {      // c0 is a multiple of 16 and v < 16: the smallest key of vector c0 + v sits at byte 16 c0 + 4 v of the free slot
                    const uint32_t sA = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)RT::HOLE_BYTE0 + (uint32_t)c0 * 16u));      // scalar parts: one add per address
                    const uint32_t sB = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)RT::TAB_BYTES + (uint32_t)c0 * 4u));
                    const uint32_t old = __hip_atomic_fetch_min(reinterpret_cast<lds_u32 *>(sA + v4), l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_min(reinterpret_cast<lds_u32 *>(sB + v4), umin(umax(old, l0), h0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    const uint32_t old = atomicMin(&bestA[c0 + v], l0);      // returns the previous minimum: the larger of the two is a runner-up
                    atomicMin(&bestB[c0 + v], umin(umax(old, l0), h0));
                }
            }
        };
        Item buf[DEPTH];
#pragma unroll
        for (int e = 0; e < DEPTH; ++e) load_next(buf[e]);
        int phase = 0;
        for (int slice = 0; slice < NS; ++slice) {
            __syncthreads();
#ifdef LSQ_TUNING
            if (slice < 8) DBG_STAMP(3 + slice);
#endif
            if constexpr (ROT) {
                static_for<NST>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    if ((r + 1) * NT <= TAB || r * NT + (int)threadIdx.x < TAB) tab[rot_entry(R_)] = nxt[r];      // (only the last round can be short)
                });
            } else {
#pragma unroll
                for (int r = 0; r < NST; ++r) {
                    const int e = (int)threadIdx.x + r * NT;
                    if (e < TAB) tab[TL::entry(e / (LSQ_H * EPR), (e / EPR) % LSQ_H, e % EPR)] = nxt[r];      // NT = 1024, EPR = 4: table r, the thread's fixed (code, chunk)
                }
            }
            __syncthreads();
#ifdef LSQ_TUNING
            if (slice == 4) DBG_STAMP(20);
#endif
            if (slice + 1 < NS) prefetch_tab(slice + 1);
#ifdef LSQ_TUNING
            if (slice == 4) DBG_STAMP(21);
#endif
            int c0 = wave * VPW, t = 0;
            auto run = [&](auto P_) {
                constexpr int PH = decltype(P_)::value;
                for (; t + DEPTH <= ipw; t += DEPTH) {
#pragma unroll
                    for (int e = 0; e < DEPTH; ++e) {
                        compute(buf[(PH + e) % DEPTH], slice, c0); load_next(buf[(PH + e) % DEPTH]); c0 += step;
                    }
                }
#pragma unroll
                for (int e = 0; e < DEPTH - 1; ++e)
                    if (t < ipw) {
                        compute(buf[(PH + e) % DEPTH], slice, c0); load_next(buf[(PH + e) % DEPTH]); c0 += step;
                        ++t; phase = (PH + e + 1) % DEPTH;
                    }
            };
            bool ran = false;
            auto try_phase = [&](auto P_) {
                if constexpr (decltype(P_)::value < DEPTH) {
                    if (!ran && phase == decltype(P_)::value) { run(P_); ran = true; }
                }
            };
            try_phase(std::integral_constant<int, 0>{}); try_phase(std::integral_constant<int, 1>{});
            try_phase(std::integral_constant<int, 2>{}); try_phase(std::integral_constant<int, 3>{});
#ifdef LSQ_TUNING
            if (slice == 4) DBG_STAMP(22);
#endif
        }
        __syncthreads();
    };

    // one wave per vector, everything in f32: the light-block routine of icm_walk_kernel (also the last resort of the filter);
    // `count` list entries, entry r names the local index through pick(r)
    constexpr int LB = LSQ_LIGHT_LB(M);
    int64_t lo_cur = 0;                                  // first vector of the block's current pass
    auto light_list = [&](const int j, const int count, auto pick) {
        const float *__restrict__ Usj = U + (int64_t)j * n * LSQ_H;
        const float *__restrict__ Tj = T + (int64_t)j * M * LSQ_H * LSQ_H;
        for (int r0 = wave; r0 < count; r0 += NW * LB) {
            int64_t vi[LB];
            bool on[LB];
#pragma unroll
            for (int e = 0; e < LB; ++e) {
                const int r = r0 + e * NW;
                on[e] = r < count;
                vi[e] = lo_cur + __builtin_amdgcn_readfirstlane((int)pick(on[e] ? r : r0));
            }
            light_update<M, CS, LB>(rec, valid, ref_rec, ref_valid, Usj, Tj, n, SLF, j, vi, on, lane, vmir, lo_cur);
        }
    };

    const int64_t npass = (n + per_pass - 1) / per_pass;
    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int64_t lo = pass * per_pass;
        const int64_t hi = (lo + per_pass < n) ? lo + per_pass : n;
        const int cnt = (int)(hi - lo);
        lo_cur = lo;
        if (vmir) {
            __syncthreads();                                   // (a previous pass' readers are done)
            for (int idx = (int)threadIdx.x; idx < cnt; idx += NT) vmir[idx] = valid[lo + idx];
            __syncthreads();
        }
        for (int nu = 0; nu < nodes.count; ++nu) {
            const int j = nodes.j[nu];
            // the thread's index as the node update sees it: opaque, so that the per-thread LDS addresses derived from it (keys, list, mirror) are computed
            // where they are used -- as loop invariants they are hoisted out of the node loop, spilled, and every phase then opens with scratch reloads
            int tix = (int)threadIdx.x;
            asm volatile("" : "+v"(tix));
#ifdef LSQ_TUNING
            if (dbgp && threadIdx.x == 0 && pass == (int64_t)blockIdx.x && nu < 64) dbgp[24 + nu] = wall_clock64();
#endif
            {
                const int base = tix * EPT;
                int f[EPT], c = 0;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int idx = base + e;
                    f[e] = 0;
                    if (idx < cnt) f[e] = (!use_skip) || !(((vmir ? vmir[idx] : valid[lo + idx]) >> j) & 1);
                    c += f[e];
                }
                int inc = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += t;
                }
                if (lane == 63) wave_tot[wave] = inc;
                DBG_STAMP(16);
                __syncthreads();
                DBG_STAMP(17);
                int wbase = 0;
                for (int w2 = 0; w2 < wave; ++w2) wbase += wave_tot[w2];
                int pos = wbase + inc - c;
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    if (f[e]) *listp(pos++) = (unsigned short)(base + e);
                if (threadIdx.x == NT - 1) { nact_s = wbase + inc; redo_s = 0; f32_s = 0; }
                DBG_STAMP(18);
                __syncthreads();
            }
            const int nact = nact_s;
            DBG_STAMP(1);
#ifdef LSQ_TUNING
            if (dbgp && threadIdx.x == 0 && pass == (int64_t)blockIdx.x && nu < 64) dbgp[24 + 65 + nu] = (unsigned long long)nact;
#endif
            if (nact == 0) { __syncthreads(); continue; }
            if (threadIdx.x == 0) {                        // [0] node updates recomputed, [2] light / [3] filtered block-node-updates
                stat_s[0] += (unsigned)nact;
                stat_s[nact <= direct_max ? 2 : 3] += 1u;
                stat_s[4 + ((nodes.pos0 + nu) & (LSQ_WALK_TRACE - 1))] += (unsigned)nact;
            }
            if (nact <= direct_max) {                      // light block: full f32 gathers from L2, no staging
                light_list(j, nact, [&](int r) { return *listp(r); });
                __syncthreads();
                continue;
            }
            for (int ci = tix; ci < nact; ci += NT) { *keyA(ci) = 0xffffffffu; bestB[ci] = 0xffffffffu; }
            DBG_STAMP(2);
            walk_slices(j, lo, nact, nact == cnt);
            DBG_STAMP(11);
            // ---- decide.  second - best > window: the best key is the exact argmin.  Otherwise the vector is AMBIGUOUS: every
            // candidate whose level sum lies within the window of the best (the "survivors": they provably include the exact
            // argmin and all its exact ties) is evaluated exactly below.
            const lsq_q16_node &nd = P->node[j];
            const int window = nd.window;
            constexpr int EPD = (PP + NT - 1) / NT;                // vectors per thread
            constexpr int AREC = 2 + RW;                           // words of an ambiguous-vector record: {ci | a1 << 16 | a2 << 24, limit, record words}
            constexpr int ACAP = (PP * 4) / (AREC * 4);            // records that fit bestB's storage
            {
                int64_t vi[EPD];
                uint32_t vcode[EPD], vkey[EPD], vlim[EPD];
                bool von[EPD], vamb[EPD], vf32[EPD];
                // m <= 8: every global load of the thread's vectors is issued before the first use (ONE round trip: the loads do not wait for the flag
                // word that says whether the vector is decided, ambiguous or out of the level range) and before the first store.  m > 8 (16-byte
                // records, twice the registers) keeps the two-trip order: flags first, then only what the vector's class needs -- measured faster there
                constexpr bool ONE_TRIP = (M <= 8);
                uint32_t rw[EPD][RW], rr[EPD][RW];
                unsigned short vo[EPD], rv[EPD], fl[EPD];
                uint32_t kA[EPD], kB[EPD];
                const bool have_ref = ref_rec && ref_valid;
#pragma unroll
                for (int e = 0; e < EPD; ++e) {
                    const int ci = tix + e * NT;
                    vi[e] = lo; kA[e] = 0; kB[e] = 0; vo[e] = 0; rv[e] = 0; fl[e] = 0;
#pragma unroll
                    for (int w2 = 0; w2 < RW; ++w2) { rw[e][w2] = 0; rr[e][w2] = 0; }
                    if (ci < nact) {
                        kA[e] = *keyA(ci); kB[e] = bestB[ci];
                        vi[e] = lo + *listp(ci);
                        fl[e] = qflag[vi[e]];
                        if constexpr (ONE_TRIP) {
#pragma unroll
                            for (int w2 = 0; w2 < RW; ++w2) rw[e][w2] = reinterpret_cast<const uint32_t *>(rec + vi[e] * CS)[w2];
                            if (valid) vo[e] = valid[vi[e]];
                            if (have_ref) {
#pragma unroll
                                for (int w2 = 0; w2 < RW; ++w2) rr[e][w2] = reinterpret_cast<const uint32_t *>(ref_rec + vi[e] * CS)[w2];
                                rv[e] = ref_valid[vi[e]];
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < EPD; ++e) {
                    const int ci = tix + e * NT;
                    von[e] = false; vamb[e] = false; vf32[e] = false; vcode[e] = 0; vkey[e] = 0; vlim[e] = 0;
                    if (ci < nact) {
                        vcode[e] = kA[e] & 0xffffu;
                        vkey[e] = (uint32_t)ci | ((kA[e] & 0xffu) << 16) | ((kB[e] & 0xffu) << 24);
                        vlim[e] = (kA[e] >> 16) + (uint32_t)window;
                        vf32[e] = (fl[e] >> j) & 1;                    // a unary of this node fell outside the sampled level range: its levels mean nothing
                        vamb[e] = !vf32[e] && ((int)(kB[e] >> 16) - (int)(kA[e] >> 16) <= window);
                        von[e] = !vamb[e] && !vf32[e];
                    }
                    if constexpr (!ONE_TRIP) {                         // ambiguous vectors only need their record
                        if (von[e] || vamb[e]) {
#pragma unroll
                            for (int w2 = 0; w2 < RW; ++w2) rw[e][w2] = reinterpret_cast<const uint32_t *>(rec + vi[e] * CS)[w2];
                        }
                        if (von[e]) {
                            if (valid) vo[e] = valid[vi[e]];
                            if (have_ref) {
#pragma unroll
                                for (int w2 = 0; w2 < RW; ++w2) rr[e][w2] = reinterpret_cast<const uint32_t *>(ref_rec + vi[e] * CS)[w2];
                                rv[e] = ref_valid[vi[e]];
                            }
                        }
                    }
                }
                __syncthreads();                                       // every bestA[] / bestB[] has been read: their storage is reused below
                uint32_t *arec = bestB;                                // ambiguous-vector records
#pragma unroll
                for (int e = 0; e < EPD; ++e) {
                    if (von[e]) {
                        const uint8_t c8 = (uint8_t)vcode[e];
                        const uint8_t old = (uint8_t)(rw[e][j >> 2] >> (8 * (j & 3)));
                        if (c8 != old) store_code<CS>(rec, vi[e], j, c8, rw[e]);      // an unchanged record is not written
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
                            valid[vi[e]] = vm;
                            if (vmir) vmir[vi[e] - lo] = vm;
                        }
                    }
                    bool tof32 = vf32[e];
                    if (vamb[e]) {
                        const int slot = atomicAdd(&redo_s, 1);
                        if (slot < ACAP) {
                            arec[slot * AREC] = vkey[e];
                            arec[slot * AREC + 1] = vlim[e];
#pragma unroll
                            for (int w2 = 0; w2 < RW; ++w2) arec[slot * AREC + 2 + w2] = rw[e][w2];
                        } else tof32 = true;                           // more ambiguous vectors than records (degenerate data): full f32 for the rest
                    }
                    if (tof32) *f32slot(atomicAdd(&f32_s, 1)) = (unsigned short)(vkey[e] & 0xffffu);      // vectors for the f32 path
                }
            }
            __syncthreads();
            DBG_STAMP(12);
            // ---- exact refinement of the ambiguous vectors
            const int namb = redo_s < ACAP ? redo_s : ACAP;
            int nexact = q16_refine<M, SLQ, NT>(U, Uq, Tq, T, rec, valid, ref_rec, ref_valid, n, j, lo, list, bestB, namb, SLF, abl, vmir, LHOLE);
            {   // vectors outside the sampled level range: one wave each, in full f32
                const int nf32 = f32_s;
                light_list(j, nf32, [&](int r) { return *listp(*f32slot(r)); });
                if (threadIdx.x == 0 && nf32) atomicAdd(&stat_s[4 + LSQ_WALK_TRACE + 2], (unsigned)nf32);
            }
            {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) nexact += __shfl_xor(nexact, o, 64);
                if (lane == 0 && nexact) atomicAdd(&stat_s[4 + LSQ_WALK_TRACE + 1], (unsigned)nexact);
                if (threadIdx.x == 0 && namb) atomicAdd(&stat_s[4 + LSQ_WALK_TRACE], (unsigned)namb);
            }
            __syncthreads();
            DBG_STAMP(13);
#ifdef LSQ_TUNING
            if (dbgp && threadIdx.x == 0) { dbgp[14] = (unsigned long long)nact; dbgp[15] = (unsigned long long)namb; }
#endif
        }
    }
    __syncthreads();
#ifdef LSQ_TUNING
    if (dbgp && threadIdx.x == 0) dbgp[24 + 64] = wall_clock64();
    if (g_walkq_blk && threadIdx.x == 0) g_walkq_blk[2 * blockIdx.x + 1] = wall_clock64();
#endif
    if (active_total)
        for (int e = threadIdx.x; e < LSQ_WALK_COUNTERS; e += NT)
            if (stat_s[e]) atomicAdd(active_total + e, (unsigned long long)stat_s[e]);
}

}  // namespace

#ifdef LSQ_TUNING
// tools only: device buffer of 4096 x LSQ_WALKQ_DBG_WORDS (154) u64 that block 0 of every icm_walkq_kernel launch fills with phase timestamps (wall_clock64)
extern "C" __attribute__((visibility("default"))) int lsq_tuning_set_walkq_debug(void *buf) {
    unsigned zero = 0;
    LSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_walkq_dbg), &buf, sizeof(buf)));
    LSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_walkq_dbg_slot), &zero, sizeof(zero)));
    return LSQ_OK;
}
// tools only: device buffer of 512 x 2 u64: start / end clock (wall_clock64, 100 MHz) of every block of the latest icm_walkq_kernel launch
extern "C" __attribute__((visibility("default"))) int lsq_tuning_set_walkq_block_clock(void *buf) {
    LSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_walkq_blk), &buf, sizeof(buf)));
    return LSQ_OK;
}
#endif

// ---- the chunk's road decided ON THE DEVICE (option "async": no host round trip anywhere in the call) ------------------------------------------
// road[0]: 2 = the filtered walk runs this chunk (icm_walkq_kernel's `gate`; the f32 walk launched behind it idles on the same word), 0 = the f32 walk
// does; road[1] += 1 whenever a chunk is handed to the f32 walk.  Same rules as the host's: verdict after the unary GEMM (usable bounds, at most
// 1 / fallback_div of the (vector, node) pairs outside the sampled level range), probe after the first ILS iteration (or first sweep).
__global__ void q16_road_kernel(const lsq_q16_params *__restrict__ P, unsigned *__restrict__ road, long long pairs, long long fallback_div) {
    if (threadIdx.x != 0) return;
    const bool filtered = P->ok == 1 && (long long)P->nflag * fallback_div <= pairs;
    road[0] = filtered ? 2u : 0u;
    if (!filtered) road[1] += 1u;
}
__global__ void q16_probe_kernel(const unsigned long long *__restrict__ probe, unsigned long long *__restrict__ totals, unsigned *__restrict__ road,
                                 unsigned long long probe_div) {
    const int e = threadIdx.x;
    if (e < LSQ_WALK_COUNTERS && totals) totals[e] += probe[e];      // the probed launches' statistics join the call's
    if (e == 0 && road[0] == 2u && probe_div > 0) {
        const unsigned long long hard = probe[4 + LSQ_WALK_TRACE] + probe[4 + LSQ_WALK_TRACE + 2];
        if ((double)hard * (double)probe_div > (double)probe[0]) { road[0] = 0u; road[1] += 1u; }
    }
}
int lsq_launch_q16_road(hipStream_t s, const lsq_q16_params *P, unsigned *road, int64_t pairs, int64_t fallback_div) {
    hipLaunchKernelGGL(q16_road_kernel, dim3(1), dim3(64), 0, s, P, road, (long long)pairs, (long long)fallback_div);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}
int lsq_launch_q16_probe(hipStream_t s, const unsigned long long *probe, unsigned long long *totals, unsigned *road, int64_t probe_div) {
    hipLaunchKernelGGL(q16_probe_kernel, dim3(1), dim3(128), 0, s, probe, totals, road, (unsigned long long)probe_div);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

const void *lsq_probe_kernel_icmq() { return reinterpret_cast<const void *>(&q16_range_init_kernel); }

int lsq_q16_slice_width(int m) {
#ifdef LSQ_TUNING
    if (m <= 8 && LSQ_KNOB("LSQ_WALKQ_BPC", 1) == 2) return 16;
#endif
    return 2 * lsq_walk_slice_width(m);
}      // candidates per 16-bit slice: the same bytes per piece as the f32 walk

int lsq_launch_unary_shift_panel(hipStream_t s, const float *Xp, int64_t rows, int d, int m, const float *means, float *sigma_p, unsigned *qrange,
                                 unsigned short *qflag, int64_t row0, lsq_q16_params *P) {
    if (rows <= 0) return LSQ_OK;
    const int64_t shift_blocks = (rows * 16 + 255) / 256;
    // the chunk maximum is not collected here (a scratch word takes it): the parameters were fixed from the sample
    LSQ_SHIFT_LAUNCH(d, dim3((unsigned)(shift_blocks < 2048 ? shift_blocks : 2048)), s, Xp, means, rows, d, m, sigma_p,
                       qrange + 2 * LSQ_MAX_M + 3, qrange + 2 * LSQ_MAX_M + 2, qflag, row0, P);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

int lsq_q16_sample_rows(int64_t n, int d, int64_t *rts_out) {
    const int64_t nsample = d <= 128 ? 16384 : (d <= 512 ? 8192 : 4096);      // the pass costs 2 d m h flops per sampled vector: fewer of them at large d
    const int64_t rts = n > nsample ? n / nsample : 1;
    if (rts_out) *rts_out = rts;
    return (int)((n + 128 * rts - 1) / (128 * rts));      // number of 128-row panels in the sample (the last one may be short)
}

int lsq_launch_q16_prepare(hipStream_t s, const float *X, int64_t n, int d, const float *K, const float *sci, const float *T, int m, uint16_t *Tq,
                           int *bad, float *trange, unsigned *qrange, unsigned short *qflag, lsq_q16_params *P, int tables_changed,
                           float *rowmin, float *means, float *sigma, float *colmean, float *colshift, const float *Xsample, int64_t nsample_rows,
                           float *sigma_sample) {
    // bad[0]: a non-finite pair table (per call); rowmin / colmean [m*m*256], colshift [m*256], means [m*d]: per call; sigma [n*m]: per chunk;
    // qrange[2*16 + 1]: sample flag, [2*16 + 2]: max |sigma|
    if (tables_changed) {
        LSQ_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
        hipLaunchKernelGGL(table_colmean_kernel, dim3((unsigned)(m * m)), dim3(256), 0, s, T, m, colmean);
        hipLaunchKernelGGL(unary_colshift_kernel, dim3((unsigned)m), dim3(256), 0, s, colmean, m, colshift);
        if (m > 1) hipLaunchKernelGGL(table_range_kernel, dim3((unsigned)(m * m)), dim3(256), 0, s, T, m, colmean, trange, rowmin, bad);
        hipLaunchKernelGGL(codebook_means_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)m), dim3(256), 0, s, K, m, d, means);
    }
    // sampled range of the SHIFTED unaries: about 16 384 vectors at d <= 128 (every rts-th panel of 128 consecutive ones) through the range-only GEMM pass
    hipLaunchKernelGGL(q16_range_init_kernel, dim3(1), dim3(64), 0, s, qrange, m);      // zeros; min slots start at the largest key (one launch: m + 1 fills cost 5 us each)
    if (Xsample) {
        // host-buffer pipeline: the sample (the same rows the strided pass would read) was uploaded ahead of the panels, compacted; sigma of the sample,
        // its maximum (widened below) and the value ranges come from it alone; the panels' own sigma follow panel by panel (lsq_launch_unary_shift_panel)
        const int64_t shift_blocks = (nsample_rows * 16 + 255) / 256;
        LSQ_SHIFT_LAUNCH(d, dim3((unsigned)(shift_blocks < 2048 ? shift_blocks : 2048)), s, Xsample, means, nsample_rows, d, m,
                           sigma_sample, qrange + 2 * LSQ_MAX_M + 1);
        LSQ_TRY(lsq_launch_chain_gemm(s, Xsample, K, sci, -2.0f, nsample_rows, m * LSQ_H, d, LSQ_H, 0, 0, nullptr, 0, nsample_rows, 0, nullptr, 0, nullptr, 0, nullptr,
                                      qrange, 1, sigma_sample, colshift));
        LSQ_HIP(hipMemsetAsync(qflag, 0, sizeof(unsigned short) * (size_t)((n + 1) & ~(int64_t)1), s));
    } else if (n > 0) {
        const int64_t shift_blocks = (n * 16 + 255) / 256;           // 16 vectors per block per pass, at most 8 blocks per CU in flight
        LSQ_SHIFT_LAUNCH(d, dim3((unsigned)(shift_blocks < 2048 ? shift_blocks : 2048)), s, X, means, n, d, m, sigma,
                           qrange + 2 * LSQ_MAX_M + 1);
        int64_t rts64 = 1;
        (void)lsq_q16_sample_rows(n, d, &rts64);
        const int rts = (int)rts64;
        LSQ_TRY(lsq_launch_chain_gemm(s, X, K, sci, -2.0f, n, m * LSQ_H, d, LSQ_H, 0, 0, nullptr, 0, n, 0, nullptr, 0, nullptr, 0, nullptr, qrange, rts, sigma,
                                      colshift));
        LSQ_HIP(hipMemsetAsync(qflag, 0, sizeof(unsigned short) * (size_t)((n + 1) & ~(int64_t)1), s));
    }
    hipLaunchKernelGGL(q16_params_kernel, dim3(1), dim3(64), 0, s, trange, bad, qrange, m, P, qrange + 2 * LSQ_MAX_M + 1, Xsample ? 2.0f : 1.0f);
    if (m > 1) {
        const int slq = lsq_q16_slice_width(m);
        const int64_t total = (int64_t)m * (LSQ_H / slq) * (m - 1) * LSQ_H * (slq / 8);
        const unsigned grid = (unsigned)((total + 255) / 256);
        if (slq == 32) hipLaunchKernelGGL(tables_to_q16_slices_kernel<32>, dim3(grid), dim3(256), 0, s, T, Tq, m, P, rowmin, colmean);
        else hipLaunchKernelGGL(tables_to_q16_slices_kernel<16>, dim3(grid), dim3(256), 0, s, T, Tq, m, P, rowmin, colmean);
    }
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

template <int M, int SLQ, int CPL, int DEPTH, int NT, int BPC>
static int launch_walkq_t(hipStream_t s, const float *U, const uint16_t *Uq, const uint16_t *Tq, const float *T, uint8_t *rec, unsigned short *valid,
                          int64_t n, const WalkNodes &nodes, int use_skip, unsigned long long *active_total, int light,
                          const uint8_t *ref_rec, const unsigned short *ref_valid, const lsq_q16_params *P, const unsigned short *qflag, const unsigned *gate) {
    constexpr bool ROT = walkq_rot(M, SLQ, CPL, NT, BPC);
    constexpr int PP = ROT ? WalkqRot<M>::pp() : WalkqTab<SLQ, CPL>::pp(M, BPC);
    constexpr int LDS_BYTES = walkq_main_bytes<M, SLQ, CPL, NT, BPC>() + WALKQ_MISC_BYTES;      // slice table + two smallest keys + active list (+ validity mirror) + the block's scalars
    static_assert(LDS_BYTES * BPC <= 160 * 1024 && WALKQ_MISC_BYTES <= 768, "slice table + keys must fit the block's share of the 160 KiB LDS");
    constexpr int NBLK = 256 * BPC;
    const int64_t rounds = (n + NBLK * (int64_t)PP - 1) / (NBLK * (int64_t)PP);      // passes per block
    int64_t per = rounds > 0 ? (n + NBLK * rounds - 1) / (NBLK * rounds) : 1;
    per = per > PP ? PP : (per < 1 ? 1 : per);
    const int per_pass = (int)per, npass = (int)((n + per - 1) / per);
    const int direct_max = light >= 0 ? light : LSQ_KNOB("LSQ_WALK_DIRECT", 160);
    const int skip = (use_skip && valid) ? 1 : 0;
    static LdsOptIn optin;
    LSQ_TRY(optin_lds(optin, &icm_walkq_kernel<M, SLQ, CPL, DEPTH, NT, BPC>, LDS_BYTES));
    if (ROT) {      // the rotated placement addresses LDS by number: the kernel must have been compiled without static LDS (it also traps at entry otherwise)
        static std::atomic<int> static_lds_known{-1};      // (lsq_multi_* runs one host thread per device through here)
        int static_lds = static_lds_known.load(std::memory_order_relaxed);
        if (static_lds < 0) {
            hipFuncAttributes fa;
            LSQ_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&icm_walkq_kernel<M, SLQ, CPL, DEPTH, NT, BPC>)));
            static_lds = (int)fa.sharedSizeBytes;
            static_lds_known.store(static_lds, std::memory_order_relaxed);
        }
        if (static_lds != 0) { lsq_set_error("icm_walkq_kernel<%d>: %d bytes of static LDS in a kernel that addresses LDS from 0", M, static_lds); return LSQ_EHIP; }
    }
    const unsigned grid = (unsigned)(npass < NBLK ? npass : NBLK);
    hipLaunchKernelGGL((icm_walkq_kernel<M, SLQ, CPL, DEPTH, NT, BPC>), dim3(grid), dim3(NT), LDS_BYTES, s, U, Uq, Tq, T, rec, valid, n, nodes, per_pass, skip,
                       direct_max, active_total, skip ? ref_rec : nullptr, skip ? ref_valid : nullptr, P, lsq_walk_slice_width(M), qflag, LSQ_KNOB("LSQ_Q16_ABL", 0), gate);
    LSQ_HIP(hipGetLastError());
    return LSQ_OK;
}

// the filtered counterpart of lsq_launch_icm_walk (the caller has read the chunk's verdict on the host)
int lsq_launch_icm_walkq(hipStream_t s, const float *U, const uint16_t *Uq, const uint16_t *Tq, const float *T, uint8_t *rec, unsigned short *valid,
                         int64_t n, int m, const int32_t *order, int nnodes, int pos0, int use_skip, unsigned long long *active_total, int light,
                         const uint8_t *ref_rec, const unsigned short *ref_valid, const lsq_q16_params *P, const unsigned short *qflag, const unsigned *gate) {
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
#define LSQ_WQ_ARGS s, U, Uq, Tq, T, rec, valid, n, nodes, use_skip, active_total, light, ref_rec, ref_valid, P, qflag, gate
#define LSQ_WQ_CASE(MM, SLL, CPLL, DD, NTT) case MM: LSQ_TRY((launch_walkq_t<MM, SLL, CPLL, DD, NTT, 1>(LSQ_WQ_ARGS))); break;
#define LSQ_WQ_CASE2(MM, SLL, CPLL, DD, NTT) case MM: LSQ_TRY((launch_walkq_t<MM, SLL, CPLL, DD, NTT, 2>(LSQ_WQ_ARGS))); break;
        // m <= 8: slices of 32 candidates, four lanes per vector (8 candidates per lane); above: slices of 16, two lanes of 8
#ifdef LSQ_TUNING
        if (m <= 8 && LSQ_KNOB("LSQ_WALKQ_CPL", 8) == 16) {      // two lanes per vector, 16 candidates per lane: 22 % fewer instructions per candidate, dense
            switch (m) {                                          // node updates 7 % faster, sparse ones up to 30 % slower (coarser items): DESIGN.md 4.2
                LSQ_WQ_CASE(1, 32, 16, 3, 1024) LSQ_WQ_CASE(2, 32, 16, 3, 1024) LSQ_WQ_CASE(3, 32, 16, 3, 1024) LSQ_WQ_CASE(4, 32, 16, 3, 1024)
                LSQ_WQ_CASE(5, 32, 16, 3, 1024) LSQ_WQ_CASE(6, 32, 16, 3, 1024) LSQ_WQ_CASE(7, 32, 16, 3, 1024) LSQ_WQ_CASE(8, 32, 16, 3, 1024)
            }
            continue;
        }
#endif
#ifdef LSQ_TUNING
        if (m <= 8 && LSQ_KNOB("LSQ_WALKQ_BPC", 1) == 2) {       // two 512-thread blocks per CU, slices of 16 candidates
            switch (m) {
                LSQ_WQ_CASE2(1, 16, 8, 3, 512) LSQ_WQ_CASE2(2, 16, 8, 3, 512) LSQ_WQ_CASE2(3, 16, 8, 3, 512) LSQ_WQ_CASE2(4, 16, 8, 3, 512)
                LSQ_WQ_CASE2(5, 16, 8, 3, 512) LSQ_WQ_CASE2(6, 16, 8, 3, 512) LSQ_WQ_CASE2(7, 16, 8, 3, 512) LSQ_WQ_CASE2(8, 16, 8, 3, 512)
            }
            continue;
        }
#endif
        switch (m) {
            LSQ_WQ_CASE(1, 32, 8, 3, 1024) LSQ_WQ_CASE(2, 32, 8, 3, 1024) LSQ_WQ_CASE(3, 32, 8, 3, 1024) LSQ_WQ_CASE(4, 32, 8, 3, 1024)
            LSQ_WQ_CASE(5, 32, 8, 3, 1024) LSQ_WQ_CASE(6, 32, 8, 3, 1024) LSQ_WQ_CASE(7, 32, 8, 3, 1024) LSQ_WQ_CASE(8, 32, 8, 3, 1024)
            LSQ_WQ_CASE(9, 16, 8, 2, 1024) LSQ_WQ_CASE(10, 16, 8, 2, 1024) LSQ_WQ_CASE(11, 16, 8, 2, 1024) LSQ_WQ_CASE(12, 16, 8, 2, 1024)
            LSQ_WQ_CASE(13, 16, 8, 2, 1024) LSQ_WQ_CASE(14, 16, 8, 2, 1024) LSQ_WQ_CASE(15, 16, 8, 2, 1024) LSQ_WQ_CASE(16, 16, 8, 2, 1024)
        }
#undef LSQ_WQ_CASE
#undef LSQ_WQ_CASE2
#undef LSQ_WQ_ARGS
    }
    return LSQ_OK;
}
