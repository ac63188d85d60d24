// lsq_lsqr.hip -- the codebook update ON THE DEVICE: all d least-squares problems of update_codebooks at once.
//
// Reference: update_codebooks(X, B, h) (src/codebook_update.jl:52-86): K[t, :] = lsqr(S, X[t, :]) for every dimension t, S = sparsify_codes(B, h)
// (src/utils.jl:50-69) the n x (m h) matrix with one 1 per codebook and row; IterativeSolvers.lsqr defaults (x0 = 0, damp = 0,
// atol = btol = sqrt(eps(Float32)), conlim = 1e8, maxiter = max(size(S))) -- un-vendored and un-pinned by the reference (PARITY UNPINNED).
// This is the SAME restatement of Paige & Saunders' LSQR as the host code (lsq_codebook.hip: Float32 recurrences, the three long sums -- ||u||, S'u,
// ||v|| -- accumulated in double, the same stopping rules), with the d right-hand sides advanced together: the vectors of a system are the
// columns of [n][d] / [m h][d] matrices (the layout of X and K), its scalars live in arrays of d, and a system that has met its stopping rule
// freezes while the others go on.  S is never materialised:
//     (S v)[i]    = sum_j v[j h + b_ij]                         gather, codebooks ascending
//     (S' u)[c]   = sum of u[i] over the rows i that hold code c  the rows are SORTED by code once per call (radix sort of (column, row) keys):
//                                                               a thread walks its column's segment in ascending row order -- the host's
//                                                               order of addition, no atomics -- accumulating in double
// The norms are summed in a fixed order too (partial sums of fixed item sets, combined in a fixed order), so a call returns the same bits every time;
// S'u is added in the host's own order, the norms are not (the host adds its squares one by one: the double sums differ in their last bits, their Float32 roundings have not differed on any tested problem -- tests require 1e-5 and compare both solvers with scipy's).
// One iteration = two passes over U (n d floats each, the second through the sorted rows, m times) and a few over V, plus d-thread scalar kernels.
#include <hipcub/hipcub.hpp>

#include <cmath>

#include "lsq_internal.h"

#pragma clang fp contract(off)

namespace {

struct Scal {        // per-system scalars, arrays of d floats each
    float *alpha, *beta, *rhobar, *phibar, *Anorm, *ddnorm, *xnorm, *xxnorm, *z, *sn2, *cs2, *bnorm, *rho, *t1, *t2, *ia, *ib, *phi, *theta, *tau;
    int *done;       // 1: the system has stopped
    double *sumU, *sumV, *dk2;
    int *active;     // [1] systems still running (written by the last scalar kernel of an iteration)
};

constexpr int TB = 64;        // systems (dimensions) per block column
constexpr int RS = 64;        // rows per block of the row passes

// keys of the sort: (column j h + b_ij) << 32 | row
__global__ __launch_bounds__(256) void lsqr_make_keys(const uint8_t *__restrict__ codes, int64_t n, int m, uint64_t *__restrict__ keys) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * m) return;
    const int64_t i = e / m;
    const int j = (int)(e % m);
    keys[e] = ((uint64_t)(j * LSQ_H + codes[e]) << 32) | (uint64_t)i;
}
// seg[c] = first sorted position of column c (seg[cols] = n m): binary search in the sorted keys
__global__ void lsqr_segments(const uint64_t *__restrict__ sorted, int64_t total, int cols, int64_t *__restrict__ seg) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > cols) return;
    const uint64_t want = (uint64_t)c << 32;
    int64_t lo = 0, hi = total;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] < want) lo = mid + 1; else hi = mid; }
    seg[c] = lo;
}

// first pass: u = b.  later passes: u = S v - alpha (u ib)   (u is stored UNSCALED: fl(u ib) -- the host's scaled vector -- is formed wherever it is read).
// part[block][t] = the block's sum of u^2 (double, rows ascending).
__global__ __launch_bounds__(256) void lsqr_u_update(const float *__restrict__ X, float *__restrict__ U, const float *__restrict__ V, const uint8_t *__restrict__ codes,
                                                     int64_t n, int d, int m, Scal S, double *__restrict__ part, int init) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int sub = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + sub) * RS;
    __shared__ double red[4][TB];
    double acc = 0.0;
    if (t < d && !(S.done[t])) {
        const float alpha = S.alpha[t], ib = S.ib[t];
        const int64_t r1 = (r0 + RS < n) ? r0 + RS : n;
        int64_t i = r0;
        constexpr int G = 8;                                       // rows in flight: their loads are issued together, the sums keep the row order
        for (; i + G <= r1; i += G) {
            float u[G];
            if (init) {
#pragma unroll
                for (int q = 0; q < G; ++q) u[q] = X[(i + q) * d + t];
            } else {
                float old[G], sv[G];
#pragma unroll
                for (int q = 0; q < G; ++q) { old[q] = U[(i + q) * d + t]; sv[q] = 0.0f; }
                for (int j = 0; j < m; ++j) {                      // codebooks ascending within every row, as the host adds them
                    float g[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) g[q] = V[((int64_t)j * LSQ_H + codes[(i + q) * m + j]) * d + t];
#pragma unroll
                    for (int q = 0; q < G; ++q) sv[q] += g[q];
                }
#pragma unroll
                for (int q = 0; q < G; ++q) u[q] = -alpha * (old[q] * ib) + sv[q];
            }
#pragma unroll
            for (int q = 0; q < G; ++q) { U[(i + q) * d + t] = u[q]; acc += (double)u[q] * (double)u[q]; }
        }
        for (; i < r1; ++i) {
            float u;
            if (init) u = X[i * d + t];
            else {
                const uint8_t *c = codes + i * m;
                float sv = 0.0f;
                for (int j = 0; j < m; ++j) sv += V[((int64_t)j * LSQ_H + c[j]) * d + t];
                u = -alpha * (U[i * d + t] * ib) + sv;
            }
            U[i * d + t] = u;
            acc += (double)u * (double)u;
        }
    }
    red[sub][threadIdx.x & (TB - 1)] = acc;
    __syncthreads();
    if (sub == 0 && t < d) part[(int64_t)blockIdx.x * d + t] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// v = (float)(S'(u ib))  (first pass)  or  (float)(-beta v + S'(u ib)):  one thread per (column, system) walks the column's rows in ascending order
__global__ __launch_bounds__(TB) void lsqr_v_update(float *__restrict__ V, const float *__restrict__ U, const uint64_t *__restrict__ sorted,
                                                    const int64_t *__restrict__ seg, int d, Scal S, int first) {
    const int c = blockIdx.x, t = blockIdx.y * TB + threadIdx.x;
    if (t >= d || S.done[t] || !(S.beta[t] > 0.0f)) return;
    const float ib = S.ib[t];
    const int64_t e0 = seg[c], e1 = seg[c + 1];
    double acc = 0.0;
    int64_t e = e0;
    for (; e + 8 <= e1; e += 8) {                                 // eight rows in flight; the additions stay in row order
        float u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = U[(int64_t)(uint32_t)sorted[e + q] * d + t];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (double)(u[q] * ib);
    }
    for (; e < e1; ++e) acc += (double)(U[(int64_t)(uint32_t)sorted[e] * d + t] * ib);
    const int64_t o = (int64_t)c * d + t;
    V[o] = first ? (float)acc : (float)(-(double)S.beta[t] * (double)V[o] + acc);
}

// sums in a fixed order: a block = 64 systems x 4 groups; group g adds the items g, g + 4, g + 8, ... (eight loads in flight), the four partial sums are
// combined ((0 + 1) + 2) + 3.  mode 0: sumU[t] = SUM_blocks part[block][t];  mode 1: sumV[t] = SUM_c v[c][t]^2;  mode 2: dk2[t] = SUM_c (w[c][t] / rho)^2
// (V = the W matrix then).  The same bits every time; not the host's one-by-one order.
__global__ __launch_bounds__(256) void lsqr_reduce(const double *__restrict__ part, int64_t nblocks, const float *__restrict__ V, int cols, int d, Scal S,
                                                   int mode) {
    const int t = blockIdx.x * TB + (threadIdx.x & (TB - 1)), g = threadIdx.x >> 6;
    __shared__ double red[4][TB];
    double acc = 0.0;
    if (t < d && !S.done[t]) {
        if (mode == 0) {
            for (int64_t b = g; b < nblocks; b += 4) acc += part[b * d + t];
        } else {
            const float rho = mode == 2 ? S.rho[t] : 1.0f;
            int c = g;
            for (; c + 28 < cols; c += 32) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = V[(int64_t)(c + 4 * q) * d + t];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float x = mode == 2 ? v[q] / rho : v[q]; acc += (double)x * (double)x; }
            }
            for (; c < cols; c += 4) { const float x = mode == 2 ? V[(int64_t)c * d + t] / rho : V[(int64_t)c * d + t]; acc += (double)x * (double)x; }
        }
    }
    red[g][threadIdx.x & (TB - 1)] = acc;
    __syncthreads();
    if (g == 0 && t < d && !S.done[t]) {
        const double tot = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        if (mode == 0) S.sumU[t] = tot; else if (mode == 1) S.sumV[t] = tot; else S.dk2[t] = tot;
    }
}

// v *= ia (when it was updated), x += t1 w, w = t2 w + v;  init: v *= ia, w = v, x = 0   ((w / rho)^2 is summed by lsqr_reduce mode 2 before this pass)
__global__ __launch_bounds__(256) void lsqr_xw_update(float *__restrict__ V, float *__restrict__ W, float *__restrict__ Xs, int cols, int d, Scal S, int init) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t >= d || S.done[t]) return;
    const bool scale = S.beta[t] > 0.0f && S.alpha[t] > 0.0f;
    const float ia = S.ia[t], t1 = S.t1[t], t2 = S.t2[t];
    for (int c = c0; c < c0 + 16 && c < cols; ++c) {
        const int64_t e = (int64_t)c * d + t;
        float v = V[e];
        if (scale) { v = v * ia; V[e] = v; }
        if (init) { W[e] = v; Xs[e] = 0.0f; continue; }
        const float wc = W[e];
        Xs[e] = Xs[e] + t1 * wc;
        W[e] = t2 * wc + v;
    }
}

// scalar steps (one thread per system)
__global__ void lsqr_scal_beta(int d, Scal S, int init) {          // after a u pass: beta, 1 / beta, Anorm; clears sumU
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d || S.done[t]) return;
    const float beta = (float)sqrt(S.sumU[t]);
    S.sumU[t] = 0.0;
    S.beta[t] = beta;
    S.ib[t] = beta > 0.0f ? 1.0f / beta : 0.0f;
    if (!init && beta > 0.0f) {
        const float A = S.Anorm[t], a = S.alpha[t];
        S.Anorm[t] = sqrtf(A * A + a * a + beta * beta);
    }
}

__global__ void lsqr_scal_init(int d, Scal S) {                    // after the first v pass: alpha, the start values, systems with b = 0 or S'b = 0 stop
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d) return;
    const float beta = S.beta[t];
    const float alpha = beta > 0.0f ? (float)sqrt(S.sumV[t]) : 0.0f;
    S.sumV[t] = 0.0;
    S.alpha[t] = alpha;
    S.ia[t] = alpha > 0.0f ? 1.0f / alpha : 0.0f;
    S.rhobar[t] = alpha; S.phibar[t] = beta; S.bnorm[t] = beta;
    S.Anorm[t] = 0.0f; S.ddnorm[t] = 0.0f; S.xnorm[t] = 0.0f; S.xxnorm[t] = 0.0f; S.z[t] = 0.0f; S.sn2[t] = 0.0f; S.cs2[t] = -1.0f;
    S.rho[t] = 1.0f; S.t1[t] = 0.0f; S.t2[t] = 0.0f;
    S.dk2[t] = 0.0;
    // x and w are initialised by the xw pass that follows (it runs for every system: done is set after it, in lsqr_scal_stop0)
}
__global__ void lsqr_scal_stop0(int d, Scal S) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d) return;
    if (S.alpha[t] * S.beta[t] == 0.0f) S.done[t] = 1;
    else atomicAdd(S.active, 1);
}

__global__ void lsqr_scal_rotate(int d, Scal S) {                  // after a v pass: alpha, the plane rotation, the step sizes of the x / w pass
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d || S.done[t]) return;
    const float beta = S.beta[t];
    float alpha = S.alpha[t];
    if (beta > 0.0f) {
        alpha = (float)sqrt(S.sumV[t]);
        S.alpha[t] = alpha;
        S.ia[t] = alpha > 0.0f ? 1.0f / alpha : 0.0f;
    }
    S.sumV[t] = 0.0;
    // plane rotation (damp = 0: rhobar1 = rhobar, cs1 = 1, sn1 = 0, psi = 0)
    const float rhobar1 = S.rhobar[t];
    const float rho = sqrtf(rhobar1 * rhobar1 + beta * beta);
    const float cs = rhobar1 / rho, sn = beta / rho;
    const float theta = sn * alpha;
    S.rhobar[t] = -cs * alpha;
    const float phi = cs * S.phibar[t];
    S.phibar[t] = sn * S.phibar[t];
    S.rho[t] = rho;
    S.t1[t] = phi / rho;
    S.t2[t] = -theta / rho;
    S.phi[t] = phi;
    S.theta[t] = theta;
    S.tau[t] = sn * phi;
}

__global__ void lsqr_scal_stop(int d, Scal S, float atol, float btol, float ctol) {      // after the x / w pass: norm estimates and the stopping rules of Paige & Saunders
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d || S.done[t]) return;
    const float rho = S.rho[t], phi = S.phi[t], theta = S.theta[t], tau = S.tau[t], alpha = S.alpha[t], Anorm = S.Anorm[t], bnorm = S.bnorm[t];
    const float ddnorm = S.ddnorm[t] + (float)S.dk2[t];
    S.ddnorm[t] = ddnorm;
    S.dk2[t] = 0.0;
    const float sn2 = S.sn2[t], cs2 = S.cs2[t], z0 = S.z[t];
    const float delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * z0, zbar = rhs / gambar;
    const float xnorm = sqrtf(S.xxnorm[t] + zbar * zbar);
    const float gamma = sqrtf(gambar * gambar + theta * theta);
    S.cs2[t] = gambar / gamma; S.sn2[t] = theta / gamma;
    const float z = rhs / gamma;
    S.z[t] = z;
    S.xxnorm[t] = S.xxnorm[t] + z * z;
    S.xnorm[t] = xnorm;
    const float Acond = Anorm * sqrtf(ddnorm);
    const float phibar = S.phibar[t];
    const float res1 = phibar * phibar;
    const float rnorm = sqrtf(res1 + 0.0f);                       // res2 stays 0 without damping
    const float Arnorm = alpha * fabsf(tau);
    const float test1 = rnorm / bnorm;
    const float test2 = Arnorm / (Anorm * rnorm);
    const float test3 = 1.0f / Acond;
    const float tt1 = test1 / (1 + Anorm * xnorm / bnorm);
    const float rtol = btol + atol * Anorm * xnorm / bnorm;
    bool stop = (1 + test3 <= 1) || (1 + test2 <= 1) || (1 + tt1 <= 1);
    stop = stop || (test3 <= ctol) || (test2 <= atol) || (test1 <= rtol);
    if (stop) S.done[t] = 1;
    else atomicAdd(S.active, 1);
}

}  // namespace

struct lsq_lsqr_state {
    DevBuf work;      // U, V, W, the per-block norm partials and the per-system scalars of one update
    DevBuf keys;      // (column, row) keys: unsorted, sorted, segment starts, the sort's temporary storage
};

void lsq_lsqr_free(lsq_lsqr_state *st) {
    if (!st) return;
    st->work.release();
    st->keys.release();
    delete st;
}

// dX [n][d], dcodes [n][m] u8 0-based, dK [m*256][d] (output); all device pointers.  iters_out (optional, host): iterations of the slowest system.
int lsq_lsqr_update_codebooks(hipStream_t s, lsq_lsqr_state **pst, const float *dX, const uint8_t *dcodes, int d, int64_t n, int m, float *dK,
                              int *iters_out) {
    if (!*pst) *pst = new lsq_lsqr_state();
    lsq_lsqr_state *st = *pst;
    const int cols = m * LSQ_H;
    const int64_t total = n * (int64_t)m;
    if (total >= (int64_t)1 << 31) { lsq_set_error("lsq_update_codebooks_dev: n * m = %lld exceeds 2^31 - 1", (long long)total); return LSQ_EINVAL; }
    const size_t nd = (size_t)n * d, cd = (size_t)cols * d;
    const int64_t nblocks = (n + 4 * RS - 1) / (4 * RS);
    const size_t f_scal = 20;
    const size_t off_U = 0, off_V = off_U + nd * 4, off_W = off_V + cd * 4, off_part = (off_W + cd * 4 + 15) & ~(size_t)15,
                 off_sc = off_part + (size_t)nblocks * d * 8, off_dbl = (off_sc + f_scal * d * 4 + d * 4 + 15) & ~(size_t)15,
                 wtotal = off_dbl + 3 * (size_t)d * 8 + 64;
    LSQ_TRY(st->work.ensure(wtotal));
    char *base = st->work.as<char>();
    float *U = reinterpret_cast<float *>(base + off_U), *V = reinterpret_cast<float *>(base + off_V), *W = reinterpret_cast<float *>(base + off_W);
    double *part = reinterpret_cast<double *>(base + off_part);
    float *sc = reinterpret_cast<float *>(base + off_sc);
    Scal S;
    float **fields[] = {&S.alpha, &S.beta, &S.rhobar, &S.phibar, &S.Anorm, &S.ddnorm, &S.xnorm, &S.xxnorm, &S.z, &S.sn2, &S.cs2, &S.bnorm, &S.rho, &S.t1,
                        &S.t2, &S.ia, &S.ib, &S.phi, &S.theta, &S.tau};
    static_assert(sizeof(fields) / sizeof(fields[0]) == 20, "scalar slots");
    for (size_t f = 0; f < f_scal; ++f) *fields[f] = sc + f * d;
    S.done = reinterpret_cast<int *>(sc + f_scal * d);
    double *dbl = reinterpret_cast<double *>(base + off_dbl);
    S.sumU = dbl; S.sumV = dbl + d; S.dk2 = dbl + 2 * d;
    S.active = reinterpret_cast<int *>(dbl + 3 * d);
    LSQ_HIP(hipMemsetAsync(base + off_sc, 0, wtotal - off_sc, s));      // scalars, done flags, double sums, the counter

    // the rows sorted by code, once per call: (column << 32 | row) keys, radix sort on their 32 + ceil(log2 cols) bits
    size_t sort_bytes = 0;
    int end_bit = 33;
    while (end_bit < 64 && ((uint64_t)cols >> (end_bit - 32)) != 0) ++end_bit;
    LSQ_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)total, 0, end_bit, s));
    const size_t off_sorted = (size_t)total * 8, off_seg = off_sorted + (size_t)total * 8, off_tmp = (off_seg + ((size_t)cols + 1) * 8 + 255) & ~(size_t)255;
    LSQ_TRY(st->keys.ensure(off_tmp + sort_bytes + 16));
    uint64_t *keys = st->keys.as<uint64_t>(), *sorted = reinterpret_cast<uint64_t *>(st->keys.as<char>() + off_sorted);
    int64_t *seg = reinterpret_cast<int64_t *>(st->keys.as<char>() + off_seg);
    hipLaunchKernelGGL(lsqr_make_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dcodes, n, m, keys);
    LSQ_HIP(hipcub::DeviceRadixSort::SortKeys(st->keys.as<char>() + off_tmp, sort_bytes, keys, sorted, (int)total, 0, end_bit, s));
    hipLaunchKernelGGL(lsqr_segments, dim3((unsigned)((cols + 1 + 255) / 256)), dim3(256), 0, s, sorted, total, cols, seg);

    const dim3 rows_grid((unsigned)nblocks, (unsigned)((d + TB - 1) / TB));
    const dim3 cols_grid((unsigned)((cols + 63) / 64), (unsigned)((d + TB - 1) / TB));
    const dim3 seg_grid((unsigned)cols, (unsigned)((d + TB - 1) / TB));
    const unsigned sgrid = (unsigned)((d + 127) / 128);
    const float tol = sqrtf(1.1920929e-07f);                      // sqrt(eps(Float32)): IterativeSolvers' default atol = btol
    const float ctol = 1.0f / 1e8f;
    const int64_t maxiter = n > cols ? n : cols;

    // start: beta u = b, alpha v = S' u, w = v, x = 0
    hipLaunchKernelGGL(lsqr_u_update, rows_grid, dim3(256), 0, s, dX, U, V, dcodes, n, d, m, S, part, 1);
    hipLaunchKernelGGL(lsqr_reduce, dim3((unsigned)((d + TB - 1) / TB)), dim3(256), 0, s, part, nblocks, V, cols, d, S, 0);
    hipLaunchKernelGGL(lsqr_scal_beta, dim3(sgrid), dim3(128), 0, s, d, S, 1);
    hipLaunchKernelGGL(lsqr_v_update, seg_grid, dim3(TB), 0, s, V, U, sorted, seg, d, S, 1);
    hipLaunchKernelGGL(lsqr_reduce, dim3((unsigned)((d + TB - 1) / TB)), dim3(256), 0, s, part, nblocks, V, cols, d, S, 1);
    hipLaunchKernelGGL(lsqr_scal_init, dim3(sgrid), dim3(128), 0, s, d, S);
    hipLaunchKernelGGL(lsqr_xw_update, cols_grid, dim3(256), 0, s, V, W, dK, cols, d, S, 1);
    hipLaunchKernelGGL(lsqr_scal_stop0, dim3(sgrid), dim3(128), 0, s, d, S);
    int active = 0;
    LSQ_HIP(hipMemcpyAsync(&active, S.active, sizeof(int), hipMemcpyDeviceToHost, s));
    LSQ_HIP(hipStreamSynchronize(s));
    int64_t itn = 0;
    while (active > 0 && itn < maxiter) {
        const int burst = itn < 8 ? 4 : 2;                        // iterations between two looks at the counter (a frozen system costs nothing but its slot)
        for (int b = 0; b < burst && itn < maxiter; ++b, ++itn) {
            LSQ_HIP(hipMemsetAsync(S.active, 0, sizeof(int), s));
            hipLaunchKernelGGL(lsqr_u_update, rows_grid, dim3(256), 0, s, dX, U, V, dcodes, n, d, m, S, part, 0);
            hipLaunchKernelGGL(lsqr_reduce, dim3((unsigned)((d + TB - 1) / TB)), dim3(256), 0, s, part, nblocks, V, cols, d, S, 0);
            hipLaunchKernelGGL(lsqr_scal_beta, dim3(sgrid), dim3(128), 0, s, d, S, 0);
            hipLaunchKernelGGL(lsqr_v_update, seg_grid, dim3(TB), 0, s, V, U, sorted, seg, d, S, 0);
            hipLaunchKernelGGL(lsqr_reduce, dim3((unsigned)((d + TB - 1) / TB)), dim3(256), 0, s, part, nblocks, V, cols, d, S, 1);
            hipLaunchKernelGGL(lsqr_scal_rotate, dim3(sgrid), dim3(128), 0, s, d, S);
            hipLaunchKernelGGL(lsqr_reduce, dim3((unsigned)((d + TB - 1) / TB)), dim3(256), 0, s, part, nblocks, W, cols, d, S, 2);
            hipLaunchKernelGGL(lsqr_xw_update, cols_grid, dim3(256), 0, s, V, W, dK, cols, d, S, 0);
            hipLaunchKernelGGL(lsqr_scal_stop, dim3(sgrid), dim3(128), 0, s, d, S, tol, tol, ctol);
        }
        LSQ_HIP(hipMemcpyAsync(&active, S.active, sizeof(int), hipMemcpyDeviceToHost, s));
        LSQ_HIP(hipStreamSynchronize(s));
    }
    LSQ_HIP(hipGetLastError());
    if (iters_out) *iters_out = (int)itn;
    return LSQ_OK;
}
