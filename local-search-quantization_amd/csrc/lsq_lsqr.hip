// lsq_lsqr.hip -- the codebook update ON THE DEVICE: all d least-squares problems of update_codebooks at once.
//
// Reference: update_codebooks(X, B, h) (src/codebook_update.jl:52-86): K[t, :] = lsqr(S, X[t, :]) for every dimension t, S = sparsify_codes(B, h)
// (src/utils.jl:50-69) the n x (m h) matrix with one 1 per codebook and row; IterativeSolvers.lsqr defaults (x0 = 0, damp = 0,
// atol = btol = sqrt(eps(Float32)), conlim = 1e8, maxiter = max(size(S))) -- un-vendored and un-pinned by the reference (PARITY UNPINNED).
// This is the SAME restatement of Paige & Saunders' LSQR as the host code (lsq_codebook.hip: Float32 recurrences, the three long sums -- ||u||, S'u,
// ||v|| -- accumulated in double, the same stopping rules), with the d right-hand sides advanced together: the vectors of a system are the
// columns of [n][d] / [m h][d] matrices (the layout of X and K), its scalars live in arrays of d, and a system that has met its stopping rule
// freezes while the others go on.  S is never materialised:
//     (S v)[i]    = sum_j v[j h + b_ij]             gather, codebooks ascending
//     (S' u)[c]  += u[i] for every code c of row i   double atomics (the order of the addends differs from the host's sequential loop: the double
//                                                    sums agree to ~1e-16 and their Float32 roundings almost always bit for bit -- tests compare
//                                                    the two solvers to 1e-5 and both to scipy)
// One iteration = four passes over U / V (HBM-bound: 2 x n d + 4 x m h d floats) and three d-thread scalar kernels.
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

// u = b; sumU = sum b^2
__global__ __launch_bounds__(256) void lsqr_init_u(const float *__restrict__ X, float *__restrict__ U, int64_t n, int d, double *__restrict__ sumU) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RS;
    if (t >= d) return;
    double s = 0.0;
    for (int64_t i = r0; i < r0 + RS && i < n; ++i) { const float b = X[i * d + t]; U[i * d + t] = b; s += (double)b * (double)b; }
    if (s != 0.0) atomicAdd(&sumU[t], s);
}

// u *= ib (when beta > 0), tmpn[c][t] += u   for every code c of the row
__global__ __launch_bounds__(256) void lsqr_scale_scatter(float *__restrict__ U, const uint8_t *__restrict__ codes, int64_t n, int d, int m, Scal S,
                                                          double *__restrict__ tmpn) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RS;
    if (t >= d || S.done[t] || !(S.beta[t] > 0.0f)) return;
    const float ib = S.ib[t];
    for (int64_t i = r0; i < r0 + RS && i < n; ++i) {
        const float u = U[i * d + t] * ib;
        U[i * d + t] = u;
        const uint8_t *c = codes + i * m;
        for (int j = 0; j < m; ++j) atomicAdd(&tmpn[((int64_t)j * LSQ_H + c[j]) * d + t], (double)u);
    }
}

// first pass: v = (float)tmpn, sumV;  later passes: v = (float)(-beta v + tmpn), sumV
__global__ __launch_bounds__(256) void lsqr_v_update(float *__restrict__ V, const double *__restrict__ tmpn, int cols, int d, Scal S, int first) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t >= d || S.done[t] || !(S.beta[t] > 0.0f)) return;
    const double beta = (double)S.beta[t];
    double s = 0.0;
    for (int c = c0; c < c0 + 16 && c < cols; ++c) {
        const float v = first ? (float)tmpn[(int64_t)c * d + t] : (float)(-beta * (double)V[(int64_t)c * d + t] + tmpn[(int64_t)c * d + t]);
        V[(int64_t)c * d + t] = v;
        s += (double)v * (double)v;
    }
    if (s != 0.0) atomicAdd(&S.sumV[t], s);
}

// u = S v - alpha u, sumU
__global__ __launch_bounds__(256) void lsqr_u_update(float *__restrict__ U, const float *__restrict__ V, const uint8_t *__restrict__ codes, int64_t n, int d,
                                                     int m, Scal S) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RS;
    if (t >= d || S.done[t]) return;
    const float alpha = S.alpha[t];
    double acc = 0.0;
    for (int64_t i = r0; i < r0 + RS && i < n; ++i) {
        const uint8_t *c = codes + i * m;
        float s = 0.0f;
        for (int j = 0; j < m; ++j) s += V[((int64_t)j * LSQ_H + c[j]) * d + t];
        const float u = -alpha * U[i * d + t] + s;
        U[i * d + t] = u;
        acc += (double)u * (double)u;
    }
    if (acc != 0.0) atomicAdd(&S.sumU[t], acc);
}

// v *= ia (when it was updated), x += t1 w, dk2 += (w / rho)^2, w = t2 w + v;  init: v *= ia, w = v, x = 0
__global__ __launch_bounds__(256) void lsqr_xw_update(float *__restrict__ V, float *__restrict__ W, float *__restrict__ Xs, int cols, int d, Scal S, int init) {
    const int t = blockIdx.y * TB + (threadIdx.x & (TB - 1));
    const int c0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t >= d || S.done[t]) return;
    const bool scale = S.beta[t] > 0.0f && S.alpha[t] > 0.0f;
    const float ia = S.ia[t], t1 = S.t1[t], t2 = S.t2[t], rho = S.rho[t];
    double dk = 0.0;
    for (int c = c0; c < c0 + 16 && c < cols; ++c) {
        const int64_t e = (int64_t)c * d + t;
        float v = V[e];
        if (scale) { v = v * ia; V[e] = v; }
        if (init) { W[e] = v; Xs[e] = 0.0f; continue; }
        const float wc = W[e];
        Xs[e] = Xs[e] + t1 * wc;
        const float wr = wc / rho;
        dk += (double)wr * (double)wr;
        W[e] = t2 * wc + v;
    }
    if (!init && dk != 0.0) atomicAdd(&S.dk2[t], dk);
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
    DevBuf work;      // U, V, W, the double column sums and the per-system scalars of one update
};

void lsq_lsqr_free(lsq_lsqr_state *st) {
    if (!st) return;
    st->work.release();
    delete st;
}

// dX [n][d], dcodes [n][m] u8 0-based, dK [m*256][d] (output); all device pointers.  iters_out (optional, host): iterations of the slowest system.
int lsq_lsqr_update_codebooks(hipStream_t s, lsq_lsqr_state **pst, const float *dX, const uint8_t *dcodes, int d, int64_t n, int m, float *dK,
                              int *iters_out) {
    if (!*pst) *pst = new lsq_lsqr_state();
    lsq_lsqr_state *st = *pst;
    const int cols = m * LSQ_H;
    const size_t nd = (size_t)n * d, cd = (size_t)cols * d;
    const size_t f_scal = 20, off_U = 0, off_V = off_U + nd * 4, off_W = off_V + cd * 4, off_tmpn = (off_W + cd * 4 + 15) & ~(size_t)15,
                 off_sc = off_tmpn + cd * 8, off_dbl = (off_sc + f_scal * d * 4 + d * 4 + 15) & ~(size_t)15, total = off_dbl + 3 * (size_t)d * 8 + 64;
    LSQ_TRY(st->work.ensure(total));
    char *base = st->work.as<char>();
    float *U = reinterpret_cast<float *>(base + off_U), *V = reinterpret_cast<float *>(base + off_V), *W = reinterpret_cast<float *>(base + off_W);
    double *tmpn = reinterpret_cast<double *>(base + off_tmpn);
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
    LSQ_HIP(hipMemsetAsync(base + off_sc, 0, total - off_sc, s));      // scalars, done flags, double sums, the counter
    LSQ_HIP(hipMemsetAsync(tmpn, 0, cd * 8, s));

    const dim3 rows_grid((unsigned)((n + 4 * RS - 1) / (4 * RS)), (unsigned)((d + TB - 1) / TB));
    const dim3 cols_grid((unsigned)((cols + 63) / 64), (unsigned)((d + TB - 1) / TB));
    const unsigned sgrid = (unsigned)((d + 127) / 128);
    const float tol = sqrtf(1.1920929e-07f);                      // sqrt(eps(Float32)): IterativeSolvers' default atol = btol
    const float ctol = 1.0f / 1e8f;
    const int64_t maxiter = n > cols ? n : cols;

    // start: beta u = b, alpha v = S' u, w = v, x = 0
    hipLaunchKernelGGL(lsqr_init_u, rows_grid, dim3(256), 0, s, dX, U, n, d, S.sumU);
    hipLaunchKernelGGL(lsqr_scal_beta, dim3(sgrid), dim3(128), 0, s, d, S, 1);
    hipLaunchKernelGGL(lsqr_scale_scatter, rows_grid, dim3(256), 0, s, U, dcodes, n, d, m, S, tmpn);
    hipLaunchKernelGGL(lsqr_v_update, cols_grid, dim3(256), 0, s, V, tmpn, cols, d, S, 1);
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
            hipLaunchKernelGGL(lsqr_u_update, rows_grid, dim3(256), 0, s, U, V, dcodes, n, d, m, S);
            hipLaunchKernelGGL(lsqr_scal_beta, dim3(sgrid), dim3(128), 0, s, d, S, 0);
            LSQ_HIP(hipMemsetAsync(tmpn, 0, cd * 8, s));
            hipLaunchKernelGGL(lsqr_scale_scatter, rows_grid, dim3(256), 0, s, U, dcodes, n, d, m, S, tmpn);
            hipLaunchKernelGGL(lsqr_v_update, cols_grid, dim3(256), 0, s, V, tmpn, cols, d, S, 0);
            hipLaunchKernelGGL(lsqr_scal_rotate, dim3(sgrid), dim3(128), 0, s, d, S);
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
