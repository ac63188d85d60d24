// lsq_codebook.hip -- codebook update by sparse least squares (HOST code; north_star keeps it on the host).
//
// Replaces update_codebooks(X, B, h) of the reference (src/codebook_update.jl:52-86): for every dimension t,
//     K[t, :] = lsqr(S, X[t, :])        S = sparsify_codes(B, h)  (src/utils.jl:50-69)
// where S is the n x (m*h) matrix with S[i, (j-1)*h + B[j,i]] = 1 (one 1 per codebook and row), solved by
// IterativeSolvers.lsqr with its defaults (x0 = 0, damp = 0, atol = btol = sqrt(eps(Float32)),
// conlim = 1e8, maxiter = max(size(S))).  IterativeSolvers is not vendored nor version-pinned by the
// reference (PARITY UNPINNED); this file restates the published LSQR algorithm of Paige & Saunders
// (ACM TOMS 8(1), 1982) in Float32 with the same stopping rules, and never materialises S:
//     (S v)[i]   = sum_j v[j*h + b_ij]           (gather)
//     (S' u)[c] += u[i] for every code c of row i (scatter)
// Rows of K (dimensions) are independent and are spread over std::thread workers -- the reference spreads
// them over `julia -p` workers the same way (codebook_update.jl:67-79, splitarray(1:d, nworkers())).
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "lsq_internal.h"

namespace {

struct CodeView {
    const int32_t *col;      // [n][m] absolute column j*h + (b_ij - 1)
    int64_t n;
    int m, cols;
};

// Euclidean norm with a double accumulator: at training scale (n ~ 1e5..1e6) a sequential Float32 sum loses sqrt(n) eps .. n eps, which
// reaches the sqrt(eps) stopping tolerance (the reference's IterativeSolvers calls BLAS nrm2, which does not have that problem).
inline float norm2(const std::vector<float> &v) {
    double s = 0.0;
    for (float x : v) s += (double)x * (double)x;
    return (float)std::sqrt(s);
}

// LSQR for one right-hand side b (length n); x (length cols) starts at 0.  Returns the iteration count.
int lsqr_one(const CodeView &A, const std::vector<float> &b, std::vector<float> &x, float atol, float btol, float conlim, int64_t maxiter) {
    const int64_t n = A.n;
    const int m = A.m, cols = A.cols;
    std::fill(x.begin(), x.end(), 0.0f);
    std::vector<float> u(b), v((size_t)cols, 0.0f), w, tmpm((size_t)n);
    std::vector<double> tmpn((size_t)cols);                  // S' u: up to n / h addends per column -> accumulated in double
    const float ctol = conlim > 0 ? 1.0f / conlim : 0.0f;
    float Anorm = 0, Acond = 0, ddnorm = 0, res2 = 0, xnorm = 0, xxnorm = 0, z = 0, sn2 = 0, cs2 = -1;
    float beta = norm2(u), alpha = 0;
    if (beta > 0) {
        const float ib = 1.0f / beta;
        for (auto &e : u) e *= ib;
        std::fill(tmpn.begin(), tmpn.end(), 0.0);
        for (int64_t i = 0; i < n; ++i) { const double ui = u[(size_t)i]; const int32_t *c = A.col + i * m; for (int j = 0; j < m; ++j) tmpn[(size_t)c[j]] += ui; }
        for (int c = 0; c < cols; ++c) v[(size_t)c] = (float)tmpn[(size_t)c];
        alpha = norm2(v);
    }
    if (alpha > 0) { const float ia = 1.0f / alpha; for (auto &e : v) e *= ia; }
    w = v;
    float Arnorm = alpha * beta;
    if (Arnorm == 0) return 0;
    float rhobar = alpha, phibar = beta;
    const float bnorm = beta;
    float rnorm = beta;
    int64_t itn = 0;
    while (itn < maxiter) {
        ++itn;
        // u = S v - alpha u
        for (int64_t i = 0; i < n; ++i) { const int32_t *c = A.col + i * m; float s = 0.0f; for (int j = 0; j < m; ++j) s += v[(size_t)c[j]]; tmpm[(size_t)i] = s; }
        for (int64_t i = 0; i < n; ++i) u[(size_t)i] = -alpha * u[(size_t)i] + tmpm[(size_t)i];
        beta = norm2(u);
        if (beta > 0) {
            const float ib = 1.0f / beta;
            for (auto &e : u) e *= ib;
            Anorm = std::sqrt(Anorm * Anorm + alpha * alpha + beta * beta);
            // v = S' u - beta v
            std::fill(tmpn.begin(), tmpn.end(), 0.0);
            for (int64_t i = 0; i < n; ++i) { const double ui = u[(size_t)i]; const int32_t *c = A.col + i * m; for (int j = 0; j < m; ++j) tmpn[(size_t)c[j]] += ui; }
            for (int c = 0; c < cols; ++c) v[(size_t)c] = (float)(-(double)beta * (double)v[(size_t)c] + tmpn[(size_t)c]);
            alpha = norm2(v);
            if (alpha > 0) { const float ia = 1.0f / alpha; for (auto &e : v) e *= ia; }
        }
        // plane rotation (damp = 0: rhobar1 = rhobar, cs1 = 1, sn1 = 0, psi = 0)
        const float rhobar1 = rhobar;
        const float rho = std::sqrt(rhobar1 * rhobar1 + beta * beta);
        const float cs = rhobar1 / rho, sn = beta / rho;
        const float theta = sn * alpha;
        rhobar = -cs * alpha;
        const float phi = cs * phibar;
        phibar = sn * phibar;
        const float tau = sn * phi;
        const float t1 = phi / rho, t2 = -theta / rho;
        double dk2 = 0.0;
        for (int c = 0; c < cols; ++c) {
            const float wc = w[(size_t)c];
            x[(size_t)c] += t1 * wc;
            const float wr = wc / rho;
            dk2 += (double)wr * (double)wr;
            w[(size_t)c] = t2 * wc + v[(size_t)c];
        }
        ddnorm += (float)dk2;
        // norm estimates and the stopping rules of Paige & Saunders
        const float delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * z, zbar = rhs / gambar;
        xnorm = std::sqrt(xxnorm + zbar * zbar);
        const float gamma = std::sqrt(gambar * gambar + theta * theta);
        cs2 = gambar / gamma; sn2 = theta / gamma; z = rhs / gamma;
        xxnorm += z * z;
        Acond = Anorm * std::sqrt(ddnorm);
        const float res1 = phibar * phibar;
        rnorm = std::sqrt(res1 + res2);
        Arnorm = alpha * std::fabs(tau);
        const float test1 = rnorm / bnorm;
        const float test2 = Arnorm / (Anorm * rnorm);
        const float test3 = 1.0f / Acond;
        const float tt1 = test1 / (1 + Anorm * xnorm / bnorm);
        const float rtol = btol + atol * Anorm * xnorm / bnorm;
        if (1 + test3 <= 1 || 1 + test2 <= 1 || 1 + tt1 <= 1) break;
        if (test3 <= ctol || test2 <= atol || test1 <= rtol) break;
    }
    return (int)itn;
}

// ---- LSMR (Fong & Saunders, SIAM J. Sci. Comput. 33(5), 2011) for one right-hand side: the reference's other choice, codebook_upd_method = "lsmr"
// (src/codebook_update.jl:18-21 -> IterativeSolvers.lsmr, un-vendored and unpinned like lsqr).  Same operator, same Float32 state with the two
// matrix-vector products' long sums in double, lambda = 0, the published stopping rules (atol, btol, conlim).  Returns the iteration count.
inline void sym_ortho(float a, float b, float &c, float &s, float &r) {      // stable Givens rotation
    if (b == 0.0f) { c = a >= 0 ? 1.0f : -1.0f; s = 0.0f; r = std::fabs(a); return; }
    if (a == 0.0f) { c = 0.0f; s = b >= 0 ? 1.0f : -1.0f; r = std::fabs(b); return; }
    if (std::fabs(b) > std::fabs(a)) {
        const float tau = a / b;
        s = (b >= 0 ? 1.0f : -1.0f) / std::sqrt(1.0f + tau * tau);
        c = s * tau;
        r = b / s;
    } else {
        const float tau = b / a;
        c = (a >= 0 ? 1.0f : -1.0f) / std::sqrt(1.0f + tau * tau);
        s = c * tau;
        r = a / c;
    }
}

int lsmr_one(const CodeView &A, const std::vector<float> &b, std::vector<float> &x, float atol, float btol, float conlim, int64_t maxiter) {
    const int64_t n = A.n;
    const int m = A.m, cols = A.cols;
    std::fill(x.begin(), x.end(), 0.0f);
    std::vector<float> u(b), v((size_t)cols, 0.0f), hv, hbar((size_t)cols, 0.0f), tmpm((size_t)n);
    std::vector<double> tmpn((size_t)cols);
    auto At_u = [&]() {      // tmpn = S' u
        std::fill(tmpn.begin(), tmpn.end(), 0.0);
        for (int64_t i = 0; i < n; ++i) { const double ui = u[(size_t)i]; const int32_t *c = A.col + i * m; for (int j = 0; j < m; ++j) tmpn[(size_t)c[j]] += ui; }
    };
    float beta = norm2(u), alpha = 0.0f;
    if (beta > 0) {
        const float ib = 1.0f / beta;
        for (auto &e : u) e *= ib;
        At_u();
        for (int c = 0; c < cols; ++c) v[(size_t)c] = (float)tmpn[(size_t)c];
        alpha = norm2(v);
    }
    if (alpha > 0) { const float ia = 1.0f / alpha; for (auto &e : v) e *= ia; }
    float zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
    hv = v;
    float betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, dsum = 0;
    float normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e30f;
    const float normb = beta, ctol = conlim > 0 ? 1.0f / conlim : 0.0f;
    if (alpha * beta == 0) return 0;
    int64_t itn = 0;
    while (itn < maxiter) {
        ++itn;
        // u = S v - alpha u;  v = S' u - beta v
        for (int64_t i = 0; i < n; ++i) { const int32_t *c = A.col + i * m; float sacc = 0.0f; for (int j = 0; j < m; ++j) sacc += v[(size_t)c[j]]; tmpm[(size_t)i] = sacc; }
        for (int64_t i = 0; i < n; ++i) u[(size_t)i] = -alpha * u[(size_t)i] + tmpm[(size_t)i];
        beta = norm2(u);
        if (beta > 0) {
            const float ib = 1.0f / beta;
            for (auto &e : u) e *= ib;
            At_u();
            for (int c = 0; c < cols; ++c) v[(size_t)c] = (float)(-(double)beta * (double)v[(size_t)c] + tmpn[(size_t)c]);
            alpha = norm2(v);
            if (alpha > 0) { const float ia = 1.0f / alpha; for (auto &e : v) e *= ia; }
        }
        // lambda = 0: the first rotation is the identity (chat = 1, shat = 0, alphahat = alphabar)
        const float alphahat = alphabar, chat = 1.0f, shat = 0.0f;
        const float rhoold = rho;
        float c, sn;
        sym_ortho(alphahat, beta, c, sn, rho);
        const float thetanew = sn * alpha;
        alphabar = c * alpha;
        const float rhobarold = rhobar, zetaold = zeta;
        const float thetabar = sbar * rho, rhotemp = cbar * rho;
        sym_ortho(cbar * rho, thetanew, cbar, sbar, rhobar);
        zeta = cbar * zetabar;
        zetabar = -sbar * zetabar;
        const float f1 = thetabar * rho / (rhoold * rhobarold), f2 = zeta / (rho * rhobar), f3 = thetanew / rho;
        for (int cc = 0; cc < cols; ++cc) {
            const float hb = hv[(size_t)cc] - f1 * hbar[(size_t)cc];
            hbar[(size_t)cc] = hb;
            x[(size_t)cc] += f2 * hb;
            hv[(size_t)cc] = v[(size_t)cc] - f3 * hv[(size_t)cc];
        }
        // estimate of ||r||
        const float betaacute = chat * betadd, betacheck = -shat * betadd;
        const float betahat = c * betaacute;
        betadd = -sn * betaacute;
        const float thetatildeold = thetatilde;
        float ctildeold, stildeold, rhotildeold;
        sym_ortho(rhodold, thetabar, ctildeold, stildeold, rhotildeold);
        thetatilde = stildeold * rhobar;
        rhodold = ctildeold * rhobar;
        betad = -stildeold * betad + ctildeold * betahat;
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
        const float taud = (zeta - thetatilde * tautildeold) / rhodold;
        dsum += betacheck * betacheck;
        const float normr = std::sqrt(dsum + (betad - taud) * (betad - taud) + betadd * betadd);
        normA2 += beta * beta;
        const float normA = std::sqrt(normA2);
        normA2 += alpha * alpha;
        maxrbar = std::max(maxrbar, rhobarold);
        if (itn > 1) minrbar = std::min(minrbar, rhobarold);
        const float condA = std::max(maxrbar, rhotemp) / std::min(minrbar, rhotemp);
        const float normar = std::fabs(zetabar), normx = norm2(x);
        const float test1 = normr / normb;
        const float test2 = (normA * normr != 0) ? normar / (normA * normr) : INFINITY;
        const float test3 = 1.0f / condA;
        const float t1 = test1 / (1 + normA * normx / normb);
        const float rtol = btol + atol * normA * normx / normb;
        if (1 + test3 <= 1 || 1 + test2 <= 1 || 1 + t1 <= 1) break;
        if (test3 <= ctol || test2 <= atol || test1 <= rtol) break;
    }
    return (int)itn;
}

}  // namespace

static int update_codebooks_host(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K, bool lsmr);

extern "C" int lsq_update_codebooks(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K) {
    return update_codebooks_host(X, B, d, n, m, h, nthreads, K, false);
}

// codebook_upd_method = "lsmr" (src/codebook_update.jl:18-21)
extern "C" int lsq_update_codebooks_lsmr(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K) {
    return update_codebooks_host(X, B, d, n, m, h, nthreads, K, true);
}

static int update_codebooks_host(const float *X, const int16_t *B, int d, int64_t n, int m, int h, int nthreads, float *K, bool lsmr) {
    if (d < 1 || n < 1 || m < 1 || h < 1 || !X || !B || !K) { lsq_set_error("lsq_update_codebooks: bad arguments"); return LSQ_EINVAL; }
    const int cols = m * h;
    std::vector<int32_t> col((size_t)n * m);
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            const int b = B[i * m + j];
            if (b < 1 || b > h) { lsq_set_error("lsq_update_codebooks: code %d outside 1..%d", b, h); return LSQ_ECODE; }
            col[(size_t)(i * m + j)] = j * h + (b - 1);
        }
    const CodeView A{col.data(), n, m, cols};
    const float tol = std::sqrt(1.1920929e-07f);             // sqrt(eps(Float32)): IterativeSolvers' default atol = btol
    const int64_t maxiter = n > cols ? n : cols;
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > d) nt = d;
    auto work = [&](int t0, int t1) {
        std::vector<float> b((size_t)n), x((size_t)cols);
        for (int t = t0; t < t1; ++t) {
            for (int64_t i = 0; i < n; ++i) b[(size_t)i] = X[i * d + t];
            if (lsmr) lsmr_one(A, b, x, tol, tol, 1e8f, maxiter);
            else lsqr_one(A, b, x, tol, tol, 1e8f, maxiter);
            for (int c = 0; c < cols; ++c) K[(size_t)c * d + t] = x[(size_t)c];
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, (int)((int64_t)d * t / nt), (int)((int64_t)d * (t + 1) / nt));
    for (auto &th : pool) th.join();
    return LSQ_OK;
}
