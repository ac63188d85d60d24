"""CPU simulation of the energy-identity cost filter with the oracle as the engine.

For every ILS iteration: candidate = oracle worker (perturb + sweeps) on the current codes; exact canonical costs of both; the filter's
quantity F(b) = SUM_j U_j[b_j] + SUM_{j<k} T_jk[b_j][b_k] (f64 sum of the f32 table entries) and its rigorous error bound.  Reports, per iteration,
how many non-identical candidates the filter certifies as `rejected` (canonical new cost > canonical prev cost) and checks that no
certified vector is in fact accepted or equal.

    python tools/sim_cost_filter.py [n] [d] [m] [trained]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle as O          # noqa: E402

U32 = 2.0 ** -24


def gamma(k):
    return k * U32 / (1.0 - k * U32)


def F_of(U, T, codes):
    """U (m, n, h) f32, T (m, m, h, h) f32 [j,k,b,a], codes (n, m) u8 -> f64 (n,)"""
    m, n, h = U.shape
    ar = np.arange(n)
    f = np.zeros(n, dtype=np.float64)
    for j in range(m):
        f += U[j, ar, codes[:, j]].astype(np.float64)
        for k in range(j + 1, m):
            f += T[j, k, codes[:, k], codes[:, j]].astype(np.float64)
    return f


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    trained = len(sys.argv) > 4 and sys.argv[4] == "trained"
    h, I, J, npert = 256, 16, 4, 4
    O.build()
    X = O.synth_data_u8(1234, n, d)
    if d == 960:
        X = (X / np.float32(255.0) * np.float32(0.3)).astype(np.float32)
    rng = np.random.default_rng(0)
    pool = O.synth_data_u8(77, m * h, d)
    if d == 960:
        pool = (pool / np.float32(255.0) * np.float32(0.3)).astype(np.float32)
    K = np.ascontiguousarray(pool / np.float32(m))
    B = O.randinit(42, n, m, h)
    if trained:                                   # a few alternating steps: oracle encode + least-squares codebook update (numpy)
        for it in range(6):
            Bs, _ = O.encode_icm(X, B, K, m, h, [2], J, npert, True, 100 + it)
            B = Bs[0]
            S = np.zeros((n, m * h), dtype=np.float32)
            for j in range(m):
                S[np.arange(n), j * h + B[:, j] - 1] = 1.0
            Kn, *_ = np.linalg.lstsq(S.astype(np.float64), X.astype(np.float64), rcond=None)
            used = S.sum(0) > 0
            K = np.where(used[:, None], Kn, K).astype(np.float32)
            print("train", it, O.qerror(X, B, K, m, h))
        B = O.randinit(42, n, m, h)
    U = O.unaries(X, K, m, h)
    T = O.tables(K, m, h)
    xn = np.sqrt((X.astype(np.float64) ** 2).sum(1))
    cn = np.sqrt((K.astype(np.float64) ** 2).sum(1)).reshape(m, h)
    cmax = cn.max(1)
    # bounds (real arithmetic vs computed): |U - U*| <= gamma_d 2 |x||c| + gamma_d |c|^2 + u |U|,  |T - T*| <= gamma_d 2 |c_j||c_k|
    eps_T = sum(gamma(d) * 2 * cmax[j] * cmax[k] for j in range(m) for k in range(j + 1, m))
    eps_U = lambda norms: sum(gamma(d) * 2 * norms * cmax[j] + gamma(d) * cmax[j] ** 2 for j in range(m))      # + u|U| below
    codes = (B - 1).astype(np.uint8)
    prev = O.veccost(X, K, codes, h)
    Fcur = F_of(U, T, codes)
    tot_cand = tot_rej = tot_cert = 0
    for it in range(I):
        cand = (O.encode_icm_fully(X, codes.astype(np.int16) + 1, K, m, h, J, True, npert, seed=42, it=it) - 1).astype(np.uint8)
        same = (cand == codes).all(1)
        new = O.veccost(X, K, cand, h)
        Fnew = F_of(U, T, cand)
        # canonical-cost error: |c(b) - cost*(b)| <= gamma_(m+10) * (|x| + SUM|c|)^2 (crude)
        eps_C = gamma(m + 12) * (xn + cmax.sum()) ** 2
        absU = sum(np.abs(U[j, np.arange(n), cand[:, j]]) + np.abs(U[j, np.arange(n), codes[:, j]]) for j in range(m)) * U32
        eps = 2 * (eps_U(xn) + eps_T + eps_C) + absU
        cert = (~same) & ((Fnew - Fcur) > eps)
        cacc = (~same) & ((Fnew - Fcur) < -eps)                # certainly accepted (lazy scheme: no exact evaluation either)
        assert not (cacc & ~(new < prev)).any(), "filter certified a rejected vector as accepted"
        und = (~same) & ~cert & ~cacc
        acc = new < prev
        eq = (new == prev) & ~same
        assert not (cert & (acc | eq)).any(), "filter certified an accepted / equal vector"
        ncand = int((~same).sum())
        nrej = int(((~same) & ~acc).sum())
        tot_und = locals().get("tot_und", 0) + int(und.sum())
        print("it %2d  undecided %5.1f%% of all  same %5.1f%%  accepted %5.1f%%  rejected-nonidentical %5.1f%%  certified %5.1f%% of those  (eps median %.1f, |dF| median of rejected %.1f, cost median %.0f)"
              % (it, 100 * und.mean(), 100 * same.mean(), 100 * acc.mean(), 100 * nrej / n, 100 * cert.sum() / max(nrej, 1), np.median(eps),
                 np.median((Fnew - Fcur)[(~same) & ~acc]) if nrej else 0, np.median(prev)))
        tot_cand += ncand
        tot_rej += nrej
        tot_cert += int(cert.sum())
        codes = np.where(acc[:, None], cand, codes)
        Fcur = np.where(acc, Fnew, Fcur)
        prev = np.where(acc, new, prev)
    print("lazy scheme: undecided candidates %d (each costs up to two exact evaluations) + one final pass of %d" % (tot_und, n))
    print("total: candidates evaluated today %d, of which rejected %d, certified by the filter %d (%.1f%% of evaluated)" % (
        tot_cand, tot_rej, tot_cert, 100.0 * tot_cert / max(tot_cand, 1)))


if __name__ == "__main__":
    main()
