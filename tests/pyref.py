"""Independent pure-numpy restatement of the ILS/ICM control flow (small cases only).

Written directly from the reference sources (src/encodings/encode_icm.jl:55-125,131-189;
src/utils.jl:225-254), NOT from oracle/lsq_oracle.c, so that the C oracle is cross-checked by a
second implementation.  It takes the tables (U, T) as inputs -- the fmaf-chain contraction is
checked separately -- and shares only the Philox word stream with the oracle.
"""
import numpy as np

f32 = np.float32


def node_update(uj, T, code, j):
    """encode_icm.jl:76-119 for one vector: copy, absorb in ascending k, first argmin."""
    m = T.shape[0]
    s = uj.astype(f32).copy()
    for k in range(m):
        if k == j:
            continue
        s = (s + T[j, k, code[k]]).astype(f32)          # plain f32 adds, ascending k
    return int(np.argmin(s))                             # np.argmin returns the first minimum


def cost(x, K, code, h):
    """utils.jl:238-249 with the build-defined reduction order (64 strided partials + pairwise tree)."""
    m = len(code)
    d = x.shape[0]
    cb = np.zeros(d, dtype=f32)
    for k in range(m):
        cb = (cb + K[k * h + code[k]]).astype(f32)
    r = (cb - x).astype(f32)
    sq = (r * r).astype(f32)
    p = np.zeros(64, dtype=f32)
    for t in range(d):
        p[t & 63] = f32(p[t & 63] + sq[t])
    s = 1
    while s < 64:
        for l in range(0, 64, 2 * s):
            p[l] = f32(p[l] + p[l + s])
        s *= 2
    return p[0]


def perturb(rng_word, seed, gidx, it, code, h, npert):
    """encode_icm.jl:55-70 with the selection-sampling scan of cudautils.cu:48-70 in integer form."""
    m = len(code)
    code = code.copy()
    need = min(npert, m)
    for p in range(m):
        if need == 0:
            break
        r = rng_word(seed, gidx, it, 1, p)
        if (r * (m - p)) >> 32 < need:
            rv = rng_word(seed, gidx, it, 1, 16 + p)
            code[p] = (rv * h) >> 32
            need -= 1
    return code


def perm(rng_word, seed, it, m, randord):
    order = list(range(m))
    if randord:
        for p in range(m - 1, 0, -1):
            r = rng_word(seed, 0, it, 2, m - 1 - p)
            q = (r * (p + 1)) >> 32
            order[p], order[q] = order[q], order[p]
    return order


def encode(rng_word, X, B0, K, U, T, h, ilsiters, icmiter, npert, randord, seed, global_offset=0):
    """-> Bs (nr, n, m) int16 1-based, objs (nr,), stats (I, 2)."""
    n, d = X.shape
    m = B0.shape[1]
    I = max(ilsiters)
    cur = (B0 - 1).astype(np.int64)
    prev = np.array([cost(X[i], K, cur[i], h) for i in range(n)], dtype=f32)
    Bs = np.zeros((len(ilsiters), n, m), dtype=np.int16)
    objs = np.zeros(len(ilsiters), dtype=np.float64)
    stats = np.zeros((I, 2), dtype=np.int64)
    for it in range(I):
        order = perm(rng_word, seed, it, m, randord)
        for i in range(n):
            nw = perturb(rng_word, seed, global_offset + i, it, cur[i], h, npert)
            for _ in range(icmiter):
                for j in order:
                    nw[j] = node_update(U[j, i], T, nw, j)
            c = cost(X[i], K, nw, h)
            stats[it, 0] += int(c == prev[i])
            if c < prev[i]:                              # strict improvement only (encode_icm.jl:183-186)
                stats[it, 1] += 1
                cur[i] = nw
                prev[i] = c
        for r, target in enumerate(ilsiters):
            if target == it + 1:
                Bs[r] = (cur + 1).astype(np.int16)
                objs[r] = prev.astype(np.float64).sum() / n
    return Bs, objs.astype(np.float32), stats
