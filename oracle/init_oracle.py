"""oracle/init_oracle.py -- CHECKER of the initialisers: PQ, OPQ and ChainQ restated in numpy.  TEST INFRASTRUCTURE: only tests/ may
import it; the product (local-search-quantization_amd/initializers.py) runs its two data-parallel steps -- the nearest-codeword assignment and the
chain's Viterbi encoder -- as HIP kernels (csrc/lsq_init.hip) and never touches this file.

    C, B, R, err = train_opq(x_train, m, h, niter, "natural")        # src/opq/OPQ.jl:21-101
    C, B, R, err = train_chainq(x_train, m, h, R, B, C, niter)       # src/chainq/chainq.jl:10-58

Two layers:
  * `assign_codewords_exact` / `encoding_viterbi_exact`: the arithmetic the device kernels are held to, BIT FOR BIT -- the oracle's own unaries and
    pair tables (oracle/lsq_oracle.c: k-ascending fmaf chains, the frozen choice (1) of DESIGN 2), then first-minimum scans and plain f32 adds in the
    reference's order (encode_chain.jl:37-83; PQ.jl:12-41).  h = 256 (what the C restatement and the engine support).
  * the trainers and their general-h numpy steps (`kmeans`, `train_pq`, `train_opq`, `train_chainq`, ...): the algorithms, property-tested on the CPU
    with small h; with `exact=True` they take the layer above for their assignment / Viterbi steps, which is how the GPU tests compare whole training
    runs of the product with this checker.
PARITY UNPINNED: the reference delegates to Clustering.jl k-means, StatsBase sampling, Distances.jl and IterativeSolvers LSQR, none vendored or
version-pinned, and has no tests for them; what is mirrored is the algorithm and the interfaces.
Shapes follow Julia: X is d x n, codes B are m x n Int16 **1-based**, codebooks are lists of (rows x h) matrices.
"""
import numpy as np

from . import oracle as O


def splitarray(n, nparts):
    """utils.jl:152-177 -> list of (start, stop) 0-based half-open ranges (the first n mod nparts one longer)."""
    per, xtra = divmod(n, nparts)
    out, s = [], 0
    for p in range(nparts):
        ln = per + 1 if p < xtra else per
        out.append((s, s + ln))
        s += ln
    return out


def stack_codebooks(C, d=None, dims=None):
    """list of (rows x h) codebooks -> K (m h, d) = hcat(C...) row-major; `dims` (slices): codebook i lives in rows dims[i] of a d-row zero matrix."""
    if dims is None:
        return np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.float32).T for c in C], axis=0))
    h = np.asarray(C[0]).shape[1]
    K = np.zeros((len(C) * h, d), dtype=np.float32)
    for i, c in enumerate(C):
        K[i * h:(i + 1) * h, dims[i]] = np.asarray(c, dtype=np.float32).T
    return K


def assign_codewords_exact(X, K, m, h=256):
    """X (n, d), K (m h, d) -> codes (n, m) int64 0-based, minima (n, m) f32: per codebook the first argmin of the oracle's unaries."""
    U = O.unaries(np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(K, dtype=np.float32), m, h)      # (m, n, h)
    a = U.argmin(axis=2)                                                                                            # first minimum
    return a.T.copy(), np.take_along_axis(U, a[:, :, None], axis=2)[:, :, 0].T.copy()


def encoding_viterbi_exact(X, K, m, h=256, block=128):
    """X (n, d), K (m h, d) -> codes (n, m) int64 0-based: encode_chain.jl:2-89 on the oracle's unaries and pair tables."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    K = np.ascontiguousarray(K, dtype=np.float32)
    n = X.shape[0]
    T = O.tables(K, m, h)                                        # T[j, k, b, a] = 2 <c_kb, c_ja>
    bins = [T[i + 1, i] for i in range(m - 1)]                   # bb_i[k (source, codebook i)][j (target, codebook i + 1)]
    out = np.zeros((n, m), dtype=np.int64)
    for lo in range(0, n, block):
        U = O.unaries(X[lo:lo + block], K, m, h)                 # (m, nb, h)
        nb = U.shape[1]
        back = np.zeros((m - 1, nb, h), dtype=np.int64)
        acc = U[0]
        for i in range(m - 1):
            cost = acc[:, :, None] + bins[i][None, :, :]         # f32: (nb, from k, to j)   encode_chain.jl:52-54
            back[i] = cost.argmin(axis=1)                        # first minimum over k       :58-66
            acc = U[i + 1] + np.take_along_axis(cost, back[i][:, None, :], axis=1)[:, 0, :]   # :41-45, :70-72
        path = acc.argmin(axis=1)                                # :74
        out[lo:lo + nb, m - 1] = path
        for i in range(m - 2, -1, -1):                           # :77-80
            path = back[i][np.arange(nb), path]
            out[lo:lo + nb, i] = path
    return out


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _subdims(d, m):
    """splitarray(1:d, m) as 0-based slices (src/utils.jl:152-177)."""
    return [slice(lo, hi) for lo, hi in splitarray(d, m)]


def _assign_exact(C, X):
    """One sub-space, h = 256: the device kernel's arithmetic (assign_codewords_exact with m = 1); costs = minimum + ||x||^2."""
    a, mv = assign_codewords_exact(_f32(X).T, _f32(C).T, 1, C.shape[1])
    return a[:, 0], (mv[:, 0] + np.einsum("ij,ij->j", _f32(X), _f32(X))).astype(np.float32)


def _sqdist(C, X):
    """pairwise SqEuclidean: (h, n) distances between the columns of C (r x h) and X (r x n)."""
    cc = np.einsum("ij,ij->j", C, C)[:, None]
    xx = np.einsum("ij,ij->j", X, X)[None, :]
    return np.maximum(cc + xx - 2.0 * (C.T @ X), 0.0).astype(np.float32)


def _assign_scalar(c, x):
    """r = 1 (the norm codebook of train_lsq: n scalars, h centres): argmin_k (x - c_k)^2 evaluated in f32, lowest index on ties -- the result of the
    brute-force h x n scan, in O(n log h): the f32 distance is monotone in |x - c_k|, so the minimisers form a contiguous run of the SORTED centres
    around x; the run is found from the two neighbours of x and widened while the distance stays equal.  (The pairwise expansion of _sqdist
    cancels catastrophically for scalars of the size of squared norms; this is also the better-conditioned formula.)
    Behaviour note: EVERY width-1 k-means goes through here, i.e. also a PQ / OPQ sub-space of width 1 (d == m) -- its assignments are those of
    (x - c)^2, which can differ from the expanded-distance scan of wider sub-spaces on near-ties; the reference (Clustering.jl) is unpinned either way."""
    c = np.asarray(c, dtype=np.float32).reshape(-1)
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    h, n = c.shape[0], x.shape[0]
    order = np.argsort(c, kind="stable")
    cs = c[order]
    pos = np.searchsorted(cs, x)                                   # cs[pos - 1] < x <= cs[pos]
    L = np.clip(pos - 1, 0, h - 1)
    R = np.clip(pos, 0, h - 1)
    dL = (x - cs[L]) ** 2
    dR = (x - cs[R]) ** 2
    best = np.minimum(dL, dR)
    lo = np.where(dL <= dR, L, R)                                  # one member of the run of minimisers
    hi = lo.copy()
    idx = order[lo].copy()                                         # lowest ORIGINAL index seen in the run so far
    other = np.where(dL <= dR, R, L)
    tie = (dL == dR) & (L != R)
    idx[tie] = np.minimum(idx[tie], order[other[tie]])
    lo = np.where(tie, np.minimum(lo, other), lo)
    hi = np.where(tie, np.maximum(hi, other), hi)
    live = np.arange(n)
    while live.size:                                               # widen the run while the next centre on either side is exactly as far (duplicates, rounding)
        l2, h2 = lo[live] - 1, hi[live] + 1
        okl = l2 >= 0
        okh = h2 < h
        el = np.zeros(live.size, dtype=bool)
        eh = np.zeros(live.size, dtype=bool)
        el[okl] = (x[live[okl]] - cs[l2[okl]]) ** 2 == best[live[okl]]
        eh[okh] = (x[live[okh]] - cs[h2[okh]]) ** 2 == best[live[okh]]
        if el.any():
            t = live[el]
            lo[t] -= 1
            idx[t] = np.minimum(idx[t], order[lo[t]])
        if eh.any():
            t = live[eh]
            hi[t] += 1
            idx[t] = np.minimum(idx[t], order[hi[t]])
        live = live[el | eh]
    return idx.astype(np.int64), best.astype(np.float32)


def _assign(C, X, exact=False):
    """Nearest codeword per column of X, lowest index on ties (update_assignments!, src/opq/kmeans.jl:6-75).  exact: the device kernel's arithmetic."""
    if X.shape[0] == 1:
        return _assign_scalar(C, X)
    if exact:
        return _assign_exact(C, X)
    dm = _sqdist(C, X)
    a = dm.argmin(axis=0)
    return a, dm[a, np.arange(X.shape[1])]


def _centers(X, a, h, rng, old=None):
    """Cluster means (update_centers!, src/opq/kmeans.jl:77-123); an empty cluster is re-seeded with a random point."""
    r, n = X.shape
    C = np.zeros((r, h), dtype=np.float32)
    cnt = np.bincount(a, minlength=h)
    nz = cnt > 0
    if r == 1:                                           # n scalars (the norm codebook): one weighted bincount, sums in f64
        C[0] = np.bincount(a, weights=X[0].astype(np.float64), minlength=h).astype(np.float64)[:h] / np.maximum(cnt, 1)
    else:
        np.add.at(C.T, a, X.T)
        C[:, nz] /= cnt[nz]
    for k in np.nonzero(~nz)[0]:
        C[:, k] = X[:, rng.integers(n)] if old is None else old[:, k]
    return C


def kmeans(X, h, niter=25, seed=0, exact=False):
    """Lloyd k-means with k-means++ seeding on the columns of X (r x n) -> centers (r x h), assignments (n,) 0-based, total cost.
    Stands in for Clustering.jl's `kmeans(X, h, init=:kmpp)` (src/pq/PQ.jl:60)."""
    X = _f32(X)
    r, n = X.shape
    rng = np.random.default_rng(seed)
    C = np.empty((r, h), dtype=np.float32)
    C[:, 0] = X[:, rng.integers(n)]
    d2 = ((X - C[:, :1]) ** 2).sum(axis=0)
    for k in range(1, h):
        tot = float(d2.sum())
        idx = rng.integers(n) if tot <= 0 else int(np.searchsorted(np.cumsum(d2), rng.random() * tot))
        C[:, k] = X[:, min(idx, n - 1)]
        d2 = np.minimum(d2, ((X - C[:, k:k + 1]) ** 2).sum(axis=0))
    a, cost = _assign(C, X, exact)
    for _ in range(niter):
        C = _centers(X, a, h, rng)
        a2, cost = _assign(C, X, exact)
        if np.array_equal(a2, a):
            break
        a = a2
    return C, a, float(cost.sum())


# ---- PQ (src/pq/PQ.jl) ---------------------------------------------------------------------------------------
def quantize_pq(X, C, V=False, exact=False):
    """quantize_pq(X, C) -> B (m x n Int16, 1-based).  src/pq/PQ.jl:12-41"""
    X = _f32(X)
    sd = _subdims(X.shape[0], len(C))
    return np.stack([_assign(_f32(C[i]), X[sd[i]], exact)[0] + 1 for i in range(len(C))]).astype(np.int16)


def qerror_pq(X, B, C):
    """Mean squared reconstruction error of PQ codes (codebooks hold sub-vectors)."""
    X = _f32(X)
    sd = _subdims(X.shape[0], len(C))
    err = 0.0
    for i in range(len(C)):
        err += float(((X[sd[i]] - _f32(C[i])[:, np.asarray(B[i], dtype=np.int64) - 1]) ** 2).sum())
    return err / X.shape[1]


def train_pq(X, m, h, V=False, *, seed=0, exact=False):
    """train_pq(X, m, h) -> C, B, error.  src/pq/PQ.jl:44-76 (k-means per subspace)."""
    X = _f32(X)
    sd = _subdims(X.shape[0], m)
    C, B = [], []
    for i in range(m):
        c, a, cost = kmeans(X[sd[i]], h, seed=seed + i, exact=exact)
        C.append(c)
        B.append(a + 1)
        if V:
            print("codebook %d / %d: error in subspace %e" % (i + 1, m, cost / X.shape[1]))
    B = np.stack(B).astype(np.int16)
    return C, B, qerror_pq(X, B, C)


# ---- OPQ (src/opq/OPQ.jl) ------------------------------------------------------------------------------------
def quantize_opq(X, R, C, V=False, exact=False):
    """quantize_opq(X, R, C) = quantize_pq(R'X, C).  src/opq/OPQ.jl:10-19"""
    return quantize_pq(_f32(R).T @ _f32(X), C, V, exact)


def _procrustes(X, CB):
    """R = U V' with U S V' = svd(X CB')  (src/opq/OPQ.jl:78-79, src/chainq/chainq.jl:44-45)."""
    U, _, Vt = np.linalg.svd(X.astype(np.float64) @ CB.astype(np.float64).T, full_matrices=False)
    return (U @ Vt).astype(np.float32)


def train_opq(X, m, h, niter, init="natural", V=False, *, seed=0, exact=False):
    """train_opq(X, m, h, niter, init) -> C, B, R, obj.  src/opq/OPQ.jl:21-101
    C[i]: (subdim x h) codebooks of the rotated space, B: m x n Int16 1-based, R: d x d, obj: niter+1 errors."""
    X = _f32(X)
    d, n = X.shape
    rng = np.random.default_rng(seed)
    if init == "natural":
        R = np.eye(d, dtype=np.float32)
    elif init == "random":
        R = np.linalg.svd(rng.standard_normal((d, d)))[0].astype(np.float32)
    else:
        raise ValueError("Intialization %s unknown" % init)
    RX = R.T @ X
    sd = _subdims(d, m)
    C = [RX[sd[i]][:, rng.choice(n, h, replace=False)].copy() for i in range(m)]     # :46-50
    B = np.zeros((m, n), dtype=np.int64)
    CB = np.zeros_like(X)
    for i in range(m):
        B[i], _ = _assign(C[i], RX[sd[i]], exact)
        CB[sd[i]] = C[i][:, B[i]]
    obj = np.zeros(niter + 1, dtype=np.float32)
    for it in range(niter + 1):
        obj[it] = float(((R @ CB - X) ** 2).sum()) / n
        if V:
            print("%3d %e" % (it, obj[it]))
        R = _procrustes(X, CB)
        RX = R.T @ X
        for i in range(m):
            C[i] = _centers(RX[sd[i]], B[i], h, rng, old=C[i])
            B[i], _ = _assign(C[i], RX[sd[i]], exact)
            CB[sd[i]] = C[i][:, B[i]]
    return C, (B + 1).astype(np.int16), R, obj


# ---- ChainQ (src/chainq/chainq.jl, src/encodings/encode_chain.jl, src/codebook_update.jl:88-169) -------------------
def get_cbdims_chain(d, m):
    """Dimensions each codebook of a chain covers: consecutive codebooks overlap on one of the m-1 blocks.
    src/codebook_update.jl:88-102.  Returns m 0-based slices."""
    sub = splitarray(d, m - 1)
    od = [slice(sub[0][0], sub[0][1])]
    for i in range(1, m - 1):
        od.append(slice(sub[i - 1][0], sub[i][1]))
    od.append(slice(sub[-1][0], sub[-1][1]))
    return od


def update_codebooks_chain(X, B, h, V=False):
    """Least-squares codebooks under the chain's dimension structure: for every dimension t only the codebooks that
    cover t are fitted (LSQR on the corresponding columns of the one-hot code matrix).  src/codebook_update.jl:104-169"""
    from scipy.sparse import csr_matrix
    from scipy.sparse.linalg import lsqr
    X = _f32(X)
    d, n = X.shape
    B = np.asarray(B, dtype=np.int64)
    m = B.shape[0]
    od = get_cbdims_chain(d, m)
    cols = (B - 1 + (np.arange(m) * h)[:, None]).T.reshape(-1)                     # sparsify_codes, src/utils.jl:50-69
    S = csr_matrix((np.ones(n * m, dtype=np.float32), (np.repeat(np.arange(n), m), cols)), shape=(n, m * h)).tocsc()
    K = np.zeros((d, m * h), dtype=np.float32)
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    cache = {}
    for t in range(d):
        cbs = tuple(i for i in range(m) if od[i].start <= t < od[i].stop)
        if cbs not in cache:
            idx = np.concatenate([np.arange(i * h, (i + 1) * h) for i in cbs])
            cache[cbs] = (idx, S[:, idx])
        idx, St = cache[cbs]
        K[t, idx] = lsqr(St, X[t].astype(np.float64), atol=tol, btol=tol, conlim=1e8)[0].astype(np.float32)
    return [np.ascontiguousarray(K[:, i * h:(i + 1) * h]) for i in range(m)]


def encoding_viterbi(X, C, V=False, *, block=256, exact=False):
    """Exact MAP codes of a chain (unaries + binaries between consecutive codebooks only) by dynamic programming.
    src/encodings/encode_chain.jl:1-123; min / argmin take the lowest index on ties as the reference's scans do."""
    X = _f32(X)
    C = [_f32(c) for c in C]
    d, n = X.shape
    m, h = len(C), C[0].shape[1]
    if exact:
        return (encoding_viterbi_exact(X.T, stack_codebooks(C), m, h).T + 1).astype(np.int16)
    bins = [(2.0 * C[i].T @ C[i + 1]).astype(np.float32) for i in range(m - 1)]       # :103-106
    sq = [np.einsum("ij,ij->j", c, c) for c in C]
    B = np.zeros((m, n), dtype=np.int16)
    for lo in range(0, n, block):
        Xb = X[:, lo:lo + block]
        nb = Xb.shape[1]
        U = [(-2.0 * (C[i].T @ Xb) + sq[i][:, None]).T.astype(np.float32) for i in range(m)]      # (nb, h) each; utils.jl:94-122
        back = np.zeros((m - 1, nb, h), dtype=np.int64)
        acc = U[0]
        for i in range(m - 1):                                                       # forward pass :37-68
            cost = acc[:, :, None] + bins[i][None, :, :]                             # (nb, from k, to j)
            back[i] = cost.argmin(axis=1)
            acc = U[i + 1] + np.take_along_axis(cost, back[i][:, None, :], axis=1)[:, 0, :]
        path = acc.argmin(axis=1)                                                    # :74
        B[m - 1, lo:lo + nb] = path + 1
        for i in range(m - 2, -1, -1):                                               # backward trace :77-80
            path = back[i][np.arange(nb), path]
            B[i, lo:lo + nb] = path + 1
    return B


def _qerror_full(X, B, C):
    rec = np.zeros_like(X)
    for i in range(len(C)):
        rec += C[i][:, np.asarray(B[i], dtype=np.int64) - 1]
    return float(((X - rec) ** 2).sum()) / X.shape[1]


def train_chainq(X, m, h, R, B, C, niter, V=False, exact=False):
    """train_chainq(X, m, h, R, B, C, niter) -> C, B, R, obj.  src/chainq/chainq.jl:10-58
    B: initial codes (e.g. OPQ's); the incoming C is only a placeholder, as in the reference (re-fitted at :27)."""
    X = _f32(X)
    R = _f32(R)
    B = np.asarray(B, dtype=np.int16)
    RX = R.T @ X
    C = update_codebooks_chain(RX, B, h, V)                      # :27
    B = encoding_viterbi(RX, C, V, exact=exact)                  # :31
    obj = np.zeros(niter + 1, dtype=np.float32)
    for it in range(niter + 1):
        obj[it] = _qerror_full(RX, B, C)
        if V:
            print("%3d %e" % (it, obj[it]))
        CB = np.zeros_like(X)
        for i in range(m):
            CB += C[i][:, B[i].astype(np.int64) - 1]
        R = _procrustes(X, CB)                                   # :44-45
        RX = R.T @ X
        C = update_codebooks_chain(RX, B, h, V)
        B = encoding_viterbi(RX, C, V, exact=exact)
    return C, B, R, obj
