"""oracle/train_oracle.py -- checker for SURVEY 8(f)-4: the step order of the reference trainer `train_lsq` (src/lsq/LSQ.jl:10-88) restated with
INDEPENDENT pieces -- the oracle's C encoder (oracle/lsq_oracle.c) for every encoding_icm call and scipy's LSQR (the sanity oracle SURVEY 8(f)-3
names for IterativeSolvers.lsqr) for every update_codebooks call.  Test infrastructure: only tests/ may import it.

    LSQ.jl:36        C = update_codebooks(RX, B, h)              (R = I here: the rotation is glue)
    LSQ.jl:45-48     ilsiter x  B = encoding_icm(X, B, C, icmiter, randord, npert)
    LSQ.jl:53-66     niter x { obj = qerror(X, B, C); C = update_codebooks(X, B, h); ilsiter x encoding_icm }

encoding_icm keeps no iteration counter in the reference (Julia's global RNG advances); the build-defined RNG of this repo keys every call by
(seed + call index, ILS iteration within the call) -- the same keys `train_lsq` / `train_lsq_dev` use, so the product and this checker draw the same
perturbations and differ only through the LSQR arithmetic (f32 here vs f64 scipy)."""
import numpy as np

from . import oracle as O


def sparsify_codes(B, h):
    """src/utils.jl:50-69: the n x (m h) one-hot-per-codebook matrix of the codes (1-based B (n, m))."""
    import scipy.sparse as sp
    n, m = B.shape
    rows = np.repeat(np.arange(n), m)
    cols = (np.arange(m)[None, :] * h + (B.astype(np.int64) - 1)).reshape(-1)
    return sp.csr_matrix((np.ones(n * m), (rows, cols)), shape=(n, m * h))


def update_codebooks(X, B, h):
    """src/codebook_update.jl:52-86 with scipy.sparse.linalg.lsqr (x0 = 0, damp = 0, atol = btol = sqrt(eps(Float32)): IterativeSolvers' defaults) in f64.
    X (n, d) f32, B (n, m) int16 1-based -> K (m h, d) f32."""
    import scipy.sparse.linalg as spl
    S = sparsify_codes(B, h)
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    d = X.shape[1]
    K = np.stack([spl.lsqr(S, X[:, t].astype(np.float64), atol=tol, btol=tol, conlim=1e8, iter_lim=max(S.shape))[0] for t in range(d)], axis=1)
    return np.ascontiguousarray(K, dtype=np.float32)


def train_lsq(X, m, h, B0, niter, ilsiter, icmiter, randord, npert, seed=0):
    """X (n, d) f32, B0 (n, m) int16 1-based.  -> (K (m h, d) f32, B (n, m) int16, obj (niter,) f64): obj[t] = qerror before the t-th update (LSQ.jl:55)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    B = np.ascontiguousarray(B0, dtype=np.int16)
    K = update_codebooks(X, B, h)
    call = 0

    def encode(Bc, call):
        Bs, _ = O.encode_icm(X, Bc, K, m, h, [ilsiter], icmiter, npert, randord, seed + call)
        return Bs[-1]

    B = encode(B, call)
    obj = np.zeros(niter, dtype=np.float64)
    for t in range(niter):
        obj[t] = O.qerror(X, B, K, m, h)
        K = update_codebooks(X, B, h)
        call += 1
        B = encode(B, call)
    return K, B, obj
