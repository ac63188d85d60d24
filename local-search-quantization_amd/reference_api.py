"""Host-side mirror of the reference's operator interface for the encoding path.

Same function names, positional arguments, array shapes (Julia d x n / m x n / list of d x h) and
1-based Int16 codes as the reference, so parity tests read like calls into the reference:

    encode_icm_cuda(RX, B, C, ilsiters, icmiter, npert, randord, nsplits=2, V=False) -> (Bs, objs)
        src/encodings/encode_icm_cuda.jl:253-262   (call site demos/demo_lsq_gpu.jl:50)
    encoding_icm(X, oldB, C, niter, randord, npert, V=False) -> B
        src/encodings/encode_icm.jl:131-189
    encode_icm_fully(B, X, C, binaries, cbi, niter, randord, npert, IDX, V)   (in place)
        src/encodings/encode_icm.jl:4-127
    get_unaries, get_binaries, veccost, qerror, splitarray   src/utils.jl
    randinit                                                  src/initializations.jl

Extra keyword-only arguments (`seed`, `it`, `engine`) expose what the reference leaves to global
RNG state.  Everything runs in liblsq_mi355x.so on the GPU; there is no CPU fallback here.
"""
import numpy as np

from . import engine as _engine

_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = _engine.Engine(0)
    return _default_engine


def _K_of(C):
    """hcat(C...) as the (m*h, d) row-major buffer (== Julia d x (m*h) column-major)."""
    return np.ascontiguousarray(np.concatenate([np.asarray(Cj, dtype=np.float32).T for Cj in C], axis=0))


def _dims(C):
    d, h = np.asarray(C[0]).shape
    return len(C), d, h


def _X_of(X):
    return np.ascontiguousarray(np.asarray(X, dtype=np.float32).T)       # (d, n) -> (n, d)


def _B_of(B):
    return np.ascontiguousarray(np.asarray(B, dtype=np.int16).T)         # (m, n) -> (n, m)


def encode_icm_cuda(RX, B, C, ilsiters, icmiter, npert, randord, nsplits=2, V=False, *, seed=0, engine=None):
    """-> (Bs, objs): Bs = list of (m, n) int16 1-based matrices, objs = float32 vector."""
    eng = engine or default_engine()
    m, d, h = _dims(C)
    Bs, objs = eng.encode_icm(_X_of(RX), _B_of(B), _K_of(C), m, list(ilsiters), icmiter, npert, randord,
                              seed=seed, nsplits=nsplits, verbose=V, h=h)
    return [Bs[r].T for r in range(Bs.shape[0])], objs


def encoding_icm(X, oldB, C, niter, randord, npert, V=False, *, seed=0, it=0, engine=None):
    eng = engine or default_engine()
    m, d, h = _dims(C)
    return eng.encoding_icm(_X_of(X), _B_of(oldB), _K_of(C), m, niter, randord, npert, seed=seed, it=it, h=h).T


def encode_icm_fully(B, X, C, binaries, cbi, niter, randord, npert, IDX, V=False, *, seed=0, it=0, engine=None):
    """In place on B (m, n) int16.  `binaries`/`cbi` are accepted for signature parity and ignored
    (rebuilt on the device from C).  IDX = (first, last) 1-based or a range."""
    eng = engine or default_engine()
    m, d, h = _dims(C)
    first = IDX[0] if not isinstance(IDX, range) else IDX.start
    Bt = _B_of(B)
    eng.encode_icm_fully(Bt, _X_of(X), _K_of(C), m, niter, randord, npert, idx_first=int(first), seed=seed, it=it, h=h)
    B[...] = Bt.T
    return B


def get_unaries(X, C, V=False, *, engine=None):
    """-> list of m (h, n) matrices (unaries[j][a, i])."""
    eng = engine or default_engine()
    m, d, h = _dims(C)
    U = eng.get_unaries(_X_of(X), _K_of(C), m, h=h)
    return [U[j].T for j in range(m)]


def get_binaries(C, *, engine=None):
    """-> (binaries, cbi): binaries[idx] is (h, h) with [a, b] = 2<c_i[a], c_j[b]>, cbi (2, ncbi) 1-based, i<j i-major."""
    eng = engine or default_engine()
    m, d, h = _dims(C)
    T = eng.get_binaries(_K_of(C), m, h=h)
    binaries, cbi = [], []
    for i in range(m):
        for j in range(i + 1, m):
            binaries.append(T[i, j].T)        # T[i][j][b][a] -> [a, b]
            cbi.append((i + 1, j + 1))
    return binaries, np.asarray(cbi, dtype=np.int32).T.reshape(2, -1)


def veccost(X, B, C, *, engine=None):
    eng = engine or default_engine()
    m, d, h = _dims(C)
    return eng.veccost(_X_of(X), _B_of(B), _K_of(C), m, h=h)


def qerror(X, B, C, *, engine=None):
    eng = engine or default_engine()
    m, d, h = _dims(C)
    return eng.qerror(_X_of(X), _B_of(B), _K_of(C), m, h=h)


def randinit(n, m, h, *, seed=0):
    return _engine.randinit(n, m, h, seed=seed).T


def splitarray(x, nparts):
    """x: a range (1-based like Julia's 1:n) -> list of ranges."""
    n = len(x)
    return [x[s:e] for s, e in _engine.splitarray(n, nparts)]
