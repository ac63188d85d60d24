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


def encoding_icm(X, oldB, C, niter, randord, npert, V=False, *, seed=0, it=None, engine=None):
    """One ILS iteration (encode_icm.jl:131-189).  With the reference's own argument list (no `it`) the engine counts the calls, so the
    demo loop `for i = 1:ilsiter; B = encoding_icm(...); end` (demos/demo_lsq.jl:48-51) perturbs differently every time -- and equals
    encode_icm_cuda(..., [ilsiter], ...) on a fresh engine (or after engine.set_option("ils_counter", 0))."""
    eng = engine or default_engine()
    m, d, h = _dims(C)
    return eng.encoding_icm(_X_of(X), _B_of(oldB), _K_of(C), m, niter, randord, npert, seed=seed, it=it, h=h).T


def encode_icm_fully(B, X, C, binaries, cbi, niter, randord, npert, IDX, V=False, *, seed=0, it=None, engine=None):
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


def linscan_lsq(B, X, C, dbnorms, R, k=10000, *, nthreads=0, engine=None):
    """Linear scan with LSQ codes + separately stored norms  (src/linscan/Linscan.jl:46-73).
    B (m, n) uint8 0-based; X (d, nq) queries; C list of (d, h); dbnorms (n,); R (d, d) rotation.
    -> dists (k, nq) float32 ascending, res (k, nq) int32 1-based ids.
    engine=None: the host scan (lsq_linscan_aqd_query_extra_byte, the reference's own division of labour);
    engine=<Engine>: the device scan (lsq_linscan) -- same results bit for bit."""
    from . import _lib
    B = np.ascontiguousarray(np.asarray(B, dtype=np.uint8).T)                   # (n, m)
    RX = np.ascontiguousarray((np.asarray(R, dtype=np.float32).T @ np.asarray(X, dtype=np.float32)).T)   # (nq, d)
    K = _K_of(C)
    dbn = np.ascontiguousarray(dbnorms, dtype=np.float32)
    n, m = B.shape
    nq, d = RX.shape
    h = np.asarray(C[0]).shape[1]
    if engine is not None:
        dists, res = engine.linscan(B, RX, K, dbn, m, k, h=h)
        return dists.T, res.T
    dists = np.zeros((nq, k), dtype=np.float32)
    res = np.zeros((nq, k), dtype=np.int32)
    _lib.check(_lib.load().lsq_linscan_aqd_query_extra_byte(dists.ctypes.data, res.ctypes.data, B.ctypes.data, RX.ctypes.data,
                                                            K.ctypes.data, dbn.ctypes.data, nq, n, m, h, d, k, int(nthreads)))
    return dists.T, res.T


def eval_recall(ids_gnd, ids_predicted, k, V=False):
    """recall@N curve (src/linscan/Linscan.jl:76-117): ids_gnd (nq,), ids_predicted (k, nq), same id base.
    -> recall_at_i (k,) with recall_at_i[i-1] = fraction of queries whose true neighbour ranks <= i."""
    ids_gnd = np.asarray(ids_gnd)
    P = np.asarray(ids_predicted)
    nq = P.shape[1]
    assert nq == ids_gnd.shape[0]
    ranks = np.full(nq, k + 1, dtype=np.int64)
    for i in range(nq):
        pos = np.nonzero(P[:k, i] == ids_gnd[i])[0]
        if pos.size == 1:                                   # the reference requires exactly one hit (:94-98)
            ranks[i] = pos[0] + 1
    rec = np.array([(ranks <= i).sum() / nq for i in range(1, k + 1)], dtype=np.float64)
    if V:
        for i in (1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000):
            if i <= k:
                print("r@%d = %s" % (i, 100.0 * rec[i - 1]))
    return rec


def update_codebooks(X, B, h, V=False, codebook_upd_method="lsqr", *, nthreads=0, engine=None):
    """src/codebook_update.jl:52-86 -> list of m (d, h) codebooks minimising ||X - sum_j C_j[:, B_j]||^2 (LSQR, or LSMR with codebook_upd_method="lsmr": host only).
    engine=None: the host solver (std::thread workers over the dimensions, the reference's own division of labour);
    engine=<Engine>: the device solver (lsq_update_codebooks_gpu: all dimensions at once) -- the same result on every tested problem (required: 1e-5)."""
    from . import _lib
    if codebook_upd_method not in ("lsqr", "lsmr"):
        raise ValueError("Codebook update method unknown: %r" % (codebook_upd_method,))      # codebook_update.jl:22-23,60
    if codebook_upd_method == "lsmr" and engine is not None:
        raise ValueError("the device solver is LSQR (the reference's default); 'lsmr' runs on the host: pass engine=None")
    Xr, Br = _X_of(X), _B_of(B)
    n, d = Xr.shape
    m = Br.shape[1]
    if engine is not None:
        K, _ = engine.update_codebooks(Xr, Br, m, h=h)
        return [np.ascontiguousarray(K[j * h:(j + 1) * h].T) for j in range(m)]
    K = np.zeros((m * h, d), dtype=np.float32)
    fn = _lib.load().lsq_update_codebooks_lsmr if codebook_upd_method == "lsmr" else _lib.load().lsq_update_codebooks
    _lib.check(fn(Xr.ctypes.data, Br.ctypes.data, d, n, m, h, int(nthreads), K.ctypes.data))
    return [np.ascontiguousarray(K[j * h:(j + 1) * h].T) for j in range(m)]


def train_lsq(X, m, h, R, B, C, niter, ilsiter, icmiter, randord, npert, V=False, *, seed=0, engine=None, device_update=False):
    """src/lsq/LSQ.jl:10-88: alternate codebook update (host LSQR) and ILS/ICM encoding (GPU).
    -> (C, B, cbnorms, B_norms, obj).  The final norm codebook is the reference's plain k-means on the squared
    norms of the reconstructions (Clustering.kmeans there; initializers.kmeans here: k-means++ seeding + Lloyd, <= 100 sweeps, seeded -- unpinned)."""
    X = np.asarray(X, dtype=np.float32)
    R = np.asarray(R, dtype=np.float32)
    d, n = X.shape
    RX = R.T @ X
    upd_engine = engine if device_update else None          # device_update: the LSQR codebook update on the device too (needs `engine`)
    C = update_codebooks(RX, B, h, V, engine=upd_engine)
    C = [R @ Ci for Ci in C]
    it = 0

    def encode(Bc, it):
        Bs, _ = encode_icm_cuda(X, Bc, C, [ilsiter], icmiter, npert, randord, 1, V, seed=seed + it, engine=engine)
        return Bs[-1]

    B = encode(np.asarray(B, dtype=np.int16), it)
    obj = np.zeros(niter, dtype=np.float32)
    for iter_ in range(niter):
        obj[iter_] = qerror(X, B, C, engine=engine)
        if V:
            print("%3d %e" % (iter_ + 1, obj[iter_]))
        C = update_codebooks(X, B, h, V, engine=upd_engine)
        it += 1
        B = encode(B, it)
    # the codebook for norms (LSQ.jl:67-84): squared norms of the reconstructions (f32, dimensions ascending), then "plain-old k-means" with h centres.
    # The reference calls Clustering.jl's kmeans(dbnorms, h) -- k-means++ seeding, Lloyd until the assignments settle (at most 100 sweeps by default); un-vendored
    # and seeded from Julia's global RNG, so the centres are unpinned: the package's own kmeans() (initializers.py) is the same algorithm under `seed`.
    from .initializers import kmeans
    CB = reconstruct(B, C)
    dbnorms = np.zeros(n, dtype=np.float32)
    for j in range(CB.shape[0]):
        dbnorms += CB[j] * CB[j]
    centers, assign, _ = kmeans(dbnorms.reshape(1, n), min(h, n), niter=100, seed=seed)
    cb = centers.reshape(-1).astype(np.float32)
    B_norms = (assign + 1).reshape(1, n).astype(np.int16)
    return C, B, cb, B_norms, obj


def train_lsq_dev(dX, m, h, dB, niter, ilsiter, icmiter, randord, npert, *, seed=0, engine, R=None, norm_codebook=True):
    """src/lsq/LSQ.jl:10-88 with everything resident in HBM: the same alternation as train_lsq -- LSQR codebook update (lsq_update_codebooks_dev),
    ILS/ICM encode (lsq_encode_icm_dev, seed + call index), objective -- on device tensors, no host copy of X, the codes or the codebooks between the
    steps.  dX (n, d) f32, dB (n, m) uint8 0-BASED: CUDA/HIP torch tensors (the layouts of Engine.encode_icm_dev); R (d, d) rotation or None (identity).
    -> (dK (m*h, d) tensor, dB (n, m) uint8 tensor, cbnorms (<= h,) f32, B_norms (1, n) int16 1-based, obj (niter,) f32).  Same codebooks, codes and
    objective as train_lsq(..., engine=engine, device_update=True) on the same inputs (tests/test_pipeline_gpu.py).  torch is used for the two
    rotations only (RX = X R once; K R' once): glue, not the path."""
    import torch
    n, d = dX.shape
    dXr = dX if R is None else (dX @ torch.as_tensor(np.asarray(R, dtype=np.float32), device=dX.device)).contiguous()      # rows of RX' = (R' x)'
    dK, _ = engine.update_codebooks_dev(dXr, dB.contiguous(), m, h=h)
    if R is not None:
        dK = (dK @ torch.as_tensor(np.asarray(R, dtype=np.float32), device=dX.device).T).contiguous()                       # C = R C: rows of K are codewords
    del dXr
    it = 0

    def encode(dBc, it):
        dBs, sums, _ = engine.encode_icm_dev(dX, dBc, dK, m, [ilsiter], icmiter, npert, randord, seed=seed + it, h=h)
        return dBs[-1], float(sums[-1])

    dB, s = encode(dB.contiguous(), it)
    obj = np.zeros(niter, dtype=np.float32)
    for iter_ in range(niter):
        obj[iter_] = s / n                                                   # qerror(X, B, C): the mean cost of the codes the encode just returned
        dK, _ = engine.update_codebooks_dev(dX, dB, m, h=h)
        it += 1
        dB, s = encode(dB, it)
    if not norm_codebook:
        return dK, dB, None, None, obj
    # the codebook for norms (LSQ.jl:67-84): squared norms of the reconstructions on the device, the reference's plain k-means on the host (n scalars)
    from .initializers import kmeans
    _, _, nrm = engine.quantize_norms_dev(dB, dK, torch.zeros(1, dtype=torch.float32, device=dX.device), m, h=h)
    dbnorms = nrm.cpu().numpy()
    centers, assign, _ = kmeans(dbnorms.reshape(1, n), min(h, n), niter=100, seed=seed)
    return dK, dB, centers.reshape(-1).astype(np.float32), (assign + 1).reshape(1, n).astype(np.int16), obj      # B_norms 1 x n like train_lsq (LSQ.jl:84)


def reconstruct(B, C):
    """src/utils.jl:203-223: CB = sum_i C[i][:, B[i, :]] accumulated in codebook order from zero (f32). -> (d, n)"""
    B = np.asarray(B)
    d = np.asarray(C[0]).shape[0]
    CB = np.zeros((d, B.shape[1]), dtype=np.float32)
    for i, Ci in enumerate(C):
        CB += np.asarray(Ci, dtype=np.float32)[:, B[i].astype(np.int64) - 1]
    return CB


def quantize_norms(B, C, cbnorms, *, engine=None):
    """src/utils.jl:6-31: squared norm of every reconstruction (f32, dimensions ascending), then the index
    (1-based Int16) of the nearest scalar centroid in `cbnorms`; ties -> lowest index (findmin).
    engine=<Engine>: the device kernel (lsq_quantize_norms) -- the same numbers bit for bit."""
    if engine is not None:
        idx, _, _ = engine.quantize_norms(np.ascontiguousarray(np.asarray(B, dtype=np.int16).T), _K_of(C), cbnorms, len(C), h=np.asarray(C[0]).shape[1])
        return idx
    CB = reconstruct(B, C)
    norms = np.zeros(CB.shape[1], dtype=np.float32)
    for j in range(CB.shape[0]):                      # sequential f32 accumulation, j ascending
        norms += CB[j] * CB[j]
    cb = np.asarray(cbnorms, dtype=np.float32)
    d2 = (norms[None, :] - cb[:, None]) ** 2          # (h, n) f32
    return (np.argmin(d2, axis=0) + 1).astype(np.int16)


def _vecs_read(filename, bounds, dtype, item):
    """TEXMEX .fvecs/.ivecs/.bvecs (src/read/*vecs_read.jl): records [int32 d][d x item]; bounds = (a, b) 1-based
    inclusive, an int n (first n vectors) or None (all).  -> (d, n) column-per-vector like the reference."""
    with open(filename, "rb") as f:
        d = int(np.frombuffer(f.read(4), dtype=np.int32)[0])
        rec = 4 + d * item
        f.seek(0, 2)
        total = f.tell() // rec
        if bounds is None:
            a, b = 1, total
        elif isinstance(bounds, (int, np.integer)):
            a, b = 1, int(bounds)
        else:
            a, b = int(bounds[0]), int(bounds[-1])
        assert a >= 1 and b <= total
        f.seek((a - 1) * rec)
        raw = np.frombuffer(f.read((b - a + 1) * rec), dtype=np.uint8).reshape(b - a + 1, rec)
    dims = raw[:, :4].copy().view(np.int32).ravel()
    assert np.all(dims == d), "inconsistent dimension headers"
    return np.ascontiguousarray(raw[:, 4:]).view(dtype).reshape(b - a + 1, d).T


def fvecs_read(bounds, filename):
    return _vecs_read(filename, bounds, np.float32, 4)


def ivecs_read(bounds, filename):
    return _vecs_read(filename, bounds, np.int32, 4)


def bvecs_read(bounds, filename):
    return _vecs_read(filename, bounds, np.uint8, 1)


def randinit(n, m, h, *, seed=0):
    return _engine.randinit(n, m, h, seed=seed).T


def splitarray(x, nparts):
    """x: a range (1-based like Julia's 1:n) -> list of ranges."""
    n = len(x)
    return [x[s:e] for s, e in _engine.splitarray(n, nparts)]
