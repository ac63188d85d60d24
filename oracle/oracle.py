"""ctypes binding of oracle/liblsq_oracle.so (the C restatement) -- test infrastructure only.

Array conventions at this boundary are the reference's Julia buffers viewed from numpy:
  X   (n, d)  float32 C-contiguous   == Julia d x n column-major
  K   (m*h, d) float32 C-contiguous  == hcat(C...) d x (m*h) column-major
  B   (n, m)  int16, 1-based         == Julia m x n Matrix{Int16}
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("LSQ_ORACLE_LIB") or os.path.join(_HERE, "liblsq_oracle.so")      # override: the sanitizer build (make asan)
_lib = None

__all__ = [
    "build", "lib", "philox4x32_10", "rng_word", "perm", "perturb", "randinit", "synth_data_u8",
    "sqnorms", "tables", "unaries", "veccost", "icm_node", "encode_icm", "encode_icm_fully", "encoding_icm_faithful",
    "qerror", "num_threads", "ref_linscan_path", "ref_linscan", "reconstruct", "quantize_norms",
]


def build(force=False):
    """Compile the C restatement (and, when /root/reference is present, oracle/_ref)."""
    target = "asan" if os.environ.get("LSQ_ORACLE_LIB") else "liblsq_oracle.so"
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "lsq_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, target], stdout=subprocess.DEVNULL)
    subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u32p, f32p, u8p, i16p, i32p, i64p, f64p = (C.POINTER(t) for t in (
            C.c_uint32, C.c_float, C.c_uint8, C.c_int16, C.c_int32, C.c_int64, C.c_double))
        L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.orc_rng_word.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_word.restype = C.c_uint32
        L.orc_perm.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, i32p]
        L.orc_perturb.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_randinit.argtypes = [C.c_uint64, C.c_uint64, C.c_long, C.c_int, C.c_int, i16p]
        L.orc_synth_data_u8.argtypes = [C.c_uint64, C.c_uint64, C.c_long, C.c_int, f32p]
        L.orc_sqnorms.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.orc_tables.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_unaries.argtypes = [f32p, f32p, C.c_long, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_veccost.argtypes = [f32p, f32p, u8p, C.c_long, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_icm_node.argtypes = [f32p, f32p, u8p, C.c_int, C.c_int, C.c_int]
        L.orc_icm_node.restype = C.c_int
        L.orc_encode_icm.argtypes = [f32p, i16p, f32p, C.c_int, C.c_long, C.c_int, C.c_int, i64p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, i16p, f32p, f64p]
        L.orc_encode_icm.restype = C.c_int
        L.orc_encoding_icm_faithful.argtypes = [f32p, i16p, f32p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, i16p]
        L.orc_encoding_icm_faithful.restype = C.c_int
        L.orc_encode_icm_fully.argtypes = [f32p, i16p, f32p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_uint64, C.c_uint32, C.c_uint64, f32p]
        L.orc_encode_icm_fully.restype = C.c_int
        L.orc_qerror.argtypes = [f32p, i16p, f32p, C.c_int, C.c_long, C.c_int, C.c_int]
        L.orc_qerror.restype = C.c_double
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def num_threads():
    return int(lib().orc_num_threads())


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32).copy()
    k = np.asarray(key, dtype=np.uint32).copy()
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(out, C.c_uint32))
    return out


def rng_word(seed, idx, it, domain, w):
    return int(lib().orc_rng_word(seed, idx, it, domain, w))


def perm(seed, it, m, randord=True):
    out = np.zeros(m, dtype=np.int32)
    lib().orc_perm(seed, it, m, int(bool(randord)), _p(out, C.c_int32))
    return out


def perturb(seed, gidx, it, code, h, npert):
    """code: uint8 0-based (m,) -> perturbed copy."""
    c = np.ascontiguousarray(code, dtype=np.uint8).copy()
    lib().orc_perturb(seed, gidx, it, c.shape[0], h, npert, _p(c, C.c_uint8))
    return c


def randinit(seed, n, m, h, global_offset=0):
    B = np.zeros((n, m), dtype=np.int16)
    lib().orc_randinit(seed, global_offset, n, m, h, _p(B, C.c_int16))
    return B


def synth_data_u8(seed, n, d, global_offset=0):
    X = np.zeros((n, d), dtype=np.float32)
    lib().orc_synth_data_u8(seed, global_offset, n, d, _p(X, C.c_float))
    return X


def sqnorms(K):
    K = _f32(K)
    out = np.zeros(K.shape[0], dtype=np.float32)
    lib().orc_sqnorms(_p(K, C.c_float), K.shape[0], K.shape[1], _p(out, C.c_float))
    return out


def tables(K, m, h):
    """-> T (m, m, h, h): T[j,k,b,:] is the column added to node j when codebook k holds b."""
    K = _f32(K)
    d = K.shape[1]
    T = np.zeros((m, m, h, h), dtype=np.float32)
    lib().orc_tables(_p(K, C.c_float), m, h, d, _p(T, C.c_float))
    return T


def unaries(X, K, m, h):
    """-> U (m, n, h)   (the reference's unaries[j] = h x n column-major)."""
    X, K = _f32(X), _f32(K)
    n, d = X.shape
    U = np.zeros((m, n, h), dtype=np.float32)
    lib().orc_unaries(_p(X, C.c_float), _p(K, C.c_float), n, d, m, h, _p(U, C.c_float))
    return U


def veccost(X, K, codes_u8, h):
    X, K = _f32(X), _f32(K)
    codes = np.ascontiguousarray(codes_u8, dtype=np.uint8)
    n, d = X.shape
    m = codes.shape[1]
    out = np.zeros(n, dtype=np.float32)
    lib().orc_veccost(_p(X, C.c_float), _p(K, C.c_float), _p(codes, C.c_uint8), n, d, m, h, _p(out, C.c_float))
    return out


def icm_node(uj, T, code_u8, j):
    uj = _f32(uj)
    T = _f32(T)
    m, h = T.shape[0], T.shape[2]
    code = np.ascontiguousarray(code_u8, dtype=np.uint8)
    return int(lib().orc_icm_node(_p(uj, C.c_float), _p(T, C.c_float), _p(code, C.c_uint8), j, m, h))


def encode_icm(X, B, K, m, h, ilsiters, icmiter, npert, randord, seed, global_offset=0, want_stats=False):
    """Whole encode call (encode_icm_cuda-shaped).  -> Bs (nr, n, m) int16 1-based, objs (nr,) f32[, stats]."""
    X, K = _f32(X), _f32(K)
    B = np.ascontiguousarray(B, dtype=np.int16)
    n, d = X.shape
    ils = np.ascontiguousarray(ilsiters, dtype=np.int64)
    nr = ils.shape[0]
    Bs = np.zeros((nr, n, m), dtype=np.int16)
    objs = np.zeros(nr, dtype=np.float32)
    I = int(ils.max())
    stats = np.zeros((I, 2), dtype=np.float64)
    rc = lib().orc_encode_icm(_p(X, C.c_float), _p(B, C.c_int16), _p(K, C.c_float), d, n, m, h,
                              _p(ils, C.c_int64), nr, icmiter, npert, int(bool(randord)), seed, global_offset,
                              _p(Bs, C.c_int16), _p(objs, C.c_float), _p(stats, C.c_double))
    if rc != 0:
        raise ValueError("orc_encode_icm failed with %d" % rc)
    return (Bs, objs, stats) if want_stats else (Bs, objs)


def encoding_icm_faithful(X, oldB, K, m, h, niter, randord, npert, seed, it, nworkers=1, global_offset=0):
    """One ILS iteration with the reference's own loop nest (encode_icm.jl:131-189) -> B (n, m) int16."""
    X, K = _f32(X), _f32(K)
    oldB = np.ascontiguousarray(oldB, dtype=np.int16)
    n, d = X.shape
    out = np.zeros((n, m), dtype=np.int16)
    rc = lib().orc_encoding_icm_faithful(_p(X, C.c_float), _p(oldB, C.c_int16), _p(K, C.c_float), d, n, m, h,
                                         niter, int(bool(randord)), npert, seed, it, global_offset, nworkers,
                                         _p(out, C.c_int16))
    if rc != 0:
        raise ValueError("orc_encoding_icm_faithful failed with %d" % rc)
    return out


def encode_icm_fully(X, B, K, m, h, niter, randord, npert, seed=0, it=0, global_offset=0, want_margins=False):
    """The worker `encode_icm_fully!` (encode_icm.jl:4-127): perturb + niter sweeps, NO accept test.  -> B' (n, m) int16
    [, margins (n,) f32: the smallest runner-up gap met in any node update of the vector]."""
    X, K = _f32(X), _f32(K)
    out = np.array(B, dtype=np.int16, order="C", copy=True)
    n, d = X.shape
    margins = np.zeros(max(n, 1), dtype=np.float32)
    rc = lib().orc_encode_icm_fully(_p(X, C.c_float), _p(out, C.c_int16), _p(K, C.c_float), d, n, m, h, niter,
                                    int(bool(randord)), npert, seed, it, global_offset, _p(margins, C.c_float))
    if rc != 0:
        raise ValueError("orc_encode_icm_fully failed with %d" % rc)
    return (out, margins[:n]) if want_margins else out


def qerror(X, B, K, m, h):
    X, K = _f32(X), _f32(K)
    B = np.ascontiguousarray(B, dtype=np.int16)
    n, d = X.shape
    return float(lib().orc_qerror(_p(X, C.c_float), _p(B, C.c_int16), _p(K, C.c_float), d, n, m, h))


# ---- the REAL reference, where it compiles: the ADC linear scan (oracle/_ref) ----------------

def reconstruct(B, C):
    """Reference src/utils.jl:203-223 restated: CB[:, j] = ((0 + C[0][:, B[0, j]]) + C[1][:, B[1, j]]) + ...  -- codebooks ascending, plain f32
    adds from +0, one vector at a time (the reference's loop nest is codebook-outer / vector-inner; the sums of ONE vector see the same order).
    B: (m, n) Int16 1-based, C: list of m (d, h) f32.  -> (d, n) f32.  Checker only: tests/ and the bench's checking legs may call it."""
    B = np.asarray(B)
    m, n = B.shape
    d = np.asarray(C[0]).shape[0]
    CB = np.zeros((d, n), dtype=np.float32)
    for j in range(n):                                   # vector by vector: an independent loop order from the product's vectorised mirror
        acc = np.zeros(d, dtype=np.float32)
        for i in range(m):
            acc = acc + np.asarray(C[i], dtype=np.float32)[:, int(B[i, j]) - 1]
        CB[:, j] = acc
    return CB


def quantize_norms(B, C, cbnorms, want_norms=False):
    """Reference src/utils.jl:6-31 restated: ithnorm = SUM_j CB[j, i]^2 accumulated in f32 with j ASCENDING, each square rounded before its add
    (the reference's `@simd` loop leaves the order to the compiler: [build-defined], frozen here as the sequential order); then
    dists2norm[c] = (ithnorm - cbnorms[c])^2 in f32 and `findmin`: the FIRST index of the minimum (1-based Int16).
    -> idx (n,) int16 [, norms (n,) f32].  Checker only."""
    CB = reconstruct(B, C)
    d, n = CB.shape
    cb = np.asarray(cbnorms, dtype=np.float32)
    idx = np.empty(n, dtype=np.int16)
    norms = np.empty(n, dtype=np.float32)
    for i in range(n):
        acc = np.float32(0.0)
        col = CB[:, i]
        for j in range(d):
            acc = np.float32(acc + np.float32(col[j] * col[j]))
        norms[i] = acc
        best, bi = None, 0
        for c in range(cb.shape[0]):                     # strict '<' scan from the first entry = findmin
            dv = np.float32(np.float32(acc - cb[c]) * np.float32(acc - cb[c]))
            if best is None or dv < best:
                best, bi = dv, c
        idx[i] = bi + 1
    return (idx, norms) if want_norms else idx


def ref_linscan_path():
    p = os.path.join(_HERE, "_ref", "linscan_aqd_pairwise_byte.so")
    return p if os.path.exists(p) else None


def ref_linscan(codes_u8, queries, K, dbnorms, m, h, knn):
    """Calls the reference's own linscan_aqd_query_extra_byte
    (src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:97-104, bound as in src/linscan/Linscan.jl:63-69).
    -> dists (nq, knn) f32, idx (nq, knn) int32 (1-based)."""
    path = ref_linscan_path()
    if path is None:
        raise FileNotFoundError("oracle/_ref/linscan_aqd_pairwise_byte.so not built (run `make -C oracle ref`)")
    L = C.CDLL(path)
    codes = np.ascontiguousarray(codes_u8, dtype=np.uint8)
    Q, K = _f32(queries), _f32(K)
    dbn = _f32(dbnorms)
    nq, d = Q.shape
    n = codes.shape[0]
    dists = np.zeros((nq, knn), dtype=np.float32)
    idx = np.zeros((nq, knn), dtype=np.int32)
    f = L.linscan_aqd_query_extra_byte
    f.restype = None
    f.argtypes = [C.c_void_p] * 6 + [C.c_int] * 6
    f(dists.ctypes.data, idx.ctypes.data, codes.ctypes.data, Q.ctypes.data, K.ctypes.data, dbn.ctypes.data,
      nq, n, m, h, d, knn)
    return dists, idx
