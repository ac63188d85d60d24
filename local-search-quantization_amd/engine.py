"""Engine: thin, typed Python handle on the C-ABI context (one per process / per GPU).

Row-major conventions (the Julia buffers seen from numpy, see include/lsq_mi355x.h):
    X (n, d) float32 | K (m*h, d) float32 | B (n, m) int16 1-based (host) / uint8 0-based (device)
torch is used only as the owner of device memory and streams; every computation happens inside
liblsq_mi355x.so.
"""
import contextlib
import ctypes as C

import numpy as np

from . import _lib

H = 256
IT_AUTO = 0xFFFFFFFF      # LSQ_IT_AUTO: the context counts the ILS iterations of the CPU-shaped entry points


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Engine:
    def __init__(self, device=0, chunk=None, profile=False, schedule=None, skip=None, tuning=False):
        """tuning=True loads liblsq_mi355x_tuning.so (same ABI + option "ablation", environment knobs, clock stamps)."""
        self._L = _lib.load(tuning=tuning)
        h = C.c_void_p()
        self._check(self._L.lsq_create(C.byref(h), int(device)))
        self._h = h
        self.device = int(device)
        if chunk is not None:
            self.set_option("chunk", int(chunk))
        if schedule is not None:
            self.set_option("schedule", int(schedule))
        if skip is not None:
            self.set_option("skip", int(bool(skip)))
        if profile:
            self.set_option("profile", 1)

    def _check(self, rc):
        return _lib.check(rc, self._L)

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.lsq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- options / timing -----------------------------------------------------------------
    def set_option(self, key, value):
        self._check(self._L.lsq_set_option(self._h, key.encode(), int(value)))

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.lsq_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    @contextlib.contextmanager
    def _on_torch_stream(self):
        """Bind the context to torch's CURRENT stream for the duration of one call, then return to the context's own stream.
        The torch stream object is held for as long as its raw handle is bound, so a temporary stream
        (`with torch.cuda.stream(torch.cuda.Stream())`) can never leave a dangling hipStream_t behind."""
        import torch
        stream = torch.cuda.current_stream(self.device)
        self.set_stream(stream.cuda_stream)
        try:
            yield stream
        finally:
            self.set_option("own_stream", 1)

    def synchronize(self):
        self._check(self._L.lsq_synchronize(self._h))

    def timings(self):
        t = _lib.Timings()
        self._check(self._L.lsq_get_timings_sized(self._h, C.addressof(t), C.sizeof(t)))      # size-checked: this binding and the library may be of different versions
        return t.as_dict()

    def walk_trace(self, count=64):
        """recomputed node updates per position (sweep * m + rank in the visiting order) since reset_timings  [lsq_get_walk_trace]"""
        out = np.zeros(count, dtype=np.int64)
        self._check(self._L.lsq_get_walk_trace(self._h, out.ctypes.data, int(count)))
        return out

    def reset_timings(self):
        self._check(self._L.lsq_reset_timings(self._h))

    # -- (1) whole call, host buffers -------------------------------------------------------
    def encode_icm(self, X, B, K, m, ilsiters, icmiter, npert, randord, seed=0, nsplits=1, global_offset=0,
                   verbose=False, h=H):
        """-> Bs (nr, n, m) int16 1-based, objs (nr,) float32   [lsq_encode_icm]"""
        X, K, B = _np(X, np.float32), _np(K, np.float32), _np(B, np.int16)
        n, d = X.shape
        self._check_shapes(X, K, B, m, h)
        ils = _np(ilsiters, np.int64).reshape(-1)
        nr = ils.shape[0]
        Bs = np.empty((nr, n, m), dtype=np.int16)
        objs = np.zeros(nr, dtype=np.float32)
        self._check(self._L.lsq_encode_icm(self._h, X.ctypes.data, B.ctypes.data, K.ctypes.data, d, n, m, h,
                                          ils.ctypes.data, nr, int(icmiter), int(npert), int(bool(randord)),
                                          int(nsplits), int(seed), int(global_offset), int(bool(verbose)),
                                          Bs.ctypes.data, objs.ctypes.data))
        return Bs, objs

    # -- (1b) whole call, device-resident torch tensors -------------------------------------
    def encode_icm_dev(self, dX, dB0, dK, m, ilsiters, icmiter, npert, randord, seed=0, global_offset=0,
                       out=None, h=H, nonblocking=False):
        """dX (n,d) f32, dB0 (n,m) u8 0-based, dK (m*h,d) f32: CUDA/HIP torch tensors.
        -> dBs (nr, n, m) uint8 tensor, obj_sums (nr,) float64 numpy (SUM of costs), stats (I, 2) int64.
        nonblocking=True (option "async"): nothing in the call waits for the device; obj_sums and stats come back as DEVICE tensors, valid after the
        caller has synchronised torch's current stream (the call can be captured into a graph from its second use on a shape on)."""
        import torch
        assert dX.is_cuda and dB0.is_cuda and dK.is_cuda, "device tensors required"
        assert dX.dtype == torch.float32 and dK.dtype == torch.float32 and dB0.dtype == torch.uint8
        assert dX.is_contiguous() and dB0.is_contiguous() and dK.is_contiguous()
        n, d = dX.shape
        if dB0.shape != (n, m) or dK.shape != (m * h, d):
            raise ValueError("shape mismatch: X %s B %s K %s m=%d h=%d" % (tuple(dX.shape), tuple(dB0.shape), tuple(dK.shape), m, h))
        ils = _np(ilsiters, np.int64).reshape(-1)
        nr = ils.shape[0]
        I = int(ils.max())
        dBs = out if out is not None else torch.empty((nr, n, m), dtype=torch.uint8, device=dX.device)
        if nonblocking:
            obj_t = torch.zeros(nr, dtype=torch.float64, device=dX.device)
            stats_t = torch.zeros((I, 2), dtype=torch.int64, device=dX.device)
            with self._on_torch_stream():
                self.set_option("async", 1)
                try:
                    self._check(self._L.lsq_encode_icm_dev(self._h, dX.data_ptr(), dB0.data_ptr(), dK.data_ptr(), d, n, m, h,
                                                           ils.ctypes.data, nr, int(icmiter), int(npert), int(bool(randord)),
                                                           int(seed), int(global_offset), dBs.data_ptr(), obj_t.data_ptr(), stats_t.data_ptr()))
                finally:
                    self.set_option("async", 0)
            return dBs, obj_t, stats_t
        obj = np.zeros(nr, dtype=np.float64)
        stats = np.zeros((I, 2), dtype=np.int64)
        with self._on_torch_stream():
            self._check(self._L.lsq_encode_icm_dev(self._h, dX.data_ptr(), dB0.data_ptr(), dK.data_ptr(), d, n, m, h,
                                                   ils.ctypes.data, nr, int(icmiter), int(npert), int(bool(randord)),
                                                   int(seed), int(global_offset), dBs.data_ptr(), obj.ctypes.data,
                                                   stats.ctypes.data))
        return dBs, obj, stats

    # -- (1c) the search step after the path: ADC linear scan on the device ------------------
    def linscan(self, codes, Q, K, dbnorms, m, k, h=H):
        """codes (n,m) uint8 0-based, Q (nq,d), K (m*h,d), dbnorms (n,): host arrays.
        -> dists (nq,k) float32 ascending, ids (nq,k) int32 1-BASED   [lsq_linscan]"""
        codes, Q, K, dbn = _np(codes, np.uint8), _np(Q, np.float32), _np(K, np.float32), _np(dbnorms, np.float32)
        n, nq, d = codes.shape[0], Q.shape[0], Q.shape[1]
        if codes.shape != (n, m) or K.shape != (m * h, d) or dbn.shape != (n,):
            raise ValueError("shape mismatch: codes %s Q %s K %s dbnorms %s m=%d h=%d" % (codes.shape, Q.shape, K.shape, dbn.shape, m, h))
        dists = np.zeros((nq, k), dtype=np.float32)
        ids = np.zeros((nq, k), dtype=np.int32)
        self._check(self._L.lsq_linscan(self._h, dists.ctypes.data, ids.ctypes.data, codes.ctypes.data, Q.ctypes.data, K.ctypes.data,
                                        dbn.ctypes.data, nq, n, m, h, d, int(k)))
        return dists, ids

    def linscan_dev(self, dcodes, dQ, dK, dnorms, m, k, h=H):
        """The same on device-resident torch tensors -> (dists (nq,k) f32, ids (nq,k) int32 1-based) tensors   [lsq_linscan_dev]"""
        import torch
        assert dcodes.is_cuda and dQ.is_cuda and dK.is_cuda and dnorms.is_cuda, "device tensors required"
        assert dcodes.dtype == torch.uint8 and dQ.dtype == torch.float32 and dK.dtype == torch.float32 and dnorms.dtype == torch.float32
        assert dcodes.is_contiguous() and dQ.is_contiguous() and dK.is_contiguous() and dnorms.is_contiguous()
        n, (nq, d) = dcodes.shape[0], dQ.shape
        if dcodes.shape != (n, m) or dK.shape != (m * h, d) or dnorms.shape != (n,):
            raise ValueError("shape mismatch")
        dists = torch.empty((nq, k), dtype=torch.float32, device=dQ.device)
        ids = torch.empty((nq, k), dtype=torch.int32, device=dQ.device)
        with self._on_torch_stream():
            self._check(self._L.lsq_linscan_dev(self._h, dists.data_ptr(), ids.data_ptr(), dcodes.data_ptr(), dQ.data_ptr(), dK.data_ptr(),
                                                dnorms.data_ptr(), nq, n, m, h, d, int(k)))
        return dists, ids

    def quantize_norms(self, B, K, cbnorms, m, h=H):
        """B (n,m) int16 1-based, K (m*h,d), cbnorms (<= 256,): host arrays.
        -> idx (n,) int16 1-based index of the nearest norm centroid, dbnorms (n,) = cbnorms[idx-1], norms (n,) unquantised   [lsq_quantize_norms]"""
        B, K, cb = _np(B, np.int16), _np(K, np.float32), _np(cbnorms, np.float32).reshape(-1)
        n, d = B.shape[0], K.shape[1]
        if B.shape != (n, m) or K.shape != (m * h, d):
            raise ValueError("shape mismatch: B %s K %s m=%d h=%d" % (B.shape, K.shape, m, h))
        idx = np.zeros(n, dtype=np.int16)
        dbn = np.zeros(n, dtype=np.float32)
        nrm = np.zeros(n, dtype=np.float32)
        self._check(self._L.lsq_quantize_norms(self._h, B.ctypes.data, K.ctypes.data, cb.ctypes.data, cb.shape[0], d, n, m, h,
                                               idx.ctypes.data, dbn.ctypes.data, nrm.ctypes.data))
        return idx, dbn, nrm

    def quantize_norms_dev(self, dcodes, dK, dcb, m, h=H):
        """device tensors: codes (n,m) uint8 0-based, K (m*h,d) f32, cbnorms (<= 256,) f32 -> idx (n,) uint8 0-BASED, dbnorms (n,), norms (n,)   [lsq_quantize_norms_dev]"""
        import torch
        assert dcodes.is_cuda and dK.is_cuda and dcb.is_cuda and dcodes.dtype == torch.uint8 and dK.dtype == torch.float32 and dcb.dtype == torch.float32
        assert dcodes.is_contiguous() and dK.is_contiguous() and dcb.is_contiguous()
        n, d = dcodes.shape[0], dK.shape[1]
        if dcodes.shape != (n, m) or dK.shape != (m * h, d):
            raise ValueError("shape mismatch")
        idx = torch.empty(n, dtype=torch.uint8, device=dK.device)
        dbn = torch.empty(n, dtype=torch.float32, device=dK.device)
        nrm = torch.empty(n, dtype=torch.float32, device=dK.device)
        with self._on_torch_stream():
            self._check(self._L.lsq_quantize_norms_dev(self._h, dcodes.data_ptr(), dK.data_ptr(), dcb.data_ptr(), int(dcb.numel()), d, n, m, h,
                                                       idx.data_ptr(), dbn.data_ptr(), nrm.data_ptr()))
        return idx, dbn, nrm

    def update_codebooks(self, X, B, m, h=H):
        """X (n,d) f32, B (n,m) int16 1-based: host arrays -> K (m*h,d) least-squares codebooks, LSQR iterations   [lsq_update_codebooks_gpu]"""
        X, B = _np(X, np.float32), _np(B, np.int16)
        n, d = X.shape
        if B.shape != (n, m):
            raise ValueError("shape mismatch: X %s B %s m=%d" % (X.shape, B.shape, m))
        K = np.zeros((m * h, d), dtype=np.float32)
        it = C.c_int(0)
        self._check(self._L.lsq_update_codebooks_gpu(self._h, X.ctypes.data, B.ctypes.data, d, n, m, h, K.ctypes.data, C.byref(it)))
        return K, int(it.value)

    def update_codebooks_dev(self, dX, dcodes, m, h=H, out=None):
        """device tensors: X (n,d) f32, codes (n,m) uint8 0-based -> K (m*h,d) f32 tensor, LSQR iterations   [lsq_update_codebooks_dev]"""
        import torch
        assert dX.is_cuda and dcodes.is_cuda and dX.dtype == torch.float32 and dcodes.dtype == torch.uint8 and dX.is_contiguous() and dcodes.is_contiguous()
        n, d = dX.shape
        if dcodes.shape != (n, m):
            raise ValueError("shape mismatch")
        dK = out if out is not None else torch.empty((m * h, d), dtype=torch.float32, device=dX.device)
        it = C.c_int(0)
        with self._on_torch_stream():
            self._check(self._L.lsq_update_codebooks_dev(self._h, dX.data_ptr(), dcodes.data_ptr(), d, n, m, h, dK.data_ptr(), C.byref(it)))
        return dK, int(it.value)

    # -- the initialisers' data-parallel steps (csrc/lsq_init.hip) ----------------------------------
    def encode_viterbi(self, X, K, m, h=H):
        """X (n,d) f32, K (m*h,d) f32 (chain codebooks, zero outside their dimensions) -> B (n,m) int16 1-based: the exact chain optimum
        (encode_chain.jl:92-123)   [lsq_encode_viterbi]"""
        X, K = _np(X, np.float32), _np(K, np.float32)
        n, d = X.shape
        if K.shape != (m * h, d):
            raise ValueError("K must be (m*h, d) = (%d, %d), got %s" % (m * h, d, K.shape))
        B = np.empty((n, m), dtype=np.int16)
        self._check(self._L.lsq_encode_viterbi(self._h, X.ctypes.data, K.ctypes.data, d, n, m, h, B.ctypes.data))
        return B

    def encode_viterbi_dev(self, dX, dK, m, h=H):
        """device tensors -> codes (n,m) uint8 0-based   [lsq_encode_viterbi_dev]"""
        import torch
        assert dX.is_cuda and dK.is_cuda and dX.dtype == torch.float32 and dK.dtype == torch.float32 and dX.is_contiguous() and dK.is_contiguous()
        n, d = dX.shape
        if dK.shape != (m * h, d):
            raise ValueError("shape mismatch")
        dB = torch.empty((n, m), dtype=torch.uint8, device=dX.device)
        with self._on_torch_stream():
            self._check(self._L.lsq_encode_viterbi_dev(self._h, dX.data_ptr(), dK.data_ptr(), d, n, m, h, dB.data_ptr()))
        return dB

    def assign_codewords(self, X, K, m, h=H, want_min=False):
        """Per codebook independently the first argmin_a ||c||^2 - 2<x,c> (quantize_pq / the k-means assignment step; PQ.jl:12-41, kmeans.jl:6-75).
        X (n,d), K (m*h,d) -> B (n,m) int16 1-based [, the minima (n,m) f32]   [lsq_assign_codewords]"""
        X, K = _np(X, np.float32), _np(K, np.float32)
        n, d = X.shape
        if K.shape != (m * h, d):
            raise ValueError("K must be (m*h, d) = (%d, %d), got %s" % (m * h, d, K.shape))
        B = np.empty((n, m), dtype=np.int16)
        mv = np.empty((n, m), dtype=np.float32) if want_min else None
        self._check(self._L.lsq_assign_codewords(self._h, X.ctypes.data, K.ctypes.data, d, n, m, h, B.ctypes.data, mv.ctypes.data if want_min else None))
        return (B, mv) if want_min else B

    def assign_codewords_dev(self, dX, dK, m, h=H, want_min=False):
        """device tensors -> codes (n,m) uint8 0-based [, minima (n,m) f32]   [lsq_assign_codewords_dev]"""
        import torch
        assert dX.is_cuda and dK.is_cuda and dX.dtype == torch.float32 and dK.dtype == torch.float32 and dX.is_contiguous() and dK.is_contiguous()
        n, d = dX.shape
        if dK.shape != (m * h, d):
            raise ValueError("shape mismatch")
        dB = torch.empty((n, m), dtype=torch.uint8, device=dX.device)
        dmin = torch.empty((n, m), dtype=torch.float32, device=dX.device) if want_min else None
        with self._on_torch_stream():
            self._check(self._L.lsq_assign_codewords_dev(self._h, dX.data_ptr(), dK.data_ptr(), d, n, m, h, dB.data_ptr(), dmin.data_ptr() if want_min else None))
        return (dB, dmin) if want_min else dB

    def linscan_stats(self):
        t = _lib.LinscanStats()
        self._check(self._L.lsq_get_linscan_stats(self._h, C.byref(t)))
        return t.as_dict()

    # -- (2) CPU-path shaped ---------------------------------------------------------------
    def encoding_icm(self, X, oldB, K, m, niter, randord, npert, seed=0, it=None, global_offset=0, h=H):
        """ONE ILS iteration with the accept rule.  it=None (default): the context's own counter (LSQ_IT_AUTO) -- the k-th call uses it = k-1."""
        it = IT_AUTO if it is None else it
        X, K, oldB = _np(X, np.float32), _np(K, np.float32), _np(oldB, np.int16)
        n, d = X.shape
        self._check_shapes(X, K, oldB, m, h)
        out = np.empty((n, m), dtype=np.int16)
        self._check(self._L.lsq_encoding_icm(self._h, X.ctypes.data, oldB.ctypes.data, K.ctypes.data, d, n, m, h,
                                            int(niter), int(bool(randord)), int(npert), int(seed), int(it),
                                            int(global_offset), out.ctypes.data))
        return out

    def encode_icm_fully(self, B, X, K, m, niter, randord, npert, idx_first=1, seed=0, it=None, h=H):
        """In place on B (n, m) int16 (must be C-contiguous int16).  it=None: the context's counter, as in encoding_icm."""
        it = IT_AUTO if it is None else it
        X, K = _np(X, np.float32), _np(K, np.float32)
        if B.dtype != np.int16 or not B.flags["C_CONTIGUOUS"]:
            raise ValueError("B must be a C-contiguous int16 (n, m) array (it is updated in place)")
        n, d = X.shape
        self._check_shapes(X, K, B, m, h)
        self._check(self._L.lsq_encode_icm_fully(self._h, B.ctypes.data, X.ctypes.data, K.ctypes.data, d, n, m, h,
                                                int(niter), int(bool(randord)), int(npert), int(idx_first), int(seed), int(it)))
        return B

    # -- (3) helpers -----------------------------------------------------------------------
    def get_unaries(self, X, K, m, h=H):
        X, K = _np(X, np.float32), _np(K, np.float32)
        n, d = X.shape
        U = np.empty((m, n, h), dtype=np.float32)
        self._check(self._L.lsq_get_unaries(self._h, X.ctypes.data, K.ctypes.data, d, n, m, h, U.ctypes.data))
        return U

    def get_binaries(self, K, m, h=H):
        K = _np(K, np.float32)
        d = K.shape[1]
        T = np.empty((m, m, h, h), dtype=np.float32)
        self._check(self._L.lsq_get_binaries(self._h, K.ctypes.data, d, m, h, T.ctypes.data))
        return T

    def veccost(self, X, B, K, m, h=H):
        X, K, B = _np(X, np.float32), _np(K, np.float32), _np(B, np.int16)
        n, d = X.shape
        self._check_shapes(X, K, B, m, h)
        out = np.empty(n, dtype=np.float32)
        self._check(self._L.lsq_veccost(self._h, X.ctypes.data, B.ctypes.data, K.ctypes.data, d, n, m, h, out.ctypes.data))
        return out

    def qerror(self, X, B, K, m, h=H):
        X, K, B = _np(X, np.float32), _np(K, np.float32), _np(B, np.int16)
        n, d = X.shape
        self._check_shapes(X, K, B, m, h)
        out = C.c_double(0.0)
        self._check(self._L.lsq_qerror(self._h, X.ctypes.data, B.ctypes.data, K.ctypes.data, d, n, m, h, C.byref(out)))
        return float(out.value)

    def perturb(self, B, npert, seed=0, it=0, global_offset=0, h=H):
        B = _np(B, np.int16).copy()
        n, m = B.shape
        self._check(self._L.lsq_perturb(self._h, B.ctypes.data, n, m, h, int(npert), int(seed), int(it), int(global_offset)))
        return B

    # -- (4) device generators -------------------------------------------------------------
    def synth_data_u8_dev(self, seed, n, d, device=None, global_offset=0):
        import torch
        X = torch.empty((n, d), dtype=torch.float32, device=device or ("cuda:%d" % self.device))
        with self._on_torch_stream():
            self._check(self._L.lsq_synth_data_u8_dev(self._h, int(seed), int(global_offset), n, d, X.data_ptr()))
        return X

    def randinit_dev(self, seed, n, m, device=None, global_offset=0, h=H):
        import torch
        B = torch.empty((n, m), dtype=torch.uint8, device=device or ("cuda:%d" % self.device))
        with self._on_torch_stream():
            self._check(self._L.lsq_randinit_dev(self._h, int(seed), int(global_offset), n, m, h, B.data_ptr()))
        return B

    def synth_codebooks_dev(self, seed, m, d, device=None, h=H):
        import torch
        K = torch.empty((m * h, d), dtype=torch.float32, device=device or ("cuda:%d" % self.device))
        with self._on_torch_stream():
            self._check(self._L.lsq_synth_codebooks_dev(self._h, int(seed), m, h, d, K.data_ptr()))
        return K

    @staticmethod
    def _check_shapes(X, K, B, m, h):
        n, d = X.shape
        if K.shape != (m * h, d):
            raise ValueError("K must be (m*h, d) = (%d, %d), got %s" % (m * h, d, K.shape))
        if B.shape != (n, m):
            raise ValueError("B must be (n, m) = (%d, %d), got %s" % (n, m, B.shape))


# -- host-only pieces of the path (no GPU needed) ---------------------------------------------

def randinit(n, m, h=H, seed=0, global_offset=0):
    """initializations.jl:2-8 -> (n, m) int16 1-based (Philox-keyed, shard-invariant)."""
    B = np.empty((n, m), dtype=np.int16)
    _lib.check(_lib.load().lsq_randinit(int(seed), int(global_offset), n, m, h, B.ctypes.data))
    return B


def node_order(seed, it, m, randord=True):
    o = np.empty(m, dtype=np.int32)
    _lib.check(_lib.load().lsq_node_order(int(seed), int(it), m, int(bool(randord)), o.ctypes.data))
    return o


class MultiEngine:
    """Single-process multi-GPU handle (lsq_multi_*): one context + host thread per listed device, `splitarray` shards,
    results identical to the one-device call.  `encode_icm` has Engine.encode_icm's signature, so the reference-shaped
    functions accept it as `engine=`:  encode_icm_cuda(RX, B, C, ..., engine=MultiEngine([0, 1, 2, 3]))."""

    def __init__(self, devices, **options):
        self._L = _lib.load()
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        h = C.c_void_p()
        _lib.check(self._L.lsq_multi_create(C.byref(h), devs.ctypes.data, int(devs.shape[0])))
        self._h = h
        self.devices = [int(x) for x in devs]
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key, value):
        _lib.check(self._L.lsq_multi_set_option(self._h, key.encode(), int(value)))

    def encode_icm(self, X, B, K, m, ilsiters, icmiter, npert, randord, seed=0, nsplits=1, global_offset=0,
                   verbose=False, h=H):
        """-> Bs (nr, n, m) int16 1-based, objs (nr,) float32   [lsq_multi_encode_icm]"""
        X, K, B = _np(X, np.float32), _np(K, np.float32), _np(B, np.int16)
        n, d = X.shape
        if B.shape != (n, m) or K.shape != (m * h, d):
            raise ValueError("shape mismatch: X %s B %s K %s m=%d h=%d" % (X.shape, B.shape, K.shape, m, h))
        ils = _np(ilsiters, np.int64).reshape(-1)
        nr = ils.shape[0]
        Bs = np.empty((nr, n, m), dtype=np.int16)
        objs = np.zeros(nr, dtype=np.float32)
        _lib.check(self._L.lsq_multi_encode_icm(self._h, X.ctypes.data, B.ctypes.data, K.ctypes.data, d, n, m, h,
                                                ils.ctypes.data, nr, int(icmiter), int(npert), int(bool(randord)),
                                                int(seed), int(global_offset), int(bool(verbose)), Bs.ctypes.data, objs.ctypes.data))
        return Bs, objs

    def linscan(self, codes, Q, K, dbnorms, m, k, h=H):
        """the ADC scan over a database sharded across the devices; Engine.linscan's signature and results   [lsq_multi_linscan]"""
        codes, Q, K, dbn = _np(codes, np.uint8), _np(Q, np.float32), _np(K, np.float32), _np(dbnorms, np.float32)
        n, nq, d = codes.shape[0], Q.shape[0], Q.shape[1]
        if codes.shape != (n, m) or K.shape != (m * h, d) or dbn.shape != (n,):
            raise ValueError("shape mismatch: codes %s Q %s K %s dbnorms %s m=%d h=%d" % (codes.shape, Q.shape, K.shape, dbn.shape, m, h))
        dists = np.zeros((nq, k), dtype=np.float32)
        ids = np.zeros((nq, k), dtype=np.int32)
        _lib.check(self._L.lsq_multi_linscan(self._h, dists.ctypes.data, ids.ctypes.data, codes.ctypes.data, Q.ctypes.data, K.ctypes.data,
                                             dbn.ctypes.data, nq, n, m, h, d, int(k)))
        return dists, ids

    def close(self):
        if self._h is not None and self._h.value:
            self._L.lsq_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def splitarray(n, nparts):
    """utils.jl:152-177 -> list of (start, stop) 0-based half-open ranges."""
    L = _lib.load()
    out = []
    for p in range(nparts):
        s, ln = C.c_int64(), C.c_int64()
        _lib.check(L.lsq_splitarray(n, nparts, p, C.byref(s), C.byref(ln)))
        out.append((s.value, s.value + ln.value))
    return out


def device_count():
    c = C.c_int(0)
    rc = _lib.load().lsq_device_count(C.byref(c))
    return c.value if rc == 0 else 0
