"""ctypes binding of liblsq_mi355x.so -- every symbol include/lsq_mi355x.h declares.

There is deliberately NO fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
(or `make -C local-search-quantization_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSQ_LIB_PATH") or os.path.join(_HERE, "liblsq_mi355x.so")      # override: A/B builds of the same ABI
# Same ABI built with -DLSQ_TUNING: + option "ablation", environment knobs, clock stamps.  Profiling tools and the tests
# that cross-check the earlier schedules load it explicitly (Engine(..., tuning=True)); the product path never does.
TUNING_LIB_PATH = os.environ.get("LSQ_TUNING_LIB_PATH") or os.path.join(_HERE, "liblsq_mi355x_tuning.so")

LSQ_OK, LSQ_EINVAL, LSQ_EHIP, LSQ_ENOMEM, LSQ_ECODE, LSQ_ENODEV = 0, -1, -2, -3, -4, -5


class LsqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liblsq_mi355x error %d: %s" % (code, msg))
        self.code = code


class Timings(C.Structure):
    _fields_ = [("tables_ms", C.c_double), ("unaries_ms", C.c_double), ("perturb_ms", C.c_double),
                ("icm_ms", C.c_double), ("cost_ms", C.c_double), ("other_ms", C.c_double),
                ("icm_launches", C.c_int64), ("icm_node_updates", C.c_int64),
                ("staged_blocks", C.c_int64), ("light_blocks", C.c_int64), ("filtered_blocks", C.c_int64),
                ("filter_refined", C.c_int64), ("filter_exact", C.c_int64), ("filter_f32", C.c_int64),
                ("filter_fallback_chunks", C.c_int64), ("xs_launches", C.c_int64), ("xs_fallback_launches", C.c_int64),
                ("table_reuses", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LinscanStats(C.Structure):
    _fields_ = [("queries", C.c_int64), ("codes", C.c_int64), ("candidates", C.c_int64), ("fallback_queries", C.c_int64),
                ("batches", C.c_int64), ("exhaustive", C.c_int64), ("threshold_rank", C.c_int64), ("list_capacity", C.c_int64),
                ("lut_ms", C.c_double), ("sample_ms", C.c_double), ("scan_ms", C.c_double), ("select_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_vp, _i, _i64, _u64, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32

# name -> (restype, argtypes).  Pointers are passed as raw addresses (host or device).
SIGNATURES = {
    "lsq_last_error": (C.c_char_p, []),
    "lsq_version": (_i, []),
    "lsq_device_count": (_i, [C.POINTER(_i)]),
    "lsq_create": (_i, [C.POINTER(_vp), _i]),
    "lsq_destroy": (_i, [_vp]),
    "lsq_set_stream": (_i, [_vp, _vp]),
    "lsq_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "lsq_get_timings": (_i, [_vp, C.POINTER(Timings)]),
    "lsq_get_timings_sized": (_i, [_vp, _vp, C.c_size_t]),
    "lsq_get_walk_trace": (_i, [_vp, _vp, _i]),
    "lsq_reset_timings": (_i, [_vp]),
    "lsq_synchronize": (_i, [_vp]),
    "lsq_multi_create": (_i, [C.POINTER(_vp), _vp, _i]),
    "lsq_multi_destroy": (_i, [_vp]),
    "lsq_multi_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "lsq_multi_encode_icm": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _i, _i, _i, _i, _u64, _u64, _i, _vp, _vp]),
    "lsq_encode_icm": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _i, _i, _i, _i, _i, _u64, _u64, _i, _vp, _vp]),
    "lsq_encode_icm_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _i, _i, _i, _i, _u64, _u64, _vp, _vp, _vp]),
    "lsq_encoding_icm": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _u64, _u32, _u64, _vp]),
    "lsq_encode_icm_fully": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _i64, _u64, _u32]),
    "lsq_get_unaries": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "lsq_get_binaries": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "lsq_veccost": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "lsq_qerror": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, C.POINTER(C.c_double)]),
    "lsq_perturb": (_i, [_vp, _vp, _i64, _i, _i, _i, _u64, _u32, _u64]),
    "lsq_randinit": (_i, [_u64, _u64, _i64, _i, _i, _vp]),
    "lsq_node_order": (_i, [_u64, _u32, _i, _i, _vp]),
    "lsq_splitarray": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "lsq_linscan_aqd_query_extra_byte": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "lsq_linscan": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "lsq_linscan_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "lsq_multi_linscan": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "lsq_get_linscan_stats": (_i, [_vp, C.POINTER(LinscanStats)]),
    "lsq_quantize_norms": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "lsq_quantize_norms_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "lsq_update_codebooks": (_i, [_vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "lsq_update_codebooks_lsmr": (_i, [_vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "lsq_update_codebooks_gpu": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, C.POINTER(_i)]),
    "lsq_update_codebooks_dev": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, C.POINTER(_i)]),
    "lsq_encode_viterbi": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "lsq_encode_viterbi_dev": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "lsq_assign_codewords": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "lsq_assign_codewords_dev": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "lsq_synth_data_u8_dev": (_i, [_vp, _u64, _u64, _i64, _i, _vp]),
    "lsq_randinit_dev": (_i, [_vp, _u64, _u64, _i64, _i, _i, _vp]),
    "lsq_synth_codebooks_dev": (_i, [_vp, _u64, _i, _i, _i, _vp]),
}

_libs = {}


def load(tuning=False):
    """Load the shared library (once per flavour).  Raises if it has not been built."""
    path = TUNING_LIB_PATH if tuning else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
                "g.build()\"` -- there is no CPU fallback." % path)
        # torch bundles its own libamdhip64.so / libhsa-runtime64.so (SONAME libamdhip64.so.7).  Two HIP
        # runtimes in one process cannot both own the GPU ("No HIP GPUs are available" in whichever
        # initialises second), so when torch is installed it must be loaded FIRST: our DT_NEEDED
        # libamdhip64.so.7 then resolves to the copy already in the process.  A pure-C consumer (the
        # Julia ccall shim of INTEGRATION.md) simply gets /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is missing: loud by design
            fn.restype = res
            fn.argtypes = args
        _libs[path] = lib
    return _libs[path]


def check(rc, lib=None):
    if rc != LSQ_OK:
        msg = (lib or load()).lsq_last_error()      # thread-local inside the library that failed
        raise LsqError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
