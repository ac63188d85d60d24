"""local-search-quantization_amd: MI355X (gfx950)-native LSQ encoding engine.

Scope: the ILS/ICM encoding hot path of
una-dinosauria/local-search-quantization -- unary-table build, pair tables, perturbation, ICM
sweeps, cost + accept -- as hand-written HIP behind a C-ABI (include/lsq_mi355x.h), plus the
host-side mirror of the reference's operator interface for that path.

    from importlib import import_module
    lsq = import_module("local-search-quantization_amd")      # or: import lsq_amd
    Bs, objs = lsq.encode_icm_cuda(RX, B, C, [16], 4, 4, True, 2, False, seed=42)
"""
from . import _lib  # noqa: F401
from .engine import Engine, MultiEngine, randinit as randinit_rows, node_order, splitarray as split_ranges, device_count  # noqa: F401
from .reference_api import (  # noqa: F401
    encode_icm_cuda, encoding_icm, encode_icm_fully, get_unaries, get_binaries, veccost, qerror,
    randinit, splitarray, default_engine, linscan_lsq, eval_recall, quantize_norms, reconstruct,
    fvecs_read, ivecs_read, bvecs_read, update_codebooks, train_lsq, train_lsq_dev,
)
from .initializers import (  # noqa: F401
    train_pq, quantize_pq, train_opq, quantize_opq, train_chainq, encoding_viterbi, update_codebooks_chain, get_cbdims_chain,
)
from . import distributed  # noqa: F401

__all__ = [
    "Engine", "MultiEngine", "encode_icm_cuda", "encoding_icm", "encode_icm_fully", "get_unaries", "get_binaries",
    "veccost", "qerror", "randinit", "splitarray", "node_order", "device_count", "distributed", "linscan_lsq", "eval_recall",
    "quantize_norms", "reconstruct", "update_codebooks", "train_lsq", "train_lsq_dev", "train_pq", "quantize_pq", "train_opq", "quantize_opq",
    "train_chainq", "encoding_viterbi", "update_codebooks_chain", "get_cbdims_chain", "fvecs_read", "ivecs_read", "bvecs_read",
]
