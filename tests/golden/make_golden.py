#!/usr/bin/env python
"""Generates tests/golden/*.npz with the CPU oracle (oracle/lsq_oracle.c).

The reference (Julia 0.6 + CUDA) cannot run here and ships no golden vectors of its own
(SURVEY.md F2/F4), so these fixtures pin the BUILD's canonical outputs: inputs and expected
outputs only -- data, no source.  Re-run after any deliberate change of the canonical choices:

    python tests/golden/make_golden.py

Each fixture: integer-valued data X (uint8), codeword pool P (uint8; K = P / m in float32),
initial codes B0 (1-based int16), parameters, and the expected per-checkpoint codes Bs, objectives
objs, ==/< counters stats, plus spot rows of the unary / pair tables and the initial costs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

H = 256
OUT = os.path.dirname(os.path.abspath(__file__))

# (name, d, n, m, ilsiters, icmiter, npert, randord, seed)      -- SURVEY 8(c) shapes, small d
CASES = [
    ("g1_d16_m4", 16, 64, 4, [1, 2], 4, 2, 1, 1),
    ("g2_d32_m7", 32, 64, 7, [2], 4, 4, 1, 2),
    ("g3_d32_m8", 32, 64, 8, [1, 2, 4], 4, 4, 1, 3),
    ("g4_d16_m16", 16, 32, 16, [2], 4, 4, 1, 4),
    ("g5_d64_m8_seq", 64, 16, 8, [1], 4, 4, 0, 5),
    ("g6_d32_m8_ties", 32, 48, 8, [2], 2, 3, 1, 6),          # duplicate codewords: exact ties (P3)
]


def build_case(name, d, n, m, ils, J, npert, randord, seed):
    X8 = (O.synth_data_u8(100 + seed, n, d)).astype(np.uint8)
    P8 = (O.synth_data_u8(200 + seed, m * H, d)).astype(np.uint8)
    if name.endswith("ties"):
        P8 = P8.reshape(m, H, d).copy()
        P8[:, 1::2] = P8[:, 0::2]
        P8 = P8.reshape(m * H, d)
    X = X8.astype(np.float32)
    K = (P8.astype(np.float32) / np.float32(m))
    B0 = O.randinit(300 + seed, n, m, H)
    Bs, objs, stats = O.encode_icm(X, B0, K, m, H, ils, J, npert, bool(randord), seed, global_offset=0, want_stats=True)
    U = O.unaries(X, K, m, H)
    T = O.tables(K, m, H)
    cost0 = O.veccost(X, K, (B0 - 1).astype(np.uint8), H)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        X8=X8, P8=P8, B0=B0, params=np.array([d, n, m, H, J, npert, randord, seed], dtype=np.int64),
        ilsiters=np.array(ils, dtype=np.int64), Bs=Bs, objs=objs, stats=stats.astype(np.int64),
        U_rows=U[:, :4, :].copy(),                 # unary rows of the first 4 vectors, all codebooks
        T_rows=T[0, m - 1, :4, :].copy() if m > 1 else np.zeros((0, H), np.float32),   # 4 columns of pair (0, m-1)
        cost0=cost0,
    )
    print(name, "objs", objs, "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


if __name__ == "__main__":
    O.build()
    for c in CASES:
        build_case(*c)
