"""Determinism soak: repeat the same encode many times and require bit-identical codes / objective sums every time.
Exercises the LDS atomics, slice barriers, light-block path and chunk prefetch for rare races.  python tools/soak.py [reps]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lsq = importlib.import_module("local-search-quantization_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [  # n, d, m, ils, light
    (1_000_000, 128, 8, 4, -1), (300_000, 128, 16, 2, -1), (120_000, 960, 8, 2, -1), (50_000, 64, 5, 4, -1), (200_000, 128, 8, 4, 4096),
    (2_500_000, 32, 8, 2, -1),
]
bad = 0
for n, d, m, ils, light in cases:
    with lsq.Engine(0) as eng:
        if light >= 0:
            eng.set_option("light", light)
        dX = eng.synth_data_u8_dev(5, n, d)
        dB0 = eng.randinit_dev(6, n, m)
        dK = eng.synth_codebooks_dev(7, m, d)
        ref, sums0, st0 = eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=3)
        ref = ref.clone()
        for r in range(reps):
            out, sums, st = eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=3)
            if not torch.equal(out, ref) or not np.array_equal(sums, sums0) or not np.array_equal(st, st0):
                bad += 1
                print("MISMATCH case", (n, d, m, ils, light), "rep", r, int((out != ref).sum().item()), "codes differ")
    print("case n=%d d=%d m=%d ils=%d light=%d: %d reps identical" % (n, d, m, ils, light, reps), flush=True)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
