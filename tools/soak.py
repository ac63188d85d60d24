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

# the device ADC scan and the device LSQR: candidate lists / column sums are filled by atomics in a different order every time -- the answers must not notice
rng = np.random.default_rng(1)
H = 256
for n, nq, d, m, knn in [(1_000_000, 2000, 128, 8, 100), (200_000, 500, 32, 16, 1000), (70_000, 300, 16, 7, 10)]:
    K = (rng.standard_normal((m * H, d)) * 0.5).astype(np.float32)
    codes = rng.integers(0, H, size=(n, m), dtype=np.uint8)
    codes[n // 2:] = codes[: n - n // 2]                       # ties
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    dev = torch.device("cuda:0")
    dK, dC, dQ = torch.from_numpy(K).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(Q).to(dev)
    dN = torch.rand(n, device=dev) * 10
    dN[n // 2:] = dN[: n - n // 2]
    with lsq.Engine(0) as eng:
        d0, i0 = eng.linscan_dev(dC, dQ, dK, dN, m, knn)
        d0, i0 = d0.clone(), i0.clone()
        for r in range(reps):
            d1, i1 = eng.linscan_dev(dC, dQ, dK, dN, m, knn)
            if not torch.equal(i1, i0) or not torch.equal(d1, d0):
                bad += 1
                print("SCAN MISMATCH n=%d m=%d knn=%d rep %d" % (n, m, knn, r))
        print("scan n=%d nq=%d m=%d knn=%d: %d repetitions identical" % (n, nq, m, knn, reps))
for n, d, m in [(100_000, 128, 8), (30_000, 33, 5)]:
    X = rng.standard_normal((n, d)).astype(np.float32)
    B = rng.integers(0, H, size=(n, m), dtype=np.uint8)
    with lsq.Engine(0) as eng:
        dX, dB = torch.from_numpy(X).cuda(), torch.from_numpy(B).cuda()
        K0, it0 = eng.update_codebooks_dev(dX, dB, m)
        K0 = K0.clone()
        diff = 0
        for r in range(reps):
            K1, it1 = eng.update_codebooks_dev(dX, dB, m)
            if it1 != it0 or not torch.equal(K1, K0):
                diff += 1
        print("lsqr n=%d d=%d m=%d: %d of %d repetitions differ in some bit (fixed order of addition: none expected)" % (n, d, m, diff, reps))
print("soak (scan + lsqr) done, scan mismatches so far:", bad)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
