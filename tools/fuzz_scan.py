"""tools/fuzz_scan.py [cases] [seed] -- offline randomised campaign for the device ADC scan against this library's host scan (itself pinned to the
reference build by tests/test_linscan.py): any m, odd d, ragged query tiles, ties, +inf norms, nn from 1 to n, both selection roads, and the
threshold rank deliberately wrong in a third of the cases (fallback road).  Exit code 1 on a mismatch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lsq = importlib.import_module("local-search-quantization_amd")
H = 256
def run(ncases=100, seed0=1, verbose=True):
    """-> (mismatching cases, summary line)"""
    L = lsq._lib.load()
    bad = fallbacks = 0
    for t in range(ncases):
        rng = np.random.default_rng(seed0 * 7919 + t)
        m = int(rng.integers(1, 17)); d = int(rng.choice([1, 3, 8, 17, 32, 100, 128, 130]))
        big = bool(rng.integers(2))
        n = int(rng.integers(66_000, 400_000)) if big else int(rng.integers(1, 20_000))
        nq = int(rng.integers(1, 100))
        knn = int(min(n, rng.choice([1, 3, 10, 100, 1000, 5000]))) if big else int(rng.integers(1, n + 1))
        K = (rng.standard_normal((m * H, d)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        codes = rng.integers(0, H, size=(n, m), dtype=np.uint8)
        dbn = (rng.random(n) * 10.0 ** rng.integers(-2, 5)).astype(np.float32)
        if t % 4 == 0 and n > 1:
            codes[n // 2:] = codes[: n - n // 2]; dbn[n // 2:] = dbn[: n - n // 2]          # ties
        if t % 5 == 0:
            dbn[rng.integers(0, n, size=max(1, n // 50))] = np.inf                           # entries that can never be near
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        hd = np.zeros((nq, knn), np.float32); hi = np.zeros((nq, knn), np.int32)
        lsq._lib.check(L.lsq_linscan_aqd_query_extra_byte(hd.ctypes.data, hi.ctypes.data, codes.ctypes.data, Q.ctypes.data, K.ctypes.data, dbn.ctypes.data,
                                                          nq, n, m, H, d, knn, 0))
        with lsq.Engine(0) as eng:
            if t % 3 == 2:
                eng.set_option("linscan_rank", int(rng.integers(1, 4)))                     # a useless threshold: the fallback road
            dd, di = eng.linscan(codes, Q, K, dbn, m, knn)
            st = eng.linscan_stats()
        fallbacks += st["fallback_queries"]
        if not (np.array_equal(di, hi) and np.array_equal(dd.view(np.uint32), hd.view(np.uint32))):
            bad += 1
            print("MISMATCH case %d: n=%d nq=%d d=%d m=%d knn=%d: %d ids differ; %r" % (t, n, nq, d, m, knn, int((di != hi).sum()), st), flush=True)
    summary = "fuzz_scan: %d cases, %d mismatches, %d queries went through the fallback road" % (ncases, bad, fallbacks)
    if verbose:
        print(summary)
    return bad, summary


if __name__ == "__main__":
    nbad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(1 if nbad else 0)
