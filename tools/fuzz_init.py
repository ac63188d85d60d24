"""tools/fuzz_init.py [cases] [seed] -- run ON THE GPU BOX.  Randomised campaign for the initialisers' device kernels (csrc/lsq_init.hip): random shapes
(m = 1 .. 16, d = 4 .. 200, ragged n), scales 1e-4 .. 1e4, common offsets, duplicated codewords, integer-valued data (exact ties), resident chunks smaller than n;
every vector's Viterbi codes and nearest-codeword codes + minima against oracle/init_oracle.py, bit for bit."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.init_oracle as ini
lsq = importlib.import_module("local-search-quantization_amd")
H = 256


def run(ncases=100, seed=1, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(ncases):
        m = int(rng.integers(1, 17)); d = int(rng.integers(1, 51)) * 4; n = int(rng.integers(1, 400))
        kind = int(rng.integers(0, 5)); scale = float(10.0 ** rng.integers(-4, 5))
        X = (rng.standard_normal((n, d)) * scale).astype(np.float32)
        K = (rng.standard_normal((m * H, d)) * scale / max(m, 1)).astype(np.float32)
        if kind == 1:                                              # a large component every vector and codeword share
            u = rng.standard_normal(d).astype(np.float32); off = np.float32(scale * 100.0)
            X = X + off * u; K = K + (off / m) * u
        elif kind == 2:                                            # duplicated codewords: first copy must win
            Kr = K.reshape(m, H, d); Kr[:, 1::2] = Kr[:, 0::2]
        elif kind == 3:                                            # small integers: exact sums, many exact ties
            X = rng.integers(-3, 4, size=(n, d)).astype(np.float32); K = rng.integers(-2, 3, size=(m * H, d)).astype(np.float32)
        elif kind == 4 and m > 1:                                  # chain / PQ structure: codebooks zero outside their dimensions
            od = ini.get_cbdims_chain(d, m) if d >= m - 1 and m >= 2 and d // max(m - 1, 1) >= 1 else None
            if od is not None:
                Kr = K.reshape(m, H, d)
                for i in range(m):
                    mask = np.ones(d, dtype=bool); mask[od[i]] = False; Kr[i][:, mask] = 0
        X = np.ascontiguousarray(X, dtype=np.float32); K = np.ascontiguousarray(K, dtype=np.float32)
        chunk = None if rng.random() < 0.5 else int(rng.integers(1, n + 1))
        with lsq.Engine(0, chunk=chunk) as eng:
            wa, wmin = ini.assign_codewords_exact(X, K, m, H)
            B, mv = eng.assign_codewords(X, K, m, want_min=True)
            ok = np.array_equal(B.astype(np.int64) - 1, wa) and np.array_equal(mv, wmin)
            if m >= 2:
                wv = ini.encoding_viterbi_exact(X, K, m, H)
                V = eng.encode_viterbi(X, K, m)
                ok = ok and np.array_equal(V.astype(np.int64) - 1, wv)
        if not ok:
            bad += 1
            print("MISMATCH case %d: m=%d d=%d n=%d kind=%d scale=%g chunk=%r" % (t, m, d, n, kind, scale, chunk), flush=True)
    summary = "fuzz_init: %d cases, %d mismatches" % (ncases, bad)
    if verbose:
        print(summary)
    return bad, summary


if __name__ == "__main__":
    nbad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(1 if nbad else 0)
