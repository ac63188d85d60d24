"""tools/fuzz_filter.py [cases] [seed] -- offline randomised campaign for the filtered walk (beyond the fixed 40 cases of tests/test_gpu_random_shapes.py):
random shapes, scales, offsets (large common components: the shift-invariant levels), duplicated / near-duplicated codewords, forced filtering
(q16_min = 0, light = 0, probe off), every vector compared with the oracle.  Prints one line per failing case and a summary; exit code 1 on a mismatch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
lsq = importlib.import_module("local-search-quantization_amd")
import oracle as O
O.build()
H = 256
def run(ncases=60, seed0=1, schedule=6, verbose=True):
    """-> (mismatching cases, summary line).  tests/test_gpu_depth.py runs a fixed-seed slice of it under pytest."""
    bad = 0
    stats = dict(refined=0, f32=0, updates=0)
    for t in range(ncases):
        rng = np.random.default_rng(seed0 * 100003 + t)
        m = int(rng.integers(2, 17))
        d = int(rng.choice([2, 5, 8, 16, 24, 32, 48]))
        n = int(rng.choice([3000, 5000, 9000, 20000, 40000, 70000]))      # above 32 768 vectors the level range comes from a SAMPLE: out-of-range vectors get flagged
        if m > 8:
            n = min(n, 6000) if n < 40000 else 36000
        scale = float(10.0 ** rng.integers(-6, 7))
        X = (rng.standard_normal((n, d)) * scale).astype(np.float32)
        K = (rng.standard_normal((m * H, d)) * scale / m).astype(np.float32)
        kind = int(rng.integers(5))
        if kind == 1:                                             # large common component in data and codebooks
            u = rng.standard_normal(d).astype(np.float32); u /= np.linalg.norm(u)
            off = np.float32(scale * 10.0 ** rng.integers(1, 5))
            X = X + off * u; K = K + (off / m) * u
        elif kind == 2:                                           # duplicated and nearly duplicated codewords (ties, near-ties)
            Kr = K.reshape(m, H, d)
            Kr[:, 1::2] = Kr[:, 0::2]
            Kr[m // 2, 1::2] *= np.float32(1 + 1e-6)
        elif kind == 3:                                           # one codebook much larger than the others (table ranges dominated by a few pairs)
            K.reshape(m, H, d)[int(rng.integers(m))] *= np.float32(30.0)
        elif kind == 4:                                           # heavy-tailed vector norms
            X = X * rng.standard_cauchy((n, 1)).astype(np.float32)
        X = np.ascontiguousarray(X, dtype=np.float32); K = np.ascontiguousarray(K, dtype=np.float32)
        B0 = O.randinit(3000 + t, n, m, H)
        ils, J, npert = [int(rng.integers(1, 3))], int(rng.integers(1, 4)), int(rng.integers(0, m + 1))
        ref, objs_ref = O.encode_icm(X, B0, K, m, H, ils, J, npert, True, 11 * t + 3)
        with lsq.Engine(0, schedule=schedule) as eng:
            eng.set_option("q16_min", 0); eng.set_option("light", 0); eng.set_option("filter_probe_div", 0); eng.set_option("filter_fallback_div", 0)
            Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=11 * t + 3)
            tm = eng.timings()
        ok = np.array_equal(Bs, ref)
        stats["refined"] += tm["filter_refined"]; stats["f32"] += tm["filter_f32"]; stats["updates"] += tm["icm_node_updates"]
        if not ok:
            bad += 1
            print("MISMATCH case %d: m=%d d=%d n=%d kind=%d scale=%g: %d codes differ; %r" % (t, m, d, n, kind, scale, int((Bs != ref).sum()), tm), flush=True)
    summary = "fuzz: %d cases, %d mismatches; node updates %d, refined exactly %d (%.2f %%), sent to f32 %d (%.2f %%)" % (
        ncases, bad, stats["updates"], stats["refined"], 100.0 * stats["refined"] / max(stats["updates"], 1), stats["f32"], 100.0 * stats["f32"] / max(stats["updates"], 1))
    if verbose:
        print(summary)
    return bad, summary


if __name__ == "__main__":
    nbad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(os.environ.get("LSQ_FUZZ_SCHEDULE", "6")))
    sys.exit(1 if nbad else 0)
