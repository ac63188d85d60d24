"""Randomised parity sweep: shapes, iteration counts and options drawn from a fixed seed, every case compared with the
oracle on all vectors (codes bit-exact, objective 1e-5).  Complements the hand-picked cases of test_gpu_parity.py: odd
dimensions, every m in 1..16, n around the block / pass / light-block boundaries, multi-snapshot calls, offsets.

Which node-update path a case exercises is CONTROLLED and ASSERTED (all 40 cases used to run the L2-gather path):
cases with t % 3 == 0 force staging (option "light" = 0), cases with t % 3 == 1 use large n (66 000 / 140 000, one chunk, at least
one sweep) so that blocks stage naturally, the rest keep the defaults (mostly light blocks).  The device counters of
`lsq_timings` must agree with the intent."""
import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu
H = 256


def _cases():
    rng = np.random.default_rng(20260928)
    out = []
    n_pool = [1, 2, 63, 64, 65, 255, 256, 257, 300, 1000, 4095, 4096, 4097, 9000, 66_000, 70_001]
    for t in range(40):
        m = int(rng.integers(1, 17))
        d = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 100, 127, 128, 129, 200, 257]))
        n = int(rng.choice(n_pool))
        if n > 10_000 and d > 64:
            d = 16                                     # keep the oracle side in seconds
        nr = int(rng.integers(1, 4))
        ils = sorted(int(x) for x in rng.choice(np.arange(1, 5), size=nr, replace=False))
        J = int(rng.integers(0, 5))
        npert = int(rng.integers(0, m + 2))
        randord = bool(rng.integers(2))
        kind = "gauss" if rng.integers(2) else "sift"
        off = int(rng.choice([0, 0, 12345, 2 ** 33 + 7]))
        mode = ("forced", "natural", "default")[t % 3]
        if mode == "natural":                           # every block above the light threshold (n / 256 > 256): 66 000 or 140 000, d small
            n, d, J = int(rng.choice([66_000, 140_000])), min(d, 16), max(J, 1)
            if m > 8:
                ils, J = ils[:1], min(J, 2)             # keep the oracle side in seconds
        out.append((t, d, n, m, ils, J, npert, randord, kind, off, mode))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "c%d_d%d_n%d_m%d_%s" % (c[:4] + (c[10],)))
def test_random_shape_matches_oracle(lsq, oracle, case):
    t, d, n, m, ils, J, npert, randord, kind, off, mode = case
    X, K, B0 = make_problem(d, n, m, seed=100 + t, kind=kind)
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, randord, 7 * t + 1, global_offset=off)
    chunk = max(1, n // 3 + 1) if (mode != "natural" and t % 2 == 0) else None      # several resident chunks in half of the other cases
    with lsq.Engine(0, chunk=chunk, schedule=(4 if t % 4 == 3 else 6)) as eng:       # 6 = the default (16-bit filtered walk), 4 = the f32 walk
        if mode == "forced":
            eng.set_option("light", 0)
            eng.set_option("q16_min", 0)
        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, randord, seed=7 * t + 1, global_offset=off)
        tm = eng.timings()
    assert np.array_equal(Bs, Bs_ref), "%d of %d codes differ" % ((Bs != Bs_ref).sum(), Bs.size)
    assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
    staged, light = tm["staged_blocks"] + tm["filtered_blocks"], tm["light_blocks"]
    if J > 0 and mode != "default":
        # schedule 4 stages f32 slices; schedule 6 runs the filtered walk -- and hands a chunk over to the f32 walk when its first ILS iteration
        # came out mostly ambiguous (tiny d: the probe of option filter_probe_div), which then shows up in filter_fallback_chunks
        assert (tm["staged_blocks"] > 0) == (t % 4 == 3 or tm["filter_fallback_chunks"] > 0) and (tm["filtered_blocks"] > 0) == (t % 4 != 3), tm
    if J == 0:
        assert staged == 0 and light == 0                  # no sweeps: perturbation + accept only
    elif mode == "forced":
        assert light == 0 and staged > 0, (staged, light)
    elif mode == "natural":
        assert staged > 0, (staged, light)
