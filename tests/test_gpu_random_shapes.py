"""Randomised parity sweep: shapes, iteration counts and options drawn from a fixed seed, every case compared with the
oracle on all vectors (codes bit-exact, objective 1e-5).  Complements the hand-picked cases of test_gpu_parity.py: odd
dimensions, every m in 1..16, n around the block / pass / light-block boundaries, multi-snapshot calls, offsets."""
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
        out.append((t, d, n, m, ils, J, npert, randord, kind, off))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "c%d_d%d_n%d_m%d" % c[:4])
def test_random_shape_matches_oracle(lsq, oracle, case):
    t, d, n, m, ils, J, npert, randord, kind, off = case
    X, K, B0 = make_problem(d, n, m, seed=100 + t, kind=kind)
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, randord, 7 * t + 1, global_offset=off)
    chunk = None if t % 3 else max(1, n // 3 + 1)          # every third case: several resident chunks
    with lsq.Engine(0, chunk=chunk) as eng:
        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, randord, seed=7 * t + 1, global_offset=off)
    assert np.array_equal(Bs, Bs_ref), "%d of %d codes differ" % ((Bs != Bs_ref).sum(), Bs.size)
    assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
