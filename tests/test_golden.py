"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU leg: the oracle still reproduces them (guards the checker against drift).
GPU leg: the HIP path reproduces them through the C-ABI WITHOUT consulting the oracle."""
import glob
import os

import numpy as np
import pytest

from conftest import ENCODE_VARIANTS, open_engine

H = 256
FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _load(path):
    z = np.load(path)
    d, n, m, h, J, npert, randord, seed = [int(v) for v in z["params"]]
    X = z["X8"].astype(np.float32)
    K = z["P8"].astype(np.float32) / np.float32(m)
    return z, X, K, d, n, m, J, npert, bool(randord), seed


def test_fixtures_present():
    assert len(FIXTURES) >= 6


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_golden(oracle, path):
    z, X, K, d, n, m, J, npert, randord, seed = _load(path)
    Bs, objs, stats = oracle.encode_icm(X, z["B0"], K, m, H, z["ilsiters"], J, npert, randord, seed, want_stats=True)
    assert np.array_equal(Bs, z["Bs"])
    assert np.array_equal(stats.astype(np.int64), z["stats"])
    assert np.allclose(objs, z["objs"], rtol=1e-6, atol=0)
    assert np.array_equal(oracle.unaries(X, K, m, H)[:, :4, :], z["U_rows"])
    assert np.array_equal(oracle.veccost(X, K, (z["B0"] - 1).astype(np.uint8), H), z["cost0"])
    if m > 1:
        assert np.array_equal(oracle.tables(K, m, H)[0, m - 1, :4, :], z["T_rows"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ENCODE_VARIANTS)
@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_reproduces_golden(lsq, path, variant):
    z, X, K, d, n, m, J, npert, randord, seed = _load(path)
    with open_engine(lsq, variant, profile=True) as eng:
        Bs, objs = eng.encode_icm(X, z["B0"], K, m, z["ilsiters"], J, npert, randord, seed=seed)
        t = eng.timings()
        if variant.get("q16_min") == 0 and variant.get("light") == 0:      # every code came out of the shipped 16-bit filtered walk
            assert t["filtered_blocks"] > 0 and t["staged_blocks"] == 0 and t["light_blocks"] == 0, t
        assert np.array_equal(Bs, z["Bs"]), "%d codes differ" % (Bs != z["Bs"]).sum()
        assert np.allclose(objs, z["objs"], rtol=1e-5, atol=0)            # north_star tolerance for the MSE
        assert np.array_equal(eng.get_unaries(X, K, m)[:, :4, :], z["U_rows"])
        assert np.array_equal(eng.veccost(X, z["B0"], K, m), z["cost0"])
        if m > 1:
            assert np.array_equal(eng.get_binaries(K, m)[0, m - 1, :4, :], z["T_rows"])
