import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# How the encode is run in the parity tests, all through the shipped library: its plain defaults (what a caller gets),
# schedule 6 forced onto every chunk with every block staged and both hand-overs to the f32 walk switched off (the 16-bit filtered walk
# is then the kernel that produces every code, whatever n and whatever the data), the f32 walk in both launch shapes.
def _variant(name, marks=(), **options):
    return pytest.param(options, id=name, marks=list(marks))


ENCODE_VARIANTS = [
    _variant("default"),
    _variant("s6_forced", schedule=6, q16_min=0, light=0, filter_probe_div=0, filter_fallback_div=0),
    _variant("s6_light", schedule=6, q16_min=0),
    _variant("s4", schedule=4),
    _variant("s3", schedule=3),
]


def open_engine(lsq, options, **kw):
    """Engine(0) with a variant's options applied (`tuning` selects the tuning build)."""
    opts = dict(options)
    eng = lsq.Engine(0, tuning=bool(opts.pop("tuning", 0)), **kw)
    for k, v in opts.items():
        eng.set_option(k, v)
    return eng


@pytest.fixture(scope="session")
def lsq():
    return importlib.import_module("local-search-quantization_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def engine(lsq):
    """One C-ABI context on cuda:0 -- fails loudly (no skip, no fallback) if the HIP extension or the
    GPU is missing, so that a silent CPU path can never make the gpu tests pass."""
    eng = lsq.Engine(0)
    yield eng
    eng.close()


def make_problem(d, n, m, h=256, seed=0, kind="sift"):
    """Seeded synthetic inputs: SIFT-like integer-valued data, codebooks = sampled data vectors / m
    (SURVEY 8(d)), random initial codes.  Row-major: X (n,d), K (m*h,d), B0 (n,m) int16 1-based."""
    import oracle as O
    rng = np.random.default_rng(seed)
    if kind == "sift":
        X = O.synth_data_u8(1000 + seed, n, d)
        pool = O.synth_data_u8(2000 + seed, m * h, d)
        K = np.ascontiguousarray(pool[rng.permutation(m * h)] / np.float32(m))
    elif kind == "gauss":
        X = rng.standard_normal((n, d)).astype(np.float32)
        K = (rng.standard_normal((m * h, d)) / m).astype(np.float32)
    else:
        raise ValueError(kind)
    B0 = O.randinit(3000 + seed, n, m, h)
    return X, K, B0
