"""The initialisers' CHECKER (oracle/init_oracle.py: PQ / OPQ / ChainQ restated in numpy) and the product's host-side glue.  The reference has
no tests for these trainers and delegates to unpinned packages (PARITY UNPINNED), so these are property tests of the algorithms: k-means / OPQ / ChainQ
objectives do not increase, Viterbi is the exact chain optimum (brute force), rotations stay orthogonal, the chain's dimension structure is the
reference's (src/codebook_update.jl:88-102).  The product runs its assignment and Viterbi steps on the device (csrc/lsq_init.hip); tests/test_gpu_initializers.py
holds those kernels to this checker bit for bit."""
import itertools

import numpy as np
import pytest

import oracle.init_oracle as ini


def clustered(d, n, k=12, seed=0, spread=0.15):
    rng = np.random.default_rng(seed)
    cen = rng.standard_normal((d, k)).astype(np.float32) * 2.0
    a = rng.integers(k, size=n)
    return (cen[:, a] + spread * rng.standard_normal((d, n))).astype(np.float32)


def test_kmeans_improves_on_its_seeding(lsq):
    X = clustered(6, 800, k=10, seed=1)
    C0, a0, cost0 = ini.kmeans(X, 10, niter=0, seed=3)
    C, a, cost = ini.kmeans(X, 10, niter=25, seed=3)
    assert cost <= cost0 + 1e-3
    assert a.min() >= 0 and a.max() < 10 and C.shape == (6, 10)
    # assignments are nearest centers
    d2 = ((X[:, None, :] - C[:, :, None]) ** 2).sum(0)
    assert np.allclose(d2[a, np.arange(X.shape[1])], d2.min(0), rtol=1e-4, atol=1e-4)


def test_train_pq_and_quantize(lsq):
    X = clustered(8, 1200, seed=2)
    C, B, err = ini.train_pq(X, 4, 16, seed=0)
    assert len(C) == 4 and C[0].shape == (2, 16) and B.shape == (4, 1200) and B.dtype == np.int16
    assert B.min() >= 1 and B.max() <= 16
    B2 = ini.quantize_pq(X, C)
    assert np.array_equal(B, B2)                       # converged k-means assignments are nearest-codeword assignments
    assert err < float((X ** 2).sum() / X.shape[1])    # better than the zero codebook


def test_train_opq_monotone_and_orthogonal(lsq):
    X = clustered(12, 1500, seed=3)
    X = (np.linalg.qr(np.random.default_rng(0).standard_normal((12, 12)))[0].astype(np.float32) @ X)   # hide the axis structure
    C, B, R, obj = ini.train_opq(X, 4, 16, 6, "natural", seed=1)
    assert obj.shape == (7,) and B.shape == (4, 1500) and B.min() >= 1 and B.max() <= 16
    assert np.allclose(R.T @ R, np.eye(12), atol=1e-4)
    assert obj[-1] <= obj[0] and np.all(np.diff(obj) <= 1e-3 * obj[0])      # alternating minimisation
    assert np.array_equal(ini.quantize_opq(X, R, C), ini.quantize_pq(R.T @ X, C))
    with pytest.raises(ValueError):
        ini.train_opq(X, 4, 16, 1, "nope")


def test_chain_dimension_structure(lsq):
    od = lsq.get_cbdims_chain(8, 3)                   # blocks [0,4) [4,8): first, both, last
    assert [(s.start, s.stop) for s in od] == [(0, 4), (0, 8), (4, 8)]
    od = lsq.get_cbdims_chain(128, 7)
    assert od[0].start == 0 and od[-1].stop == 128 and len(od) == 7
    sub = lsq.splitarray(np.arange(128), 6)            # the m-1 blocks (utils.jl:152-177): 22, 22, 21, 21, 21, 21 dims
    starts = np.cumsum([0] + [len(s) for s in sub])
    for i in range(1, 6):                              # interior codebooks cover two consecutive blocks
        assert (od[i].start, od[i].stop) == (starts[i - 1], starts[i + 1])


def test_viterbi_is_the_exact_chain_optimum(lsq):
    rng = np.random.default_rng(5)
    d, n, m, h = 6, 40, 4, 5
    X = rng.standard_normal((d, n)).astype(np.float32)
    C = [rng.standard_normal((d, h)).astype(np.float32) for _ in range(m)]
    B = ini.encoding_viterbi(X, C, block=7)
    assert B.shape == (m, n) and B.min() >= 1 and B.max() <= h

    def energy(x, code):
        e = sum(float(-2.0 * C[i][:, code[i]] @ x + C[i][:, code[i]] @ C[i][:, code[i]]) for i in range(m))
        return e + sum(float(2.0 * C[i][:, code[i]] @ C[i + 1][:, code[i + 1]]) for i in range(m - 1))

    for i in range(n):
        best = min(energy(X[:, i], c) for c in itertools.product(range(h), repeat=m))
        got = energy(X[:, i], tuple(int(b) - 1 for b in B[:, i]))
        assert got <= best + 1e-3 * max(1.0, abs(best))


def test_train_chainq_decreases_error(lsq):
    X = clustered(12, 900, seed=7)
    m, h = 4, 8
    C0, B0, R0, _ = ini.train_opq(X, m, h, 2, "natural", seed=2)
    C, B, R, obj = ini.train_chainq(X, m, h, R0, B0, C0, 3)
    assert len(C) == m and C[0].shape == (12, h) and B.shape == (m, 900) and B.min() >= 1 and B.max() <= h
    assert np.allclose(R.T @ R, np.eye(12), atol=1e-4)
    assert obj[-1] <= obj[0] * 1.001
    od = lsq.get_cbdims_chain(12, m)
    for i in range(m):                                 # codebooks are zero outside the dimensions they cover
        mask = np.ones(12, dtype=bool)
        mask[od[i]] = False
        assert np.all(C[i][mask] == 0)


# ---- sanity oracles named by SURVEY 8(f)-4: sklearn k-means, numpy SVD Procrustes -------------------------------------------------
def test_kmeans_objective_close_to_sklearn(lsq):
    """The reference delegates to Clustering.jl's kmeans (un-pinned).  Sanity oracle: scikit-learn's Lloyd iterations on the same data
    from the same number of centers -- the objectives of two correct k-means runs on well-clustered data agree within a few percent
    (different seedings), and ours started FROM sklearn's centers must not get worse (Lloyd steps never increase the objective)."""
    from sklearn.cluster import KMeans
    X = clustered(8, 3000, k=16, seed=11, spread=0.2)
    C, a, cost = ini.kmeans(X, 16, niter=30, seed=5)
    km = KMeans(n_clusters=16, n_init=4, max_iter=100, random_state=0, algorithm="lloyd").fit(X.T.astype(np.float64))
    ref = float(km.inertia_)
    assert cost <= 1.25 * ref and ref <= 1.25 * cost, (cost, ref)
    # one assignment step of ours on sklearn's converged centers reproduces sklearn's labels (nearest-center rule) and its inertia
    lab, costs = ini._assign(km.cluster_centers_.T.astype(np.float32), X)
    assert (lab == km.labels_).mean() > 0.999
    assert abs(float(costs.sum()) - ref) <= 1e-3 * ref


def test_procrustes_rotation_matches_numpy_svd(lsq):
    """OPQ's rotation update (src/opq/OPQ.jl: SVD of X * CB') vs the textbook orthogonal-Procrustes solution from numpy's SVD:
    R = U V' maximises trace(R' X CB'), is orthogonal, and no random orthogonal matrix does better."""
    rng = np.random.default_rng(2)
    d, n = 10, 500
    X = rng.standard_normal((d, n)).astype(np.float32)
    Q = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)
    CB = (Q.T @ X + 0.05 * rng.standard_normal((d, n))).astype(np.float32)       # CB ~ R' X for a hidden rotation
    R = ini._procrustes(X, CB)
    U, _, Vt = np.linalg.svd(X.astype(np.float64) @ CB.astype(np.float64).T)
    Rref = U @ Vt
    assert np.allclose(R.T @ R, np.eye(d), atol=1e-4)
    assert np.allclose(R, Rref, atol=1e-3)
    err = np.linalg.norm(R.T @ X - CB)
    for _ in range(5):
        Rr = np.linalg.qr(rng.standard_normal((d, d)))[0]
        assert err <= np.linalg.norm(Rr.T @ X - CB) + 1e-6
    assert np.allclose(R, Q, atol=0.05)                                          # and it recovers the hidden rotation
