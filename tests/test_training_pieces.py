"""Row 8(f)-3/4: the host-side codebook update (north_star keeps it on the host, behind the same C-ABI) and the
train_lsq loop that alternates it with the GPU encoder.  IterativeSolvers.lsqr is un-vendored and un-pinned in
the reference (parity unpinned): the C++ LSQR is checked against scipy's implementation of the same published
algorithm, against the normal-equations optimum, and through the monotone training objective."""
import numpy as np
import pytest

H = 256


def _problem(rng, d, n, m, noise=0.01):
    B = rng.integers(1, H + 1, size=(m, n)).astype(np.int16)
    Ctrue = [rng.standard_normal((d, H)).astype(np.float32) for _ in range(m)]
    X = sum(Ctrue[j][:, B[j] - 1] for j in range(m)) + noise * rng.standard_normal((d, n)).astype(np.float32)
    return X.astype(np.float32), B


def test_update_codebooks_matches_scipy_lsqr(lsq):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(0)
    d, n, m = 12, 4000, 4
    X, B = _problem(rng, d, n, m)
    C = lsq.update_codebooks(X, B, H, nthreads=3)
    assert len(C) == m and C[0].shape == (d, H) and C[0].dtype == np.float32
    K = np.concatenate(C, axis=1)                                           # d x (m*h) = hcat(C...)
    rows = np.tile(np.arange(n), m)
    cols = np.concatenate([(B[j] - 1) + j * H for j in range(m)])
    S = sp.csr_matrix((np.ones(n * m), (rows, cols)), shape=(n, m * H))     # sparsify_codes (utils.jl:50-69)
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    Kref = np.stack([spl.lsqr(S, X[t].astype(np.float64), atol=tol, btol=tol)[0] for t in range(d)])
    assert np.linalg.norm(K - Kref) <= 1e-4 * np.linalg.norm(Kref)
    # and it is (nearly) the least-squares optimum: residual orthogonal to the column space
    rec = sum(C[j][:, B[j] - 1] for j in range(m))
    resid = (X - rec).astype(np.float64)
    assert np.abs(S.T @ resid.T).max() <= 2e-2 * np.abs(X).max() * np.sqrt(n / H)
    assert np.linalg.norm(resid) <= 1.05 * np.linalg.norm(X - sum(c for c in [S @ Kref.T]).T)


def test_update_codebooks_training_scale_matches_scipy(lsq):
    """n = 120 000 (the reference trains on 1e4..1e5 vectors, README 64-66 / 171-177): the Float32 LSQR with double-accumulated norms and
    products must still agree with scipy's double-precision LSQR of the same system (sequential Float32 sums lose
    sqrt(n) eps .. n eps, the size of the sqrt(eps) stopping tolerance, at this scale)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(3)
    d, n, m = 4, 120_000, 4
    X, B = _problem(rng, d, n, m, noise=0.05)
    C = lsq.update_codebooks(X, B, H, nthreads=4)
    K = np.concatenate(C, axis=1)
    rows = np.tile(np.arange(n), m)
    cols = np.concatenate([(B[j] - 1) + j * H for j in range(m)])
    S = sp.csr_matrix((np.ones(n * m), (rows, cols)), shape=(n, m * H))
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    Kref = np.stack([spl.lsqr(S, X[t].astype(np.float64), atol=tol, btol=tol)[0] for t in range(d)])
    # S has a (m - 1)-dimensional null space (constant shifts between codebooks): compare what is determined -- the reconstruction
    rec = (S @ K.T).T
    rec_ref = (S @ Kref.T).T
    assert np.linalg.norm(rec - rec_ref) <= 2e-4 * np.linalg.norm(rec_ref)
    r0 = np.linalg.norm(X - rec_ref)
    assert np.linalg.norm(X - rec) <= r0 * (1 + 1e-4)


def test_update_codebooks_errors(lsq):
    X = np.zeros((4, 10), np.float32)
    B = np.ones((2, 10), np.int16)
    with pytest.raises(lsq._lib.LsqError):
        lsq.update_codebooks(X, B * 300, H)
    with pytest.raises(ValueError):
        lsq.update_codebooks(X, B, H, False, "cholesky")


@pytest.mark.gpu
def test_train_lsq_objective_decreases(lsq):
    """LSQ.jl:10-88 end to end on synthetic data: codebook update (host) <-> ILS/ICM encoding (GPU)."""
    import oracle as O
    d, n, m = 32, 3000, 4
    X = np.ascontiguousarray(O.synth_data_u8(5, n, d).T)                     # (d, n)
    rng = np.random.default_rng(1)
    B0 = rng.integers(1, H + 1, size=(m, n)).astype(np.int16)
    C0 = [np.zeros((d, H), np.float32) for _ in range(m)]
    R = np.eye(d, dtype=np.float32)
    C, B, cbnorms, B_norms, obj = lsq.train_lsq(X, m, H, R, B0, C0, 4, 2, 2, True, 2, False, seed=3)
    assert obj.shape == (4,) and np.all(np.diff(obj) <= 1e-3 * obj[:-1])    # alternating minimisation: non-increasing
    assert obj[-1] < 0.8 * obj[0]
    assert B.shape == (m, n) and B.min() >= 1 and B.max() <= H
    assert cbnorms.shape == (H,) and B_norms.shape == (1, n) and B_norms.min() >= 1
    q = lsq.quantize_norms(B, C, cbnorms)
    assert (q == B_norms.ravel()).mean() > 0.95
    assert abs(lsq.qerror(X, B, C) - obj[-1]) <= obj[-1]                      # same scale; final encode only improves it


def test_scalar_kmeans_assignment_equals_the_brute_force_scan(lsq):
    """initializers._assign_scalar (the norm codebook's k-means: n scalars) returns what the h x n scan of f32 (x - c)^2 with first-minimum returns --
    duplicated centres, exact midpoints, rounding collisions at the size of squared norms, centres in descending order."""
    import importlib
    ini = importlib.import_module(lsq.__name__ + ".initializers")
    rng = np.random.default_rng(0)

    def brute(c, x):
        dm = (x[None, :].astype(np.float32) - c[:, None].astype(np.float32)) ** 2
        a = dm.argmin(axis=0)
        return a, dm[a, np.arange(x.shape[0])]

    for t in range(120):
        h, n, kind = int(rng.integers(1, 40)), int(rng.integers(1, 400)), t % 4
        if kind == 0:
            c, x = rng.standard_normal(h).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        elif kind == 1:
            c, x = rng.integers(0, 6, h).astype(np.float32), (rng.integers(0, 12, n) / 2).astype(np.float32)
        elif kind == 2:
            c, x = (4e5 + rng.integers(0, 8, h) * 0.03125).astype(np.float32), (4e5 + rng.integers(0, 64, n) * 0.0078125).astype(np.float32)
        else:
            c = np.sort(rng.standard_normal(h)).astype(np.float32)[::-1].copy()
            x = np.concatenate([c[:min(h, n)], rng.standard_normal(max(n - h, 0)).astype(np.float32)])[:n]
        a0, d0 = brute(c, x)
        a1, d1 = ini._assign_scalar(c, x)
        assert np.array_equal(a0, a1) and np.array_equal(d0, d1), (t, kind)
    x = (rng.standard_normal(20_000) * 1e5 + 4e5).astype(np.float32)
    C, a, cost = ini.kmeans(x.reshape(1, -1), 64, niter=30, seed=1)
    assert C.shape == (1, 64) and a.shape == (20_000,) and np.isfinite(cost)
    assert cost / x.size < 0.002 * float(x.var())                         # 64 centres on a Gaussian: the distortion is far below the variance


def test_update_codebooks_lsmr_matches_scipy_lsmr(lsq):
    """codebook_upd_method = "lsmr" (src/codebook_update.jl:18-21): the host LSMR against scipy's double-precision LSMR of the same system with the
    same tolerances; the unknown-method error of codebook_update.jl:22-23; and LSMR and LSQR agree on what the system determines (the reconstruction)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(11)
    d, n, m = 10, 6000, 4
    X, B = _problem(rng, d, n, m, noise=0.05)
    C = lsq.update_codebooks(X, B, H, nthreads=3, codebook_upd_method="lsmr")
    assert len(C) == m and C[0].shape == (d, H) and C[0].dtype == np.float32
    K = np.concatenate(C, axis=1)
    rows = np.tile(np.arange(n), m)
    cols = np.concatenate([(B[j] - 1) + j * H for j in range(m)])
    S = sp.csr_matrix((np.ones(n * m), (rows, cols)), shape=(n, m * H))
    tol = float(np.sqrt(np.finfo(np.float32).eps))
    Kref = np.stack([spl.lsmr(S, X[t].astype(np.float64), atol=tol, btol=tol, conlim=1e8, maxiter=max(n, m * H))[0] for t in range(d)])
    rec, rec_ref = (S @ K.T).T, (S @ Kref.T).T
    assert np.linalg.norm(rec - rec_ref) <= 1e-3 * np.linalg.norm(rec_ref)
    assert np.linalg.norm(K - Kref) <= 2e-3 * np.linalg.norm(Kref)
    Kq = np.concatenate(lsq.update_codebooks(X, B, H, nthreads=3), axis=1)
    assert np.linalg.norm((S @ Kq.T).T - rec) <= 2e-3 * np.linalg.norm(rec)
    with pytest.raises(ValueError, match="unknown"):
        lsq.update_codebooks(X, B, H, codebook_upd_method="cholesky")
