"""Row 8(f)-1: ADC linear scan.  This half of the path IS pinned by the real reference: oracle/_ref holds
src/linscan/cpp/linscan_aqd_pairwise_byte.cpp compiled from the reference's own source with its own flags
(oracle/Makefile `ref`).  lsq_linscan_aqd_query_extra_byte must reproduce it bit for bit -- distances AND ids,
including ties -- on the same inputs.  Host code: runs without a GPU."""
import numpy as np
import pytest

H = 256


def _case(rng, n, nq, d, m, ties=False):
    K = (rng.standard_normal((m * H, d)) * 0.5).astype(np.float32)
    codes = rng.integers(0, H, size=(n, m), dtype=np.uint8)
    if ties:
        codes[n // 2:] = codes[: n - n // 2]              # duplicated database entries -> exactly equal distances
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    recon = sum(K[j * H + codes[:, j].astype(np.int64)] for j in range(m))
    dbnorms = (recon.astype(np.float64) ** 2).sum(1).astype(np.float32)
    if ties:
        dbnorms[n // 2:] = dbnorms[: n - n // 2]
    return codes, Q, K, dbnorms


def _ours(lsq, codes, Q, K, dbnorms, m, knn, nthreads=0):
    L = lsq._lib.load()
    nq, d = Q.shape
    dists = np.zeros((nq, knn), np.float32)
    idx = np.zeros((nq, knn), np.int32)
    lsq._lib.check(L.lsq_linscan_aqd_query_extra_byte(dists.ctypes.data, idx.ctypes.data, codes.ctypes.data, Q.ctypes.data,
                                                      K.ctypes.data, dbnorms.ctypes.data, nq, codes.shape[0], m, H, d, knn, nthreads))
    return dists, idx


@pytest.mark.parametrize("n,nq,d,m,knn,ties", [(5000, 37, 32, 7, 100, False), (3000, 16, 128, 8, 1000, False),
                                              (2000, 20, 16, 4, 50, True), (64, 5, 8, 2, 64, True), (1000, 3, 24, 16, 10, False)])
def test_matches_reference_build(lsq, oracle, n, nq, d, m, knn, ties):
    if oracle.ref_linscan_path() is None:
        pytest.skip("oracle/_ref not built (reference sources absent and no prebuilt .so)")
    rng = np.random.default_rng(n + d)
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties)
    dref, iref = oracle.ref_linscan(codes, Q, K, dbnorms, m, H, knn)
    for nt in (1, 3):
        dists, idx = _ours(lsq, codes, Q, K, dbnorms, m, knn, nthreads=nt)
        assert np.array_equal(dists, dref), "max |diff| %g" % np.abs(dists - dref).max()
        assert np.array_equal(idx, iref)
    assert idx.min() >= 1 and idx.max() <= n                       # 1-based ids


def test_against_float64_brute_force(lsq):
    """Independent sanity check: ids are the true ADC neighbours (ranking by -2<q, sum c> + ||sum c||^2)."""
    rng = np.random.default_rng(5)
    n, nq, d, m, knn = 4000, 25, 64, 8, 10
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    dists, idx = _ours(lsq, codes, Q, K, dbnorms, m, knn)
    recon = sum(K[j * H + codes[:, j].astype(np.int64)] for j in range(m)).astype(np.float64)
    full = -2.0 * Q.astype(np.float64) @ recon.T + dbnorms.astype(np.float64)[None, :]
    assert np.all(np.diff(dists, axis=1) >= 0)
    for q in range(nq):
        truth = np.argsort(full[q], kind="stable")[:knn]
        assert len(set(truth + 1) & set(idx[q])) >= knn - 1        # f32 vs f64 may swap a near-tie at the cut
        assert np.allclose(dists[q], np.sort(full[q])[:knn], rtol=1e-4, atol=1e-3)


def test_reference_shaped_linscan_and_recall(lsq, oracle):
    rng = np.random.default_rng(9)
    n, nq, d, m, knn = 3000, 40, 32, 8, 100
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    C = [np.ascontiguousarray(K[j * H:(j + 1) * H].T) for j in range(m)]
    R = np.eye(d, dtype=np.float32)
    dists, res = lsq.linscan_lsq(codes.T, Q.T, C, dbnorms, R, knn)             # Julia shapes: B (m,n), X (d,nq)
    assert dists.shape == (knn, nq) and res.shape == (knn, nq) and res.dtype == np.int32
    if oracle.ref_linscan_path() is not None:
        dref, iref = oracle.ref_linscan(codes, Q, K, dbnorms, m, H, knn)
        assert np.array_equal(res.T, iref) and np.array_equal(dists.T, dref)
    # eval_recall: ground truth = our own first neighbour -> recall@1 = 1; a shifted truth -> recall@1 = 0
    rec = lsq.eval_recall(res[0], res, knn)
    assert rec.shape == (knn,) and rec[0] == 1.0 and np.all(np.diff(rec) >= 0)
    rec2 = lsq.eval_recall(res[4], res, knn)
    assert rec2[0] == 0.0 and rec2[3] == 0.0 and rec2[4] == 1.0
    assert lsq.eval_recall(np.full(nq, n + 5), res, knn)[-1] == 0.0


def test_bad_arguments(lsq):
    L = lsq._lib.load()
    z = np.zeros(8, np.float32)
    assert L.lsq_linscan_aqd_query_extra_byte(z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data,
                                              z.ctypes.data, 1, 4, 2, 256, 2, 5, 1) == lsq._lib.LSQ_EINVAL    # nn > n
    assert L.lsq_linscan_aqd_query_extra_byte(None, None, None, None, None, None, 0, 4, 2, 256, 2, 1, 1) == 0    # no queries


# ---- row 8(f)-2: norm quantisation and the TEXMEX readers (host glue) ---------------------------
def test_quantize_norms_and_reconstruct(lsq):
    rng = np.random.default_rng(3)
    d, n, m = 16, 600, 4
    C = [rng.standard_normal((d, H)).astype(np.float32) for _ in range(m)]
    B = rng.integers(1, H + 1, size=(m, n)).astype(np.int16)
    CB = lsq.reconstruct(B, C)
    ref = sum(C[i][:, B[i].astype(np.int64) - 1].astype(np.float64) for i in range(m))
    assert CB.shape == (d, n) and np.allclose(CB, ref, atol=1e-5)
    norms = (ref ** 2).sum(0)
    cbnorms = np.sort(rng.choice(norms, size=H, replace=False)).astype(np.float32)
    q = lsq.quantize_norms(B, C, cbnorms)
    assert q.dtype == np.int16 and q.min() >= 1 and q.max() <= H
    brute = np.argmin((norms[None, :] - cbnorms.astype(np.float64)[:, None]) ** 2, axis=0) + 1
    assert (q == brute).mean() > 0.98                          # f32 vs f64 may flip exact mid-points
    cb2 = np.concatenate([cbnorms[:1], cbnorms[:1], cbnorms[2:]])       # duplicated centroid: first index wins
    q2 = lsq.quantize_norms(B, C, cb2)
    assert not np.any(q2 == 2)
    # the host mirror against the oracle's independent restatement of src/utils.jl:6-31, 203-223 (the checker lives under oracle/, not in the product)
    import oracle as O
    assert np.array_equal(O.reconstruct(B, C).view(np.uint32), CB.view(np.uint32))
    oq, onorms = O.quantize_norms(B, C, cbnorms, want_norms=True)
    assert np.array_equal(oq, q) and np.array_equal(O.quantize_norms(B, C, cb2), q2)
    assert np.allclose(onorms, norms, rtol=1e-5)


def test_vecs_readers_roundtrip(lsq, tmp_path):
    rng = np.random.default_rng(4)
    d, n = 12, 37
    X = rng.standard_normal((n, d)).astype(np.float32)
    I = rng.integers(0, 1000, size=(n, d)).astype(np.int32)
    Bv = rng.integers(0, 256, size=(n, d)).astype(np.uint8)
    for name, arr in (("x.fvecs", X), ("x.ivecs", I), ("x.bvecs", Bv)):
        with open(tmp_path / name, "wb") as f:
            for row in arr:
                f.write(np.int32(d).tobytes())
                f.write(row.tobytes())
    assert np.array_equal(lsq.fvecs_read(None, str(tmp_path / "x.fvecs")), X.T)
    assert np.array_equal(lsq.fvecs_read(10, str(tmp_path / "x.fvecs")), X[:10].T)
    assert np.array_equal(lsq.fvecs_read((5, 20), str(tmp_path / "x.fvecs")), X[4:20].T)      # 1-based inclusive
    assert np.array_equal(lsq.ivecs_read(range(3, 8), str(tmp_path / "x.ivecs")), I[2:7].T)
    assert np.array_equal(lsq.bvecs_read(None, str(tmp_path / "x.bvecs")), Bv.T)
