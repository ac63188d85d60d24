"""The reference's demo flow (demos/demo_lsq_gpu.jl:22-76) end to end on synthetic clustered data:
OPQ init -> ChainQ init -> train_lsq (host LSQR + GPU ILS/ICM encoder) -> encode the base set on the GPU ->
quantise the database norms -> ADC linear scan -> recall.  No dataset ships with the reference, so the
checks are relative: every stage must not be worse than the one before it, and the search must find the true
nearest neighbours far more often than chance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = 256


def clustered(d, n, k, seed, spread=0.35):
    rng = np.random.default_rng(seed)
    cen = rng.standard_normal((d, k)).astype(np.float32) * 3.0
    a = rng.integers(k, size=n)
    return (cen[:, a] + spread * rng.standard_normal((d, n))).astype(np.float32)


def test_demo_flow_on_synthetic_data(lsq):
    d, m, ntrain, nbase, nq, knn = 32, 4, 3000, 6000, 64, 50
    allx = clustered(d, ntrain + nbase + nq, k=400, seed=11)
    x_train, x_base, x_query = allx[:, :ntrain], allx[:, ntrain:ntrain + nbase], allx[:, ntrain + nbase:]

    # === OPQ initialization === (demo_lsq_gpu.jl:22-26)
    C, B, R, opq_err = lsq.train_opq(x_train, m, H, 3, "natural", seed=1)
    # === ChainQ initialization === (:28-31)
    C, B, R, chain_err = lsq.train_chainq(x_train, m, H, R, B, C, 2)
    assert chain_err[-1] <= opq_err[-1] * 1.02
    # === LSQ train === (:33-40)
    ilsiter, icmiter, randord, npert = 4, 4, True, 2
    B_in, C_in = B.copy(), [c.copy() for c in C]
    C, B, cbnorms, B_norms, obj = lsq.train_lsq(x_train, m, H, R, B, C, 3, ilsiter, icmiter, randord, npert, seed=5)
    assert obj[-1] <= obj[0] * 1.001 and obj[-1] <= chain_err[-1] * 1.02
    with lsq.Engine(0) as eng:                             # the same training with the codebook update on the device too (lsq_update_codebooks_gpu)
        C2, B2, _, _, obj2 = lsq.train_lsq(x_train, m, H, R, B_in, C_in, 3, ilsiter, icmiter, randord, npert, seed=5, engine=eng, device_update=True)
    assert np.allclose(obj2, obj, rtol=1e-4) and obj2[-1] <= chain_err[-1] * 1.02       # the two LSQR solvers agree to ~1e-6; the trajectories stay together
    assert len(C) == m and C[0].shape == (d, H) and cbnorms.shape[0] <= H

    # === Encode the base set === (:42-55)
    B_base = lsq.randinit(nbase, m, H, seed=3)
    Bs, objs = lsq.encode_icm_cuda(x_base, B_base, C, [2, 8], icmiter, npert, randord, 2, False, seed=9)
    assert objs[1] <= objs[0]                              # more ILS iterations never hurt (accept rule is monotone)
    B_base = Bs[-1]
    base_err = lsq.qerror(x_base, B_base, C)
    assert np.isclose(base_err, objs[-1], rtol=1e-4)
    assert base_err < 0.25 * float((x_base ** 2).sum() / nbase)

    # === norms, search, recall === (:57-76)
    dbnormsB = lsq.quantize_norms(B_base, C, cbnorms)
    with lsq.Engine(0) as eng:                                              # the same on the device
        assert np.array_equal(lsq.quantize_norms(B_base, C, cbnorms, engine=eng), dbnormsB)
    db_norms = np.asarray(cbnorms, dtype=np.float32)[dbnormsB.astype(np.int64) - 1]
    dists, idx = lsq.linscan_lsq((B_base - 1).astype(np.uint8), x_query, C, db_norms, np.eye(d, dtype=np.float32), knn)
    assert idx.shape == (knn, nq) and idx.min() >= 1 and idx.max() <= nbase
    with lsq.Engine(0) as eng:                                              # the same search on the device: identical neighbours and distances
        dists_d, idx_d = lsq.linscan_lsq((B_base - 1).astype(np.uint8), x_query, C, db_norms, np.eye(d, dtype=np.float32), knn, engine=eng)
    assert np.array_equal(idx_d, idx) and np.array_equal(dists_d, dists)
    d2 = ((x_base[:, :, None] - x_query[:, None, :]) ** 2).sum(0)          # (nbase, nq) exact
    gt = d2.argmin(0) + 1
    rec = lsq.eval_recall(gt.astype(np.uint32), idx.astype(np.uint32), knn)
    assert rec[knn - 1] >= 0.9 and rec[0] >= 100.0 / nbase     # 32-bit codes on tight clusters: recall@1 is low, chance is 1/6000
    assert np.all(np.diff(rec) >= 0)


def test_train_lsq_dev_is_the_same_training_resident_in_hbm(lsq):
    """train_lsq_dev (LSQ.jl:10-88 on device tensors: no host copy between the steps) against train_lsq with the device codebook update on the same
    inputs: identical codebooks, codes, norm codebook; with a rotation (RX by torch instead of numpy: a different summation order) the trajectories
    stay together."""
    import torch
    d, m, n = 32, 4, 3000
    X = clustered(d, n, k=300, seed=21)
    B0 = lsq.randinit(n, m, H, seed=2)                              # (m, n) int16, 1-based
    niter, ilsiter, icmiter, randord, npert = 3, 2, 4, True, 2
    with lsq.Engine(0) as eng:
        C, B, cb, Bn, obj = lsq.train_lsq(X, m, H, np.eye(d, dtype=np.float32), B0, None, niter, ilsiter, icmiter, randord, npert, seed=5,
                                           engine=eng, device_update=True)
        dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        dB0 = torch.from_numpy(np.ascontiguousarray((B0.T - 1).astype(np.uint8))).cuda()
        dK, dB, cb2, Bn2, obj2 = lsq.train_lsq_dev(dX, m, H, dB0, niter, ilsiter, icmiter, randord, npert, seed=5, engine=eng)
        K = np.concatenate([np.asarray(Cj, dtype=np.float32).T for Cj in C], axis=0)
        assert np.array_equal(dK.cpu().numpy(), K)
        assert np.array_equal(dB.cpu().numpy().astype(np.int16) + 1, np.asarray(B).T)
        assert np.allclose(obj2, obj, rtol=1e-6) and obj2[-1] <= obj2[0]
        assert np.array_equal(cb2, cb) and np.array_equal(Bn2, np.asarray(Bn)) and Bn2.shape == (1, n)
        # with a rotation
        Q, _ = np.linalg.qr(np.random.default_rng(3).standard_normal((d, d)))
        R = Q.astype(np.float32)
        _, _, _, _, objr = lsq.train_lsq(X, m, H, R, B0, None, niter, ilsiter, icmiter, randord, npert, seed=5, engine=eng, device_update=True)
        _, _, _, _, objr2 = lsq.train_lsq_dev(dX, m, H, dB0, niter, ilsiter, icmiter, randord, npert, seed=5, engine=eng, R=R, norm_codebook=False)
        assert np.allclose(objr2, objr, rtol=2e-2) and objr2[-1] <= objr2[0] * 1.001


def test_train_lsq_dev_against_the_independent_oracle_trainer(lsq, oracle):
    """SURVEY 8(f)-4 under an INDEPENDENT check: oracle/train_oracle.py restates LSQ.jl:36-66's step order with the oracle's C encoder
    and scipy's LSQR (f64); train_lsq_dev (device LSQR in f32 + the HIP encoder, everything resident in HBM) must follow the same objective trajectory.
    The two differ only through the LSQR arithmetic (f32 vs f64, and WHERE each solver meets the reference's own stopping tolerance atol = btol =
    sqrt(eps(Float32)) = 3.5e-4: IterativeSolvers' default, codebook_update.jl:13-24): codebooks that are only determined to a few 1e-4 send a few
    near-tied node updates the other way, and the free-running trajectories drift apart at that scale -- measured 0 / 8e-4 / 5e-4 / 2e-3 relative over
    four iterations.  The first objective (before any data-dependent divergence) must agree to 1e-6, the others to 5e-3, both must decrease
    monotonically; the pieces themselves are pinned elsewhere (encoder: bit-exact vs the oracle; LSQR: 1e-3 vs scipy on identical inputs)."""
    import torch
    from oracle import train_oracle
    d, m, n = 32, 4, 6000
    X = clustered(d, n, k=300, seed=22)                               # (d, n)
    Xr = np.ascontiguousarray(X.T)
    B0 = lsq.randinit(n, m, H, seed=4)                                # (m, n) int16 1-based
    niter, ilsiter, icmiter, randord, npert = 4, 2, 4, True, 2
    Kref, Bref, objref = train_oracle.train_lsq(Xr, m, H, np.ascontiguousarray(B0.T), niter, ilsiter, icmiter, randord, npert, seed=7)
    with lsq.Engine(0) as eng:
        dX = torch.from_numpy(Xr).cuda()
        dB0 = torch.from_numpy(np.ascontiguousarray((B0.T - 1).astype(np.uint8))).cuda()
        dK, dB, _, _, obj = lsq.train_lsq_dev(dX, m, H, dB0, niter, ilsiter, icmiter, randord, npert, seed=7, engine=eng, norm_codebook=False)
        K, B = dK.cpu().numpy(), dB.cpu().numpy().astype(np.int16) + 1
    assert abs(obj[0] - objref[0]) <= 1e-6 * objref[0], (obj, objref)
    assert np.allclose(obj, objref, rtol=5e-3, atol=0), (obj, objref)
    assert np.all(np.diff(obj) < 0) and np.all(np.diff(objref) < 0)
    # (after four free-running iterations the two codebook sets are different local solutions of the same quality: their codes are not interchangeable --
    #  re-scoring one trainer's codes under the other's codebooks gives 11.5 against 3.3 -- so nothing is asserted about K or B beyond the objective)
