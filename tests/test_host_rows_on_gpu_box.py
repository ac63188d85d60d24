"""SURVEY 8(f) rows f-1 .. f-3 are host code (north_star keeps them on the host), so their tests carry no `gpu` mark and the driver's
GPU test record said nothing about them.  This module re-runs their core checks UNDER the `gpu` mark, on the GPU
box's host cores and through the same C-ABI library the GPU tests load:

  f-1  lsq_linscan_aqd_query_extra_byte == the REAL reference build in oracle/_ref (src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:14-93
       compiled from the reference's own source by oracle/Makefile; the prebuilt .so travels with the repo snapshot), bit for bit;
  f-2  quantize_norms / reconstruct (src/utils.jl:6-31, :203-223) and the *vecs readers (src/read/*.jl);
  f-3  lsq_update_codebooks == scipy's LSQR on the reference's sparse system (src/codebook_update.jl:52-86).
"""
import numpy as np
import pytest

import test_linscan as TL
import test_training_pieces as TP

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,nq,d,m,knn,ties", [(5000, 37, 32, 7, 100, False), (3000, 16, 128, 8, 1000, False), (2000, 20, 16, 4, 50, True)])
def test_f1_linscan_bit_identical_to_the_reference_build(lsq, oracle, n, nq, d, m, knn, ties):
    TL.test_matches_reference_build(lsq, oracle, n, nq, d, m, knn, ties)


def test_f1_linscan_ranks_like_float64_brute_force(lsq):
    TL.test_against_float64_brute_force(lsq)


def test_f1_reference_shaped_linscan_and_recall(lsq, oracle):
    TL.test_reference_shaped_linscan_and_recall(lsq, oracle)


def test_f2_quantize_norms_reconstruct_and_readers(lsq, tmp_path):
    TL.test_quantize_norms_and_reconstruct(lsq)
    TL.test_vecs_readers_roundtrip(lsq, tmp_path)


def test_f3_update_codebooks_matches_scipy_lsqr(lsq):
    TP.test_update_codebooks_matches_scipy_lsqr(lsq)
    TP.test_update_codebooks_training_scale_matches_scipy(lsq)


# ---- row 8(f)-2 on the device: lsq_quantize_norms / lsq_quantize_norms_dev against the Python mirror of src/utils.jl:6-31 ------------------
@pytest.mark.parametrize("d,n,m,ncb", [(16, 600, 4, 256), (128, 20_000, 8, 256), (30, 999, 7, 100), (960, 300, 16, 256), (5, 64, 1, 2)])
def test_quantize_norms_device_equals_the_mirror(lsq, d, n, m, ncb):
    import torch
    H = 256
    rng = np.random.default_rng(d + n)
    C = [rng.standard_normal((d, H)).astype(np.float32) for _ in range(m)]
    B = rng.integers(1, H + 1, size=(m, n)).astype(np.int16)
    CB = lsq.reconstruct(B, C)
    norms = np.zeros(n, dtype=np.float32)
    for t in range(d):                                              # the mirror's own order: dimensions ascending, square rounded before the add
        norms += CB[t] * CB[t]
    cb = np.sort(rng.choice(norms, size=ncb, replace=False)).astype(np.float32)
    if ncb > 4:
        cb[3] = cb[2]                                               # duplicated centroid: the first index wins (findmin)
    ref = lsq.quantize_norms(B, C, cb)
    K = np.ascontiguousarray(np.concatenate([c.T for c in C], axis=0))
    with lsq.Engine(0) as eng:
        idx, dbn, nrm = eng.quantize_norms(np.ascontiguousarray(B.T), K, cb, m)
        assert np.array_equal(nrm.view(np.uint32), norms.view(np.uint32)), "norms differ from the sequential f32 order"
        assert np.array_equal(idx, ref) and np.array_equal(dbn, cb[ref.astype(np.int64) - 1])
        assert np.array_equal(lsq.quantize_norms(B, C, cb, engine=eng), ref)
        dev = torch.device("cuda:0")
        di, dd, dn = eng.quantize_norms_dev(torch.from_numpy(np.ascontiguousarray((B.T - 1).astype(np.uint8))).to(dev), torch.from_numpy(K).to(dev), torch.from_numpy(cb).to(dev), m)
        torch.cuda.synchronize()
        assert np.array_equal(di.cpu().numpy().astype(np.int16) + 1, ref) and np.array_equal(dd.cpu().numpy(), dbn) and np.array_equal(dn.cpu().numpy(), nrm)
