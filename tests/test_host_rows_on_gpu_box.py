"""SURVEY 8(f) rows f-1 .. f-3 are host code (north_star keeps them on the host), so their tests carry no `gpu` mark and the driver's
GPU test record said nothing about them.  This module re-runs their core checks UNDER the `gpu` mark, on the GPU
box's host cores and through the same C-ABI library the GPU tests load:

  f-1  lsq_linscan_aqd_query_extra_byte == the REAL reference build in oracle/_ref (src/linscan/cpp/linscan_aqd_pairwise_byte.cpp:14-93
       compiled from the reference's own source by oracle/Makefile; the prebuilt .so travels with the repo snapshot), bit for bit;
  f-2  quantize_norms / reconstruct (src/utils.jl:6-31, :203-223) and the *vecs readers (src/read/*.jl);
  f-3  lsq_update_codebooks == scipy's LSQR on the reference's sparse system (src/codebook_update.jl:52-86).
"""
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
