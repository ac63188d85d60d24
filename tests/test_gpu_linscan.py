"""Row 8(f)-1 on the device: lsq_linscan / lsq_linscan_dev (csrc/lsq_adc.hip) must return what the reference's own
linscan_aqd_query_extra_byte returns (oracle/_ref: src/linscan/cpp/linscan_aqd_pairwise_byte.cpp compiled as it stands) -- distances bit
for bit, 1-based ids, order including ties -- on every road of the selection: exhaustive (small databases), thresholded candidate lists
(large ones), and the fallback for queries whose list came out short or overflowed."""
import numpy as np
import pytest

from test_linscan import _case, _ours

pytestmark = pytest.mark.gpu
H = 256


def _reference(lsq, oracle, codes, Q, K, dbnorms, m, knn):
    """the reference build where it travelled (oracle/_ref), else this library's host scan (itself pinned to the reference in test_linscan.py)"""
    if oracle.ref_linscan_path() is not None:
        return oracle.ref_linscan(codes, Q, K, dbnorms, m, H, knn)
    return _ours(lsq, codes, Q, K, dbnorms, m, knn)


def _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect=None, **options):
    dref, iref = _reference(lsq, oracle, codes, Q, K, dbnorms, m, knn)
    with lsq.Engine(0) as eng:
        for k, v in options.items():
            eng.set_option(k, v)
        dists, ids = eng.linscan(codes, Q, K, dbnorms, m, knn)
        st = eng.linscan_stats()
    assert np.array_equal(ids, iref), "%d of %d ids differ (%r)" % ((ids != iref).sum(), ids.size, st)
    assert np.array_equal(dists.view(np.uint32), dref.view(np.uint32)), "max |diff| %g" % np.abs(dists - dref).max()
    assert ids.min() >= 1 and ids.max() <= codes.shape[0]
    if expect is not None:
        for k, v in expect.items():
            assert (st[k] == v) if not callable(v) else v(st[k]), (k, st)
    return st


@pytest.mark.parametrize("n,nq,d,m,knn,ties", [(5000, 37, 32, 7, 100, False), (3000, 16, 128, 8, 1000, False), (2000, 20, 16, 4, 50, True),
                                              (64, 5, 8, 2, 64, True), (1000, 3, 24, 16, 10, False), (17, 1, 5, 1, 1, False),
                                              (4096, 33, 960, 8, 7, False), (300, 2, 1030, 3, 300, False)])
def test_small_databases_take_the_exhaustive_road(lsq, oracle, n, nq, d, m, knn, ties):
    rng = np.random.default_rng(n + d)
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties)
    _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 1, "fallback_queries": 0, "queries": nq})


@pytest.mark.parametrize("n,nq,d,m,knn,ties", [(200_000, 40, 32, 8, 100, False), (150_001, 19, 16, 16, 10, False), (120_000, 16, 24, 7, 1000, False),
                                              (100_000, 50, 8, 4, 1, False), (131_072, 33, 20, 12, 64, True), (90_000, 5, 12, 5, 2000, True)])
def test_thresholded_candidate_lists(lsq, oracle, n, nq, d, m, knn, ties):
    """the production road: a sampled threshold per query, (dist, id) pairs at or below it appended by the scan, lists sorted"""
    rng = np.random.default_rng(n + m)
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties)
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 0, "queries": nq})
    assert st["fallback_queries"] == 0, st                                  # exchangeable data: the estimate holds for every query
    assert st["candidates"] < 0.25 * n * nq and st["candidates"] >= knn * nq, st      # a small part of the distances was ever written


def test_reference_default_knn_and_query_batches(lsq, oracle):
    """Linscan.jl's default k = 10000 on a 300 000-code database (thresholded road with long lists), and more queries than one batch of the
    exhaustive road holds (2^28 / n per batch)"""
    rng = np.random.default_rng(21)
    codes, Q, K, dbnorms = _case(rng, 300_000, 6, 16, 8)
    st = _check(lsq, oracle, codes, Q, K, dbnorms, 8, 10_000, expect={"exhaustive": 0, "fallback_queries": 0})
    assert st["list_capacity"] < 100_000, st
    codes, Q, K, dbnorms = _case(rng, 60_000, 4500, 8, 4)
    _check(lsq, oracle, codes, Q, K, dbnorms, 4, 5, expect={"exhaustive": 1, "batches": 2})


def test_short_lists_fall_back(lsq, oracle):
    """threshold = the sample's MINIMUM (option linscan_rank = 1): every list is far too short -> every query is redone exhaustively"""
    rng = np.random.default_rng(11)
    n, nq, d, m, knn = 100_000, 21, 16, 8, 50
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 0, "fallback_queries": nq}, linscan_rank=1)


def test_overflowing_lists_fall_back(lsq, oracle):
    """a database of identical entries: every distance ties with the threshold, the lists overflow, the fallback orders the ties by id"""
    rng = np.random.default_rng(12)
    n, nq, d, m, knn = 80_000, 9, 16, 8, 30
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    codes[:] = codes[0]
    dbnorms[:] = dbnorms[0]
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"fallback_queries": nq})
    with lsq.Engine(0) as eng:
        _, ids = eng.linscan(codes, Q, K, dbnorms, m, knn)
    assert np.array_equal(ids, np.tile(np.arange(1, knn + 1, dtype=np.int32), (nq, 1))), st


def test_sorted_database_is_still_exact(lsq, oracle):
    """the near neighbours of every query sit where the strided sample does not look (and the sampled entries are all far):
    the threshold is useless, the answer must not be"""
    rng = np.random.default_rng(13)
    n, nq, d, m, knn = 160_000, 12, 16, 8, 200
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    stride = n // 16384
    dbnorms[::stride] += np.float32(1.0e6)                                   # every sampled entry is pushed far away
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn)
    assert st["fallback_queries"] == nq, st                                 # the lists overflowed (threshold above everything unsampled)


def test_forced_exhaustive_equals_thresholded(lsq, oracle):
    rng = np.random.default_rng(14)
    n, nq, d, m, knn = 70_000, 18, 32, 8, 500
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties=True)
    _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 1}, linscan_exhaustive=1)
    _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 0})


def test_device_tensors_and_reference_shaped_call(lsq, oracle):
    import torch
    rng = np.random.default_rng(15)
    n, nq, d, m, knn = 100_000, 70, 32, 8, 100
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    dref, iref = _reference(lsq, oracle, codes, Q, K, dbnorms, m, knn)
    dev = torch.device("cuda:0")
    with lsq.Engine(0, profile=True) as eng:
        dd, di = eng.linscan_dev(torch.from_numpy(codes).to(dev), torch.from_numpy(Q).to(dev), torch.from_numpy(K).to(dev),
                                 torch.from_numpy(dbnorms).to(dev), m, knn)
        torch.cuda.synchronize()
        st = eng.linscan_stats()
        assert np.array_equal(di.cpu().numpy(), iref) and np.array_equal(dd.cpu().numpy(), dref)
        assert st["scan_ms"] > 0 and st["lut_ms"] > 0 and st["select_ms"] > 0, st
        # Julia shapes: B (m,n), X (d,nq), C list of (d,h); identity rotation
        C = [np.ascontiguousarray(K[j * H:(j + 1) * H].T) for j in range(m)]
        dists, res = lsq.linscan_lsq(codes.T, Q.T, C, dbnorms, np.eye(d, dtype=np.float32), knn, engine=eng)
    assert np.array_equal(res.T, iref) and np.array_equal(dists.T, dref)


def test_sift1m_shape_sample_of_queries(lsq, oracle):
    """BASELINE configs[1]'s search shape (10^6 codes, m = 8, d = 128, knn = 1000): 48 queries against the host scan"""
    rng = np.random.default_rng(16)
    n, nq, d, m, knn = 1_000_000, 48, 128, 8, 1000
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 0})
    assert st["fallback_queries"] == 0 and st["candidates"] < 0.02 * n * nq, st


def test_bad_arguments(lsq):
    z = np.zeros((4, 2), np.uint8)
    with lsq.Engine(0) as eng:
        with pytest.raises(lsq._lib.LsqError):
            eng.linscan(z, np.zeros((1, 3), np.float32), np.zeros((2 * H, 3), np.float32), np.zeros(4, np.float32), 2, 5)      # nn > n
        with pytest.raises(lsq._lib.LsqError):
            eng.linscan(z, np.zeros((1, 3), np.float32), np.zeros((2 * 16, 3), np.float32), np.zeros(4, np.float32), 2, 1, h=16)   # h != 256
        d, i = eng.linscan(z, np.zeros((0, 3), np.float32), np.zeros((2 * H, 3), np.float32), np.zeros(4, np.float32), 2, 1)   # no queries
        assert d.shape == (0, 1) and i.shape == (0, 1)


def test_sharded_search_default_scanner_is_the_device_scan(lsq, oracle):
    """distributed.search_sharded with its product default (the device scan) in a one-rank world; the two-rank merge runs on CPU under gloo
    (tests/test_distributed_gloo.py)"""
    import torch
    rng = np.random.default_rng(31)
    n, nq, d, m, knn = 90_000, 11, 16, 8, 77
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties=True)
    dref, iref = _reference(lsq, oracle, codes, Q, K, dbnorms, m, knn)
    dev = torch.device("cuda:0")
    dd, ii = lsq.distributed.search_sharded(torch.from_numpy(codes).to(dev), torch.from_numpy(dbnorms).to(dev), torch.from_numpy(Q).to(dev),
                                            torch.from_numpy(K).to(dev), m, knn, n_total=n, shard_start=0)
    assert np.array_equal(ii.cpu().numpy(), iref) and np.array_equal(dd.cpu().numpy(), dref)


@pytest.mark.parametrize("case", range(16))
def test_random_shapes(lsq, oracle, case):
    """seeded random shapes over both roads: any m in 1..16 (dword and byte code reads), ragged query tiles, code ranges that do not divide, nn from 1 to n"""
    rng = np.random.default_rng(1000 + case)
    m = int(rng.integers(1, 17))
    d = int(rng.choice([1, 3, 8, 17, 32, 100, 128]))
    big = case % 2 == 1
    n = int(rng.integers(70_000, 260_000)) if big else int(rng.integers(1, 9_000))
    nq = int(rng.integers(1, 70))
    knn = int(min(n, rng.choice([1, 2, 10, 100, 777]) if big else rng.integers(1, n + 1)))
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties=bool(case % 3 == 0))
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"queries": nq})
    assert st["exhaustive"] == (0 if big else 1), (n, nq, d, m, knn, st)


def test_sift1m_shape_every_query(lsq):
    """BASELINE configs[1]'s search at full size -- 10^6 codes, 10^4 queries, 1000 neighbours each -- against this library's host scan (pinned to the
    reference build in tests/test_linscan.py) on EVERY query, twice (the candidate lists are filled by atomics in a different order every time;
    the answer must not notice)"""
    rng = np.random.default_rng(17)
    n, nq, d, m, knn = 1_000_000, 10_000, 128, 8, 1000
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    dref, iref = _ours(lsq, codes, Q, K, dbnorms, m, knn)
    with lsq.Engine(0) as eng:
        for _ in range(2):
            dists, ids = eng.linscan(codes, Q, K, dbnorms, m, knn)
            assert np.array_equal(ids, iref) and np.array_equal(dists.view(np.uint32), dref.view(np.uint32))
        st = eng.linscan_stats()
    assert st["fallback_queries"] == 0 and st["queries"] == 2 * nq, st


def test_more_queries_than_one_thresholded_batch(lsq, oracle):
    """20 000 queries on the thresholded road: two batches of candidate lists (16 384 queries each at most), ragged second batch"""
    rng = np.random.default_rng(41)
    n, nq, d, m, knn = 70_000, 20_000, 8, 4, 3
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m)
    st = _check(lsq, oracle, codes, Q, K, dbnorms, m, knn, expect={"exhaustive": 0, "batches": 2})
    assert st["fallback_queries"] <= 2, st          # 1e-8 per query by design


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_single_process_multi_device_scan(lsq, oracle, devices):
    """lsq_multi_linscan with the one GPU listed two / three times (contexts, host threads, shards and the host merge are those of a real multi-GPU
    node): duplicated entries tie ACROSS the shards and must come out ordered by global id; one shard is smaller than nn"""
    rng = np.random.default_rng(51)
    n, nq, d, m, knn = 90_001, 23, 16, 8, 300
    codes, Q, K, dbnorms = _case(rng, n, nq, d, m, ties=True)
    dref, iref = _reference(lsq, oracle, codes, Q, K, dbnorms, m, knn)
    with lsq.MultiEngine(devices) as mg:
        dists, ids = mg.linscan(codes, Q, K, dbnorms, m, knn)
        assert np.array_equal(ids, iref) and np.array_equal(dists.view(np.uint32), dref.view(np.uint32))
        C = [np.ascontiguousarray(K[j * H:(j + 1) * H].T) for j in range(m)]
        d2, r2 = lsq.linscan_lsq(codes.T, Q.T, C, dbnorms, np.eye(d, dtype=np.float32), knn, engine=mg)
        assert np.array_equal(r2.T, iref) and np.array_equal(d2.T, dref)
        # a database smaller than nn x devices: every shard returns all of its entries
        small = 250
        ds, is_ = mg.linscan(codes[:small], Q, K, dbnorms[:small], m, 200)
    dr, ir = _reference(lsq, oracle, codes[:small], Q, K, dbnorms[:small], m, 200)
    assert np.array_equal(is_, ir) and np.array_equal(ds, dr)
