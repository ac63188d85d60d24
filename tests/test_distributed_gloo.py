"""world_size = 2 on CPU (gloo): the sharded-encode logic of local-search-quantization_amd/distributed.py
-- splitarray sharding, codebook broadcast from rank 0, all-reduce of objective sums and counters,
padded gather of ragged shards -- with an ORACLE-backed shard encoder injected (tests only; the
product's default shard encoder is the HIP engine and refuses to run without a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 256


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_shard_encoder(X, B0, K, m, ilsiters, icmiter, npert, randord, seed, global_offset):
    import oracle as O
    Bs, objs, stats = O.encode_icm(X.numpy(), B0.numpy().astype(np.int16) + 1, K.numpy(), m, H, ilsiters, icmiter, npert,
                                   randord, seed, global_offset=global_offset, want_stats=True)
    n = X.shape[0]
    return torch.from_numpy((Bs - 1).astype(np.uint8)), objs.astype(np.float64) * n, stats


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    from conftest import make_problem
    lsq = importlib.import_module("local-search-quantization_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, m = 32, 4
        X, K, B0 = make_problem(d, n, m, seed=77, kind="gauss")
        s, e = lsq.distributed.shard_range(n, world, rank)
        Kt = torch.from_numpy(K.copy()) if rank == 0 else torch.zeros((m * H, d), dtype=torch.float32)   # only rank 0 has C
        codes, objs, stats, gathered = lsq.distributed.encode_sharded(
            torch.from_numpy(X[s:e].copy()), torch.from_numpy((B0[s:e] - 1).astype(np.uint8)), Kt, m, [1, 3], 2, 2, True, 5,
            n_total=n, shard_start=s, shard_encoder=_oracle_shard_encoder, gather_codes=True)
        assert torch.equal(Kt, torch.from_numpy(K)), "codebooks were not broadcast"
        q.put((rank, (s, e), codes.numpy(), objs, stats, None if gathered is None else gathered.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64])
def test_two_rank_sharded_encode_equals_single_process(oracle, n):
    from conftest import make_problem
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, K, B0 = make_problem(32, n, 4, seed=77, kind="gauss")
    Bs_ref, objs_ref, st_ref = oracle.encode_icm(X, B0, K, 4, H, [1, 3], 2, 2, True, 5, want_stats=True)
    # ranges are splitarray(1:n, 2): first n mod 2 shards one longer
    assert res[0][1] == (0, (n + 1) // 2) and res[1][1] == ((n + 1) // 2, n)
    both = np.concatenate([res[0][2], res[1][2]], axis=1).astype(np.int16) + 1
    assert np.array_equal(both, Bs_ref)                                   # P8: sharding-invariant codes
    for r in (0, 1):
        assert np.allclose(res[r][3], objs_ref, rtol=1e-6)                # all-reduced objective, same on every rank
        assert np.array_equal(res[r][4], st_ref.astype(np.int64))
    assert res[1][5] is None
    assert np.array_equal(res[0][5].astype(np.int16) + 1, Bs_ref)         # padded gather of ragged shards on rank 0


def test_product_shard_encoder_refuses_cpu(lsq):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lsq.distributed.encode_sharded(torch.zeros(4, 8), torch.zeros(4, 2, dtype=torch.uint8), torch.zeros(512, 8), 2, [1], 1, 1,
                                       True, 0, n_total=4, shard_start=0)


# ---- the search step sharded the same way (database partitions, one all-gather of the per-rank nearest lists) ----------------------------
def _host_shard_scanner(codes, Q, K, dbnorms, m, knn):
    """this library's HOST scan as the injected per-shard scanner (the product default is the device scan and refuses to run without a GPU)"""
    import importlib
    lsq = importlib.import_module("local-search-quantization_amd")
    L = lsq._lib.load()
    c, q, k, nrm = (np.ascontiguousarray(t.numpy()) for t in (codes, Q, K, dbnorms))
    dists = np.zeros((q.shape[0], knn), np.float32)
    ids = np.zeros((q.shape[0], knn), np.int32)
    lsq._lib.check(L.lsq_linscan_aqd_query_extra_byte(dists.ctypes.data, ids.ctypes.data, c.ctypes.data, q.ctypes.data, k.ctypes.data, nrm.ctypes.data,
                                                      q.shape[0], c.shape[0], m, H, q.shape[1], knn, 2))
    return torch.from_numpy(dists), torch.from_numpy(ids)


def _search_case(n):
    rng = np.random.default_rng(n)
    d, m, nq = 16, 4, 9
    K = (rng.standard_normal((m * H, d)) * 0.5).astype(np.float32)
    codes = rng.integers(0, H, size=(n, m), dtype=np.uint8)
    codes[n // 2:] = codes[: n - n // 2]                 # duplicated entries: exact ties ACROSS the two shards, ordered by global id
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    recon = sum(K[j * H + codes[:, j].astype(np.int64)] for j in range(m))
    dbnorms = (recon.astype(np.float64) ** 2).sum(1).astype(np.float32)
    return codes, Q, K, dbnorms, m


def _search_worker(rank, world, port, n, knn, q):
    sys.path.insert(0, ROOT)
    import importlib
    lsq = importlib.import_module("local-search-quantization_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        codes, Q, K, dbnorms, m = _search_case(n)
        s, e = lsq.distributed.shard_range(n, world, rank)
        Kt = torch.from_numpy(K.copy()) if rank == 0 else torch.zeros_like(torch.from_numpy(K))      # only rank 0 has codebooks and queries
        Qt = torch.from_numpy(Q.copy()) if rank == 0 else torch.zeros_like(torch.from_numpy(Q))
        dd, ii = lsq.distributed.search_sharded(torch.from_numpy(codes[s:e].copy()), torch.from_numpy(dbnorms[s:e].copy()), Qt, Kt, m, knn,
                                                n_total=n, shard_start=s, shard_scanner=_host_shard_scanner)
        q.put((rank, dd.numpy(), ii.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,knn", [(301, 40), (64, 64), (50, 30)])
def test_two_rank_sharded_search_equals_one_scan(lsq, n, knn):
    """knn = 64 = n: every shard is SMALLER than knn (padded lists); knn = 30 > the 25-entry shards likewise"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, world, port, n, knn, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=180)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    codes, Q, K, dbnorms, m = _search_case(n)
    dref, iref = _host_shard_scanner(*(torch.from_numpy(a) for a in (codes, Q, K, dbnorms)), m, knn)
    for r in (0, 1):
        assert np.array_equal(res[r][2], iref.numpy()) and np.array_equal(res[r][1], dref.numpy())


def test_product_shard_scanner_refuses_cpu(lsq):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lsq.distributed.search_sharded(torch.zeros(4, 2, dtype=torch.uint8), torch.zeros(4), torch.zeros(1, 8), torch.zeros(512, 8), 2, 1,
                                       n_total=4, shard_start=0)


def test_merge_puts_padding_after_nan_results():
    """a shard shorter than knn pads with (+inf, 2^31 - 1); a genuine NaN-distance result must still come before the padding."""
    import importlib
    import torch
    D = importlib.import_module("local-search-quantization_amd.distributed")
    nan, inf = float("nan"), float("inf")
    d_cat = torch.tensor([[1.0, nan, inf, 0.5, inf, inf]], dtype=torch.float32)
    i_cat = torch.tensor([[7, 3, D.NOID, 9, 11, D.NOID]], dtype=torch.int32)
    d, i = D._merge_candidates(d_cat, i_cat, 4)
    assert i.tolist() == [[9, 7, 11, 3]]
    assert d[0, 0] == 0.5 and d[0, 1] == 1.0 and d[0, 2] == inf and torch.isnan(d[0, 3])
