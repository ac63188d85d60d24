"""Sharded (multi-GPU) encoding: one process per GPU, torch.distributed (backend "nccl" = RCCL).

The path shards embarrassingly: every vector's ILS/ICM trajectory depends only on
(x_i, B0_i, C, RNG(seed, global index i, iteration)), exactly like the reference's worker split
(src/encodings/encode_icm.jl:165-173, splitarray of src/utils.jl:152-177).  So:

  * rank r owns the contiguous range splitarray(1:n, world)[r];
  * ONE collective carries data-path state: the broadcast of the codebooks K from rank 0 (1-8 MiB,
    latency-bound over xGMI).  Pair tables / unaries are rebuilt locally on every GPU;
  * no collective inside the ICM sweep;
  * the objective sums and the ==/< counters are all-reduced (a few doubles), and codes are
    optionally gathered to rank 0.

The shard encoder is injectable so the collective logic is testable on CPU with gloo; the product
default is the HIP engine and raises if no GPU is present.
"""
import numpy as np

from . import engine as _engine


def shard_range(n, world, rank):
    return _engine.splitarray(n, world)[rank]


def _hip_shard_encoder(device_index):
    eng = _engine.Engine(device_index)

    def run(X, B0, K, m, ilsiters, icmiter, npert, randord, seed, global_offset):
        import torch
        if isinstance(X, torch.Tensor) and X.is_cuda:
            dBs, sums, stats = eng.encode_icm_dev(X, B0, K, m, ilsiters, icmiter, npert, randord, seed=seed,
                                                  global_offset=global_offset)
            return dBs, sums, stats
        # host numpy shard: codes are 1-based int16 at this boundary
        raise TypeError("the HIP shard encoder takes device-resident torch tensors")

    run.engine = eng
    return run


def encode_sharded(X_shard, B0_shard, K, m, ilsiters, icmiter, npert, randord, seed, n_total, shard_start,
                   group=None, shard_encoder=None, gather_codes=False):
    """Encode this rank's shard; returns (codes_shard, objs, stats[, gathered]) where
    objs = global objective per checkpoint (float32), stats = global (I, 2) counters.

    X_shard/B0_shard/K: device torch tensors for the HIP encoder ((n_r, d) f32, (n_r, m) u8 0-based,
    (m*256, d) f32).  K needs to be valid on rank 0 only: it is broadcast in place.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > 1:
        dist.broadcast(K, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if shard_encoder is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the sharded encoder has no CPU fallback")
        shard_encoder = _hip_shard_encoder(torch.cuda.current_device())
    codes, sums, stats = shard_encoder(X_shard, B0_shard, K, m, ilsiters, icmiter, npert, randord, seed, shard_start)
    red = torch.as_tensor(np.concatenate([np.asarray(sums, dtype=np.float64).ravel(),
                                          np.asarray(stats, dtype=np.float64).ravel()]))
    if world > 1:
        if K.is_cuda:
            red = red.to(K.device)
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
        red = red.cpu()
    red = red.numpy()
    nr = len(np.atleast_1d(sums))
    objs = (red[:nr] / max(n_total, 1)).astype(np.float32)
    gstats = red[nr:].round().astype(np.int64).reshape(-1, 2)
    if not gather_codes:
        return codes, objs, gstats
    gathered = None
    if world > 1:
        # variable shard sizes (splitarray): pad to the longest shard, gather, trim on rank 0
        sizes = [e - s for s, e in _engine.splitarray(n_total, world)]
        nmax = max(sizes)
        ct = codes if isinstance(codes, torch.Tensor) else torch.as_tensor(np.asarray(codes))
        pad = torch.zeros((ct.shape[0], nmax, ct.shape[2]), dtype=ct.dtype, device=ct.device)
        pad[:, : ct.shape[1]] = ct
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, bufs, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if rank == 0:
            gathered = torch.cat([b[:, :sz] for b, sz in zip(bufs, sizes)], dim=1)
    else:
        gathered = codes
    return codes, objs, gstats, gathered


# ---- the search step, sharded the same way ---------------------------------------------------------------------------------------------
def _hip_shard_scanner(device_index):
    eng = _engine.Engine(device_index)

    def run(codes, Q, K, dbnorms, m, knn):
        return eng.linscan_dev(codes, Q, K, dbnorms, m, knn)

    run.engine = eng
    return run


_SCANNERS = {}      # device index -> the default shard scanner of search_sharded


def _pair_keys(dists, ids):
    """(dist, id) pairs -> int64 keys whose order is the pairs' lexicographic order (std::pair<float,int>'s, the reference's partial_sort order,
    linscan_aqd_pairwise_byte.cpp:84): order-preserving float bits << 31 | id.  NaN distances sort last."""
    import torch
    bits = dists.contiguous().view(torch.int32).to(torch.int64)
    u = bits & 0xFFFFFFFF
    k = torch.where(bits < 0, u ^ 0xFFFFFFFF, u ^ 0x80000000)
    k = torch.where(torch.isnan(dists), torch.full_like(k, 0xFFFFFFFF), k)
    return (k << 31) | ids.to(torch.int64)


NOID = 2 ** 31 - 1      # id of a padding entry (a shard shorter than knn)


def _merge_candidates(d_cat, i_cat, knn):
    """The knn smallest (distance, id) pairs per row of the gathered candidates.  Padding sorts after EVERYTHING, a genuine NaN-distance result
    included (lsq_multi_linscan and the single-device scan return such a result last)."""
    import torch
    keys = _pair_keys(d_cat, i_cat)
    keys = torch.where(i_cat == NOID, torch.full_like(keys, torch.iinfo(torch.int64).max), keys)
    order = torch.argsort(keys, dim=1)[:, :knn]
    return torch.gather(d_cat, 1, order), torch.gather(i_cat, 1, order)


def search_sharded(codes_shard, dbnorms_shard, Q, K, m, knn, n_total, shard_start, group=None, shard_scanner=None):
    """ADC linear scan (src/linscan/Linscan.jl:46-73) over a database sharded like the encode: rank r holds the codes and norms of
    splitarray(1:n, world)[r].  Every rank scans its share for its own knn nearest (fewer if the share is smaller), ids are made global
    (1-based, + shard_start), ONE all-gather carries the world x knn candidates per query and every rank merges them by (distance, id) --
    the result is what one scan of the whole database returns, ties included.  Q and K need to be valid on rank 0 only (broadcast in place).
    -> dists (nq, knn) float32 ascending, ids (nq, knn) int32, on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if knn > n_total:
        raise ValueError("knn = %d exceeds the database size %d" % (knn, n_total))
    if world > 1:
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(K, src=src, group=group)
        dist.broadcast(Q, src=src, group=group)
    if shard_scanner is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the sharded search has no CPU fallback")
        dev = torch.cuda.current_device()
        if dev not in _SCANNERS:                 # one Engine (and its multi-GB scan buffers) per device, not per call
            _SCANNERS[dev] = _hip_shard_scanner(dev)
        shard_scanner = _SCANNERS[dev]
    n_loc = int(codes_shard.shape[0])
    k_loc = min(knn, n_loc)
    nq = int(Q.shape[0])
    INF = float("inf")
    d_pad = torch.full((nq, knn), INF, dtype=torch.float32, device=Q.device)
    i_pad = torch.full((nq, knn), NOID, dtype=torch.int32, device=Q.device)
    if k_loc > 0 and nq > 0:
        d_loc, i_loc = shard_scanner(codes_shard, Q, K, dbnorms_shard, m, k_loc)
        d_pad[:, :k_loc] = torch.as_tensor(d_loc).to(Q.device)
        i_pad[:, :k_loc] = torch.as_tensor(i_loc).to(Q.device) + int(shard_start)
    if world == 1:
        return d_pad, i_pad
    d_all = [torch.empty_like(d_pad) for _ in range(world)]
    i_all = [torch.empty_like(i_pad) for _ in range(world)]
    dist.all_gather(d_all, d_pad, group=group)
    dist.all_gather(i_all, i_pad, group=group)
    return _merge_candidates(torch.cat(d_all, dim=1), torch.cat(i_all, dim=1), knn)
