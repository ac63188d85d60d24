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
