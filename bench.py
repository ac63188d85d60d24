#!/usr/bin/env python
"""bench.py -- headline benchmark: vectors encoded / second (ICM, m=8, h=256) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1 by default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full `encode_icm_cuda`-equivalent call: pair tables + unary build +
16 ILS iterations x (perturb + 4 sweeps x m node updates + cost + accept) over this rank's batch,
inputs already resident in HBM, through the C-ABI (lsq_encode_icm_dev).  Workload = BASELINE.json
configs[1]: SIFT1M-shaped base set (10^6 x 128 f32, synthetic: Philox uniform integers 0..255),
m = 8, h = 256, ILS 16, icmiter 4, npert 4, randord -- per GPU (weak scaling: every rank encodes its
own 10^6-vector shard of a global index space; the only data-path collective is the RCCL broadcast
of the 1 MiB codebook matrix from rank 0, inside the timed step).

Prints ONE JSON line on rank 0 with the driver's contract fields plus `roofline` (dominant kernel:
the ICM node update) and `cpu_baseline` (the oracle's structure-faithful port of the reference CPU
path, timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # same guide, L2 aggregate measured
PMC_FILE, PMC_SCHEDULE = "r01h_pmc_per_kernel.json", 4      # committed PMC passes the `traffic` field is read from


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--vectors", dest="n", type=int, default=1_000_000, help="vectors per GPU")
    p.add_argument("--dim", dest="d", type=int, default=128)
    p.add_argument("--codebooks", dest="m", type=int, default=8)
    p.add_argument("--ils", type=int, default=16)
    p.add_argument("--icmiter", type=int, default=4)
    p.add_argument("--npert", type=int, default=4)
    p.add_argument("--schedule", type=int, default=int(os.environ.get("LSQ_SCHEDULE", "4")))
    p.add_argument("--chunk", type=int, default=int(os.environ.get("LSQ_CHUNK", "0")))
    p.add_argument("--skip", type=int, default=int(os.environ.get("LSQ_SKIP", "1")),
                   help="schedules 3/4: exact memoisation of node updates whose inputs did not change (1) or recompute everything (0)")
    p.add_argument("--ablation", type=int, default=int(os.environ.get("LSQ_ABLATION", "0")), help="timing-only kernel ablation (results invalid when != 0)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget")
    return p.parse_args()


def cpu_baseline(args):
    """Time the oracle's structure-faithful restatement of encoding_icm (reference
    src/encodings/encode_icm.jl:131-189 loop nest, one OpenMP thread per `julia -p` worker) on a
    bounded sample of the same workload.  Never part of the measured GPU path."""
    import oracle as O
    O.build()
    cores = O.num_threads()
    d, m, h = args.d, args.m, 256
    nw = 4096                                 # vectors per worker (ub = 4 MiB, unaries = 32 MiB per worker)
    n = nw * cores
    X = O.synth_data_u8(1234, n, d)
    pool = O.synth_data_u8(4321, m * h, d)
    K = np.ascontiguousarray(pool / np.float32(m))
    B = O.randinit(7, n, m, h)
    t0 = time.perf_counter()
    B = O.encoding_icm_faithful(X, B, K, m, h, args.icmiter, True, args.npert, 42, 0, nworkers=cores)
    t1 = time.perf_counter() - t0
    iters = int(max(1, min(args.ils - 1, (args.cpu_seconds - t1) // max(t1, 1e-9))))
    t0 = time.perf_counter()
    for it in range(1, 1 + iters):
        B = O.encoding_icm_faithful(X, B, K, m, h, args.icmiter, True, args.npert, 42, it, nworkers=cores)
    t_iter = (time.perf_counter() - t0) / iters
    vps = n / (t_iter * args.ils)             # full encode = args.ils ILS iterations
    # also the cache-blocked per-vector CPU variant (one vector's unaries stay in L1/L2, tables shared
    # in L3), so the GPU/CPU ratio is not inflated by the reference's whole-array loop order alone.
    nb = 64 * cores
    Xb, Bb = X[:nb], O.randinit(7, nb, m, h)
    t0 = time.perf_counter()
    O.encode_icm(Xb, Bb, K, m, h, [args.ils], args.icmiter, args.npert, True, 42)
    tb = time.perf_counter() - t0
    reps = int(max(1, min(16, 4.0 // max(tb, 1e-9))))
    nb2 = min(n, nb * reps)
    Xb, Bb = X[:nb2], O.randinit(7, nb2, m, h)
    t0 = time.perf_counter()
    O.encode_icm(Xb, Bb, K, m, h, [args.ils], args.icmiter, args.npert, True, 42)
    tb = time.perf_counter() - t0
    blocked = {"value": nb2 / tb, "unit": "vectors/s", "cores": cores,
               "sample": "%d vectors x %d ILS iterations in %.2f s; oracle/lsq_oracle.c orc_encode_icm (per-vector, "
                         "cache-blocked loop order; not the reference's)" % (nb2, args.ils, tb)}
    return {
        "blocked_variant": blocked,
        "value": vps, "unit": "vectors/s", "cores": cores, "kind": "port",
        "sample": "%d vectors (%d per worker x %d OpenMP workers), %d of %d ILS iterations timed (%.2f s each), "
                  "scaled linearly to %d iterations; oracle/lsq_oracle.c orc_encoding_icm_faithful "
                  "(CPU restatement of the reference algorithm, not Julia)" % (n, nw, cores, iters, args.ils, t_iter, args.ils),
    }


def main():
    args = parse()
    import torch
    lsq = importlib.import_module("local-search-quantization_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev                      # one rank per GPU in production; modulo only for 1-GPU smoke runs
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LSQ_BENCH_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; gloo only for 1-GPU smoke tests
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    n, d, m, h = args.n, args.d, args.m, 256
    eng = lsq.Engine(dev_index, profile=True, schedule=args.schedule, chunk=(args.chunk or None))
    eng.set_option("skip", args.skip)
    eng.set_option("ablation", args.ablation)
    goff = rank * n
    data_tag = "synthetic"
    dX = None
    sift = os.path.join(os.environ.get("LSQ_DATA_DIR", ""), "sift", "sift_base.fvecs")
    if os.environ.get("LSQ_DATA_DIR") and os.path.exists(sift) and d == 128:
        # use the real SIFT1M base set when it is there (it is not in this image); rank r takes rows [r n, (r+1) n)
        rows_in_file = os.path.getsize(sift) // (4 + 4 * 128)
        if goff + n <= rows_in_file:
            X = lsq.fvecs_read((goff + 1, goff + n), sift)              # (d, n), 1-based inclusive bounds like the reference reader
            dX = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:%d" % dev_index)
            data_tag = "sift1m_base rows %d..%d (%s)" % (goff, goff + n - 1, sift)
    if dX is None:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
    dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
    dK = eng.synth_codebooks_dev(4321, m, d) if rank == 0 else torch.zeros((m * h, d), dtype=torch.float32, device=dX.device)
    dBs = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
    torch.cuda.synchronize()

    def step():
        if dist is not None:
            dist.broadcast(dK, src=0)                  # RCCL over xGMI: the one data-path collective
        return eng.encode_icm_dev(dX, dB0, dK, m, [args.ils], args.icmiter, args.npert, True, seed=42,
                                  global_offset=goff, out=dBs)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.reset_timings()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, sums, stats = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dX.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = eng.timings()

    if rank == 0:
        total_vectors = n * world * args.steps
        value = total_vectors / dt
        launches = max(tm["icm_launches"], 1)
        avg_launch_s = tm["icm_ms"] * 1e-3 / launches
        node_updates_per_launch = tm["icm_node_updates"] / launches
        cs = 8 if m <= 8 else 16
        if args.schedule != 1:
            # per vector per node-update launch: U_j row (4h B) + code record read + 1 code byte written
            hbm_bytes = node_updates_per_launch * (4 * h + cs + 1)
        else:
            # fused sweeps: per vector per launch all m unary rows + code record read/write
            hbm_bytes = (n if args.chunk == 0 else min(n, args.chunk)) * (4 * h * m + 2 * cs)
        table_bytes = node_updates_per_launch * (m - 1) * 4 * h        # table columns: on-chip (L2 gathers or LDS reads)
        achieved = hbm_bytes / avg_launch_s / 1e9
        resolved = n * (args.icmiter * m if args.schedule in (1, 4) else 1)    # vector x node updates one launch resolves
        roof = {
            "kernel": {0: "icm_node_kernel<%d>", 1: "icm_fused_kernel<%d>", 2: "icm_slice_kernel<%d,SL> + icm_combine_kernel", 3: "icm_walk_kernel<%d,SL>", 4: "icm_walk_kernel<%d,SL>"}[args.schedule] % m,
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "avg_launch_us": avg_launch_s * 1e6, "launches": int(tm["icm_launches"]),
            "algorithmic_bytes_per_launch": hbm_bytes,
            "bytes_per_node_update": 4 * h + cs + 1,
            "node_updates_recomputed_per_launch": node_updates_per_launch,
            "node_updates_resolved_per_launch": resolved,
            "recomputed_fraction": node_updates_per_launch / resolved,
            "note": "achieved counts only node updates that were actually recomputed; with skip=1 the others are resolved by "
                    "exact memoisation (inputs unchanged since the node was last minimised) and move no bytes",
        }
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this build (separate --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE runs of this same command; FETCH_SIZE x2 = the gfx950 correction for wide streaming reads,
        # /opt/skills/guides/MI355X_MICROARCH.md section HBM).  Not measurable live inside the benchmark process.
        try:
            if args.schedule == PMC_SCHEDULE and args.skip and n == 1_000_000 and d == 128 and m == 8 and args.ils == 16 and args.icmiter == 4:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                wk = [k for k in pmc if k.startswith("icm_walk_kernel")][0]
                roof["traffic"] = (2.0 * pmc[wk]["FETCH_SIZE"]["mean_per_dispatch"] + pmc[wk]["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
                roof["traffic_source"] = "profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, per dispatch)"
        except Exception:
            roof["traffic"] = None
        if args.schedule in (0, 1):
            roof["l2_gather"] = {"achieved": table_bytes / avg_launch_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                                 "frac": table_bytes / avg_launch_s / 1e9 / L2_PEAK_GBS}
        else:
            roof["lds_table_reads"] = {"achieved": table_bytes / avg_launch_s / 1e9, "peak": 150000.0, "unit": "GB/s",
                                       "frac": table_bytes / avg_launch_s / 1e9 / 150000.0,
                                       "note": "(m-1) x 1 KiB of table columns per node update come from LDS-staged slices (ds_read_b128 aggregate peak)"}
        out = {
            "metric": "vectors encoded/sec (ICM, m=%d h=%d)" % (m, h),
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": data_tag,
            "config": {
                "workload": "BASELINE configs[1]: SIFT1M-shaped base encode, %d x %d f32 per GPU, m=%d, h=%d, %d ILS iters x %d ICM sweeps, "
                            "npert=%d, randord, seed=42; inputs resident in HBM; lsq_encode_icm_dev" % (n, d, m, h, args.ils, args.icmiter, args.npert),
                "vectors_per_gpu": n, "d": d, "m": m, "h": h, "ils_iters": args.ils, "icm_iters": args.icmiter, "npert": args.npert,
                "schedule": {0: "per-node launches, L2 gathers (M2 data-flow)", 1: "fused sweeps per ILS iteration (M1 data-flow)",
                             2: "per-node launches, LDS-staged table slices + combine, slice-major U stream (M2 data-flow)",
                             3: "per-node launches, one block walks all LDS-staged slices, slice-major U stream (M2 data-flow)",
                             4: "one launch per ILS iteration (icmiter x m node updates back to back; a block owns its vectors and walks "
                                "all LDS-staged slices), slice-major U stream (M2 data-flow)"}[args.schedule],
                "skip_unchanged_nodes": bool(args.skip) and args.schedule >= 3,
                "parallelism": "%d x independent shards, RCCL broadcast of codebooks" % world,
            },
            "objective": float(sums[0] / n), "last_ils_pct_better": float(100.0 * stats[-1, 1] / n),
            "roofline": roof,
            "time_breakdown_ms_per_step": {k: tm[k] / args.steps for k in ("tables_ms", "unaries_ms", "perturb_ms", "icm_ms", "cost_ms", "other_ms")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
