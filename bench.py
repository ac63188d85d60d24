#!/usr/bin/env python
"""bench.py -- headline benchmark: vectors encoded / second (ICM, m=8, h=256) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1 by default; N > 1 without a launcher: the script starts its own N ranks
                                                              under torch.distributed.run on a free loopback port and rank 0 prints the line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (the driver's form: same ranks, same line)

A "step" = one full `encode_icm_cuda`-equivalent call: pair tables + unary build +
16 ILS iterations x (perturb + 4 sweeps x m node updates + cost + accept) over this rank's batch,
inputs already resident in HBM, through the C-ABI (lsq_encode_icm_dev).  Default workload = BASELINE.json
configs[1]: SIFT1M-shaped base set (10^6 x 128 f32, synthetic: Philox uniform integers 0..255),
m = 8, h = 256, ILS 16, icmiter 4, npert 4, randord -- per GPU (weak scaling: every rank encodes its
own 10^6-vector shard of a global index space; the only data-path collective is the RCCL broadcast
of the 1 MiB codebook matrix from rank 0, inside the timed step).

Other BASELINE configs through flags (same JSON line, `config.workload` names what ran):
    cfg3  --codebooks 16
    cfg4  --scaling strong --total 1000000 --dim 960     (GIST-shaped; `splitarray` shards: 125 000 vectors per GPU at N = 8)
    cfg5  --vectors 12500000                              (weak scaling, 12.5 M vectors per GPU generated on the device, 13 resident chunks)

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  `roofline`         dominant kernel (the ICM node update): algorithmic HBM bytes / HIP-event launch time vs 8 TB/s, the LDS-side
                     gather rate vs the guide's 150 TB/s, the M1 compulsory-bytes fraction, PMC traffic when the committed
                     profile was taken from THIS build (hash of the loaded .so), else null;
  `trained`          the representative number beside `value`: the same encode with codebooks TRAINED by this package (copied from workloads.trained);
  `workloads`        the SAME 10^6-vector encode on other inputs / options, because the GPU time is data-dependent (exact memoisation of
                     unchanged node updates + the 16-bit filter): `trained` (codebooks trained by this package's own train_lsq on a 100 000-vector
                     sample -- the reference encodes with trained codebooks, LSQ.jl:10-88 -> demo_lsq_gpu.jl:33-50), `floor` (memoisation off:
                     every node update recomputed, filtered walk and f32 walk), `heavy_tailed` (Cauchy-scaled vectors: out-of-range unaries);
  `search`           the step after the path on the codes just produced: the device ADC scan (10^4 queries, 1000 neighbours) with its LDS-gather
                     roofline, the host scan and the reference's own build as CPU figures and checkers;
  `sample_parity`    two 256-vector blocks of the timed output re-computed with the CPU oracle after the timed region;
  `north_star_point` the same workload at north_star's own operating point (4 ILS iterations), with its CPU baseline;
  `end_to_end`       the host-buffer entry point lsq_encode_icm on pageable host memory (H2D of X, D2H of the codes included);
  `cpu_baseline`     the oracle's structure-faithful port of the reference CPU path on this box's host cores (bounded sample),
                     plus cfg1 at its exact shape; rank 0, N=1 only.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (~6.3 TB/s achievable)
HBM_ACHIEVABLE_GBS = 6300.0
LDS_PEAK_GBS = 150000.0        # same guide, section LDS: ds_read_b64/b128 aggregate with every CU streaming (reproduced: profiles/r05_ubench_lds.txt)
LDS_PATTERN_GBS = LDS_PEAK_GBS / 2.125      # 16 random 64-byte rows per ds_read_b128 (four per 16-lane group): E[max rows per bank quarter] = 2.125 cycles per group (the ADC scan; the walk before its rotated placement)


def walk_conflict_factor(m):
    """LDS cycles per 16-lane group and table read of the filtered walk's ROTATED slice table (lsq_q16.h, WalkqRot): a line holds spl slots (4 up to m = 8, 8 above);
    the reads of a full group of spl tables are conflict-free, a group of nt < spl tables rotates modulo nt: ceil(spl / nt) vectors share a slot."""
    ntab = m - 1
    if ntab <= 0:
        return 1.0
    spl = 4 if m <= 8 else 8
    nt0 = min(spl, ntab)
    nt1 = ntab - nt0
    cyc = nt0 * -(-spl // nt0) + (nt1 * -(-spl // nt1) if nt1 else 0)
    return cyc / ntab
PMC_GLOB = "r[0-9][0-9]*_pmc_per_kernel.json"      # committed PMC passes; `traffic` is read from the newest one whose build hash matches


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--vectors", dest="n", type=int, default=1_000_000, help="vectors per GPU (weak scaling)")
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    p.add_argument("--total", type=int, default=1_000_000, help="strong scaling: total vectors, split with splitarray over the ranks")
    p.add_argument("--dim", dest="d", type=int, default=128)
    p.add_argument("--codebooks", dest="m", type=int, default=8)
    p.add_argument("--ils", type=int, default=16)
    p.add_argument("--icmiter", type=int, default=4)
    p.add_argument("--npert", type=int, default=4)
    p.add_argument("--schedule", type=int, default=int(os.environ.get("LSQ_SCHEDULE", "-1")), help="-1 = library default")
    p.add_argument("--chunk", type=int, default=int(os.environ.get("LSQ_CHUNK", "0")))
    p.add_argument("--skip", type=int, default=int(os.environ.get("LSQ_SKIP", "1")),
                   help="exact memoisation of node updates whose inputs did not change (1) or recompute everything (0)")
    p.add_argument("--option", action="append", default=[], help="extra engine option key=value (repeatable)")
    p.add_argument("--tuning", action="store_true", help="load liblsq_mi355x_tuning.so (ablations / knobs / schedules 0..2); never the headline")
    p.add_argument("--ablation", type=int, default=int(os.environ.get("LSQ_ABLATION", "0")), help="tuning build only; results invalid when != 0")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra-legs", action="store_true", help="skip sample_parity / north_star_point / end_to_end / the busy tail")
    p.add_argument("--no-sample-parity", action="store_true", help="skip the oracle re-encode of 512 vectors of the timed output (profiling runs)")
    p.add_argument("--trained-codebooks", action="store_true",
                   help="time the MAIN loop on codebooks trained by this package's train_lsq on the first 100 000 vectors (the representative workload: "
                        "profiling runs key their PMC files on it); default: SURVEY 8(d)'s synthetic codebooks, the trained ones appear as `trained`")
    p.add_argument("--no-workloads", action="store_true", help="skip the `workloads` leg (trained codebooks / no-memoisation floor / heavy tails)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget for the cfg2 sample")
    p.add_argument("--min-gpu-seconds", type=float, default=6.0,
                   help="after the timed region keep running identical UNTIMED steps until the GPU legs have lasted this long "
                        "(lets a 5 s utilisation sampler see the device busy; reported as steady_state)")
    p.add_argument("--multi-leg", action="store_true", help="also time the single-process multi-GPU entry point (lsq_multi_*) on all visible devices")
    return p.parse_args()


def lib_hash(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(args):
    """Time the oracle's structure-faithful restatement of encoding_icm (reference src/encodings/encode_icm.jl:131-189 loop
    nest, one OpenMP thread per `julia -p` worker) on bounded samples.  Never part of the measured GPU path.
      * cfg2 sample: a 100 000-vector subset split over the workers, as many of the 16 ILS iterations as fit the
        budget, scaled linearly (the CPU restatement, like the reference, executes EVERY node update of every vector: its time does not depend
        on the values -- unlike the GPU path, whose exact memoisation and 16-bit filter make the time data-dependent; see `workloads`);
      * cfg1 exactly: n = 10 000, d = 128, m = 8, one call = one ILS iteration with 4 sweeps (demo_lsq.jl:34);
      * the cache-blocked per-vector variant, so the ratio is not inflated by the reference's loop order alone."""
    import oracle as O
    O.build()
    cores = O.num_threads()
    d, m, h = args.d, args.m, 256
    n = 100_000
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
    out = {
        "value": vps, "unit": "vectors/s", "cores": cores, "kind": "port",
        "sample": "100 000-vector subset (%d per worker x %d OpenMP workers), %d of %d ILS iterations timed (%.3f s each), "
                  "scaled linearly to %d iterations; oracle/lsq_oracle.c orc_encoding_icm_faithful "
                  "(CPU restatement of the reference algorithm, not Julia)" % (n // cores, cores, iters, args.ils, t_iter, args.ils),
        "seconds_per_ils_iteration": t_iter,
    }
    # cfg1 at its exact shape (d = 128, m = 8 whatever the GPU flags say): ilsiter = 8 chained calls, first one discarded
    n1 = 10_000
    X1 = O.synth_data_u8(1234, n1, 128)
    K1 = np.ascontiguousarray(O.synth_data_u8(4321, 8 * h, 128) / np.float32(8))
    B1 = O.randinit(7, n1, 8, h)
    B1 = O.encoding_icm_faithful(X1, B1, K1, 8, h, 4, True, 4, 42, 0, nworkers=cores)
    t0 = time.perf_counter()
    for it in range(1, 8):
        B1 = O.encoding_icm_faithful(X1, B1, K1, 8, h, 4, True, 4, 42, it, nworkers=cores)
    t_call = (time.perf_counter() - t0) / 7
    out["cfg1"] = {"value": n1 / t_call, "unit": "vectors/s per encoding_icm call (1 ILS iteration, 4 sweeps)", "cores": cores,
                   "sample": "BASELINE configs[0] exactly: n = 10 000, d = 128, m = 8, h = 256; 7 chained calls timed (%.4f s each)" % t_call}
    # cache-blocked per-vector variant (one vector's unaries stay in L1/L2, tables shared in L3)
    nb = 64 * cores
    Xb, Bb = X[:nb], O.randinit(7, nb, m, h)
    t0 = time.perf_counter()
    O.encode_icm(Xb, Bb, K, m, h, [args.ils], args.icmiter, args.npert, True, 42)
    tb = time.perf_counter() - t0
    reps = int(max(1, min(12, 4.0 // max(tb, 1e-9))))
    nb2 = min(n, nb * reps)
    Xb, Bb = X[:nb2], O.randinit(7, nb2, m, h)
    t0 = time.perf_counter()
    O.encode_icm(Xb, Bb, K, m, h, [args.ils], args.icmiter, args.npert, True, 42)
    tb = time.perf_counter() - t0
    out["blocked_variant"] = {"value": nb2 / tb, "unit": "vectors/s", "cores": cores,
                              "sample": "%d vectors x %d ILS iterations in %.2f s; oracle/lsq_oracle.c orc_encode_icm (per-vector, "
                                        "cache-blocked loop order; not the reference's)" % (nb2, args.ils, tb)}
    return out


def sample_parity(eng, dX, dB0, dK, dBs, n, m, args, goff):
    """Untimed: two contiguous 256-vector blocks of the TIMED output vs the CPU oracle run on exactly those vectors
    (valid because results depend on the global vector index only).  -> (ok, detail)"""
    import oracle as O
    O.build()
    K = dK.cpu().numpy()
    bad, checked = 0, 0
    for a in sorted({0, max(0, n // 2 - 128)}):
        b = min(n, a + 256)
        X = dX[a:b].cpu().numpy()
        B0 = dB0[a:b].cpu().numpy().astype(np.int16) + 1
        ref, _ = O.encode_icm(X, B0, K, m, 256, [args.ils], args.icmiter, args.npert, True, 42, global_offset=goff + a)
        got = dBs[0][a:b].cpu().numpy().astype(np.int16) + 1
        bad += int((ref[0] != got).any(axis=1).sum())
        checked += b - a
    return bad == 0, "%d vectors of the timed output re-encoded by oracle/lsq_oracle.c: %d differ" % (checked, bad)


def workload_key(argv):
    """the shape-defining flags of a bench.py command line, canonical: a PMC profile only speaks for the workload it was taken on"""
    keep, shape = [], ("--vectors", "--scaling", "--total", "--dim", "--codebooks", "--ils", "--icmiter", "--npert", "--schedule", "--skip", "--option", "--chunk")
    it = iter(argv)
    for a in it:
        if a == "--trained-codebooks":
            keep.append(a)
        elif a in shape:
            keep.append(a + "=" + next(it, ""))
        elif any(a.startswith(k + "=") for k in shape):
            keep.append(a)
    return " ".join(sorted(keep))


def pmc_traffic(lib_sha):
    """HBM traffic per ICM launch from the newest committed PMC profile taken from THIS build (its `_build.lib_sha16` must equal
    the hash of the loaded .so) ON THIS WORKLOAD (same shape flags); None when there is none -- a stale or foreign profile is never
    reported next to fresh timings."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", PMC_GLOB))):
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("_build", {}).get("lib_sha16") != lib_sha:
            continue
        if workload_key(pmc.get("_build", {}).get("bench_args", "").split()) != workload_key(sys.argv[1:]):
            continue
        wk = [k for k in pmc if k.startswith("icm_walk") and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]]
        if not wk:
            continue
        # the walk that did the work (its idle twin -- the f32 walk when the filtered one ran -- moves no bytes): largest total fetch
        wk = [max(wk, key=lambda k: pmc[k]["FETCH_SIZE"]["mean_per_dispatch"] * pmc[k]["FETCH_SIZE"]["dispatches"])]
        # FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE x 2 = the guide's gfx950 correction for wide streaming reads
        tot = sum((2.0 * pmc[k]["FETCH_SIZE"]["mean_per_dispatch"] * pmc[k]["FETCH_SIZE"]["dispatches"]
                   + pmc[k]["WRITE_SIZE"]["mean_per_dispatch"] * pmc[k]["WRITE_SIZE"]["dispatches"]) * 1024.0 for k in wk)
        disp = sum(pmc[k]["FETCH_SIZE"]["dispatches"] for k in wk)
        best = {"bytes_per_icm_launch": tot / max(disp, 1), "icm_launches_profiled": disp, "source": "profiles/" + os.path.basename(f),
                "workload": pmc.get("_build", {}).get("bench_args", "")}
    return best


def train_codebooks(lsq, eng, dX, dB0, n, d, m, args):
    """Codebooks of the representative workload: this package's train_lsq_dev (8 iterations x 4 ILS, random initial codes) on the first 100 000 vectors.
    -> (K host (m*h, d), K device, objective per iteration, seconds)"""
    import tempfile
    import torch
    h = 256
    ns = min(n, 100_000)
    t0 = time.perf_counter()
    # profiling runs are several processes on one box: the codebooks are trained once and cached, so that the profiled processes contain no training launches
    # (opt-in: LSQ_BENCH_TRAIN_CACHE=1, set by tools/profile_round.sh; the key carries everything the codebooks depend on -- build, data seed / offset,
    #  shape, training parameters -- so a stale file of another build or data set can never be picked up silently: ADVICE r4)
    lib_path = lsq._lib.TUNING_LIB_PATH if args.tuning else lsq._lib.LIB_PATH
    key = "%s_%d_%d_%d_%d_%d_seed1234_goff%d_train8x4_seed42" % (lib_hash(lib_path), ns, d, m, args.icmiter, args.npert, int(getattr(args, "_goff", 0)))
    cache = os.path.join(tempfile.gettempdir(), "lsq_bench_trained_K_%s.npz" % key)
    use_cache = os.environ.get("LSQ_BENCH_TRAIN_CACHE") == "1"
    if use_cache and os.path.exists(cache):
        z = np.load(cache)
        if str(z["key"]) == key:
            return z["K"], torch.from_numpy(z["K"]).to(dX.device), z["obj"], -1.0      # -1: cache hit (reported as such)
    with lsq.Engine(eng.device) as e2:          # the training loop resident in HBM: the same codebooks as train_lsq(..., device_update=True) (tests/test_pipeline_gpu.py)
        dKt, _, _, _, obj = lsq.train_lsq_dev(dX[:ns].contiguous(), m, h, dB0[:ns].contiguous(), 8, 4, args.icmiter, True, args.npert, seed=42, engine=e2,
                                              norm_codebook=False)
    Ktr = np.ascontiguousarray(dKt.cpu().numpy())
    if use_cache:
        try:
            np.savez(cache, K=Ktr, obj=np.asarray(obj), key=key)
        except OSError:
            pass
    return Ktr, torch.from_numpy(Ktr).to(dX.device), obj, time.perf_counter() - t0


def workloads_leg(lsq, eng, dX, dB0, dK, n, d, m, args, goff):
    """The same n-vector encode under the conditions the headline number depends on.  Untimed w.r.t. `value`; every entry is
    1 warm-up + 3 timed steps with the per-class timings of exactly those steps.  -> dict of sub-objects."""
    import torch
    h = 256
    out = {}

    def timed(X, K, options, note, steps=3):
        saved = {"skip": args.skip, "fallback": 1, "schedule": args.schedule if args.schedule >= 0 else 6}
        for k, v in options.items():
            eng.set_option(k, v)
        buf = torch.empty((1, n, m), dtype=torch.uint8, device=X.device)
        run = lambda: eng.encode_icm_dev(X, dB0, K, m, [args.ils], args.icmiter, args.npert, True, seed=42, global_offset=goff, out=buf)
        run()
        torch.cuda.synchronize()
        eng.reset_timings()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, sums, stats = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tm = eng.timings()
        for k in options:
            eng.set_option(k, saved[k])
        total_nu = n * args.ils * args.icmiter * m * steps
        return {"value": n / dt, "unit": "vectors/s", "ms_per_step": dt * 1e3, "icm_ms": tm["icm_ms"] / steps, "unaries_ms": tm["unaries_ms"] / steps,
                "cost_ms": tm["cost_ms"] / steps, "recomputed_fraction": tm["icm_node_updates"] / max(total_nu, 1),
                "ambiguous_fraction": tm["filter_refined"] / max(tm["icm_node_updates"], 1),
                "filter_f32": int(tm["filter_f32"] // steps), "filter_f32_fraction": tm["filter_f32"] / max(tm["icm_node_updates"], 1),
                "blocks": {"staged_f32": int(tm["staged_blocks"] // steps), "light_f32": int(tm["light_blocks"] // steps),
                           "filtered_u16": int(tm["filtered_blocks"] // steps)},
                "objective": float(sums[0] / n), "last_ils_pct_better": float(100.0 * stats[-1, 1] / n), "options": options, "note": note}, buf

    # ---- floor: no memoisation at all (every one of the I x J x m node updates of every vector is recomputed), filtered walk and f32 walk
    out["floor"] = {}
    out["floor"]["filtered_walk"], _ = timed(dX, dK, {"skip": 0, "fallback": 0},
                                             "skip = 0, fallback = 0: what the step costs when the memoisation saves nothing (worst case of the data dependence)")
    out["floor"]["f32_walk"], _ = timed(dX, dK, {"skip": 0, "fallback": 0, "schedule": 4}, "the same through schedule 4: no memoisation and no 16-bit filter")
    out["f32_walk_with_memoisation"], _ = timed(dX, dK, {"schedule": 4}, "schedule 4 (f32 walk) with the default exact memoisation: what the 16-bit filter alone buys")

    # ---- trained codebooks: this package's own train_lsq (host LSQR codebook update + GPU encode) on the first 100 000 vectors
    ns = min(n, 100_000)
    Ktr, dKtr, obj, train_s = train_codebooks(lsq, eng, dX, dB0, n, d, m, args)
    out["trained"], btr = timed(dX, dKtr, {}, "codebooks = train_lsq(first %d vectors of the same data, random initial codes, 8 iterations x 4 ILS): "
                                "%s (not timed); default options" % (ns, "taken from this build's cache (LSQ_BENCH_TRAIN_CACHE=1)" if train_s < 0 else "trained in %.1f s" % train_s))
    out["trained"]["train_objective_first_last"] = [float(obj[0]), float(obj[-1])]
    # parity on the trained workload too: 256 vectors of its output vs the oracle
    import oracle as O
    O.build()
    a = n // 3
    ref, _ = O.encode_icm(dX[a:a + 256].cpu().numpy(), dB0[a:a + 256].cpu().numpy().astype(np.int16) + 1, Ktr, m, h, [args.ils], args.icmiter,
                          args.npert, True, 42, global_offset=goff + a)
    out["trained"]["sample_parity"] = bool(np.array_equal(ref[0], btr[0][a:a + 256].cpu().numpy().astype(np.int16) + 1))

    # ---- heavy tails: every vector scaled by a Cauchy variate (|x| spans orders of magnitude): the sampled level range misses outliers
    g = torch.Generator(device=dX.device)
    g.manual_seed(99)
    scale = torch.empty((n, 1), dtype=torch.float32, device=dX.device).cauchy_(generator=g)
    dXh = (dX * scale).contiguous()
    out["heavy_tailed"], bh = timed(dXh, dK, {}, "x_i <- x_i * Cauchy(0,1): heavy-tailed norms; unaries outside the sampled 16-bit level range take the exact f32 path "
                                    "(filter_f32); default options")
    ref, _ = O.encode_icm(dXh[a:a + 256].cpu().numpy(), dB0[a:a + 256].cpu().numpy().astype(np.int16) + 1, dK.cpu().numpy(), m, h, [args.ils],
                          args.icmiter, args.npert, True, 42, global_offset=goff + a)
    out["heavy_tailed"]["sample_parity"] = bool(np.array_equal(ref[0], bh[0][a:a + 256].cpu().numpy().astype(np.int16) + 1))
    del dXh, scale
    return out


def search_leg(lsq, eng, dK, dcodes, n, d, m, nq=10000, knn=1000):
    """The step after the path (BASELINE's other metric, recall@1, is computed from it): the ADC linear scan over the codes the timed
    loop produced, on the device (lsq_linscan_dev), with the host scan and -- where it travelled -- the reference's own build as CPU figures and
    checkers.  Synthetic queries from the data distribution; dbnorms = ||sum of codewords||^2 (f64 -> f32)."""
    import torch
    H = 256
    dQ = eng.synth_data_u8_dev(777, nq, d)
    recon = torch.zeros((n, d), dtype=torch.float32, device=dK.device)
    for j in range(m):
        recon += dK[j * H + dcodes[:, j].long()]
    dN = (recon.double() ** 2).sum(1).float().contiguous()
    del recon
    eng.linscan_dev(dcodes, dQ, dK, dN, m, knn)                 # warm-up at full size: the scan's work buffers (2 x 1.2 GB of candidate lists) are allocated here
    torch.cuda.synchronize()
    eng.reset_timings()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        dd, di = eng.linscan_dev(dcodes, dQ, dK, dN, m, knn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = eng.linscan_stats()
    lookups = float(n) * nq * m
    res = {"value": nq / dt, "unit": "queries/s", "ms_per_call": dt * 1e3, "queries": nq, "codes": n, "knn": knn,
           "breakdown_ms": {k: st[k] / reps for k in ("lut_ms", "sample_ms", "scan_ms", "select_ms")},
           "table_lookups_per_s": lookups / dt,
           "roofline": {"kernel": "adc_scan_kernel", "bound": "lds", "achieved": lookups * 16 / 4 * reps / (st["scan_ms"] * 1e-3) / 1e12 if st["scan_ms"] > 0 else None,
                        "peak": 150.0, "unit": "TB/s", "pattern_ceiling": 150.0 / 2.125, "measured_linear": 76.5, "measured_pattern": 58.4,
                        "note": "16-byte LDS reads of random 64-byte table rows (one code x four queries each) during the scan kernel alone.  peak = the guide's "
                                "ds_read_b128 rate (MI355X_MICROARCH.md, LDS: 256 B/clk/CU).  pattern_ceiling = that rate divided by the bank conflicts this access "
                                "pattern cannot avoid: a b128 read is served in four fixed 16-lane groups of 256 B; a group holds four codes = four RANDOM 64-byte rows, "
                                "each aligned to one quarter of the bank row, and costs max(rows per quarter) cycles: E = 544/256 = 2.125 for four uniform rows (no static "
                                "layout does better: any two rows share slots with probability >= 1/4).  SQ_LDS_BANK_CONFLICT / IDX_ACTIVE = 0.48 is that expectation.  "
                                "measured_linear / measured_pattern: what tools/ubench_lds (kraw: 16 independent ds_read_b128 per wait, ONE VALU op per read, 16 waves per CU) reaches "
                                "on this part for contiguous 1 KiB reads and for this pattern (profiles/r04_ubench_lds.txt) -- half the guide's figure, unexplained"},
           "candidates_per_query": st["candidates"] / max(st["queries"], 1), "fallback_queries": int(st["fallback_queries"]),
           "threshold_rank": int(st["threshold_rank"]), "list_capacity": int(st["list_capacity"]),
           "note": "lsq_linscan_dev on the codes of the timed encode, inputs resident in HBM; distances, ids and tie order identical to the reference's "
                   "linscan_aqd_query_extra_byte (checked below on a sample of the queries)"}
    if res["roofline"]["achieved"]:
        res["roofline"]["frac"] = res["roofline"]["achieved"] / res["roofline"]["peak"]
        res["roofline"]["frac_of_pattern_ceiling"] = res["roofline"]["achieved"] / res["roofline"]["pattern_ceiling"]
        res["roofline"]["frac_of_measured_pattern"] = res["roofline"]["achieved"] / res["roofline"]["measured_pattern"]
    # CPU figures + checks on a bounded sample of the queries
    nh = min(nq, 512)
    codes_h, Q_h, K_h, N_h = dcodes.cpu().numpy(), dQ[:nh].cpu().numpy(), dK.cpu().numpy(), dN.cpu().numpy()
    L = lsq._lib.load()
    hd = np.zeros((nh, knn), np.float32)
    hi = np.zeros((nh, knn), np.int32)
    t0 = time.perf_counter()
    lsq._lib.check(L.lsq_linscan_aqd_query_extra_byte(hd.ctypes.data, hi.ctypes.data, codes_h.ctypes.data, Q_h.ctypes.data, K_h.ctypes.data,
                                                      N_h.ctypes.data, nh, n, m, H, d, knn, 0))
    th = time.perf_counter() - t0
    res["host_scan"] = {"value": nh / th, "unit": "queries/s", "queries": nh, "threads": os.cpu_count(), "kind": "this library's host scan (lsq_linscan_aqd_query_extra_byte)",
                        "same_results": bool(np.array_equal(hi, di[:nh].cpu().numpy()) and np.array_equal(hd, dd[:nh].cpu().numpy()))}
    import oracle as O
    if O.ref_linscan_path() is not None:
        nr = min(nh, 256)
        t0 = time.perf_counter()
        rd, ri = O.ref_linscan(codes_h, Q_h[:nr], K_h, N_h, m, H, knn)
        tr = time.perf_counter() - t0
        res["reference_scan"] = {"value": nr / tr, "unit": "queries/s", "queries": nr, "kind": "reference (oracle/_ref: the reference's own linscan_aqd_pairwise_byte.cpp, OpenMP)",
                                 "same_results": bool(np.array_equal(ri, di[:nr].cpu().numpy()) and np.array_equal(rd, dd[:nr].cpu().numpy()))}
    return res


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run with N local ranks on a free
    loopback port.  -> the launcher's exit code (rank 0 prints the JSON line; the children inherit stdout / stderr)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    import torch
    lsq = importlib.import_module("local-search-quantization_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, the same command the driver's
        # torch.distributed.run form runs); rank 0's JSON line is this process' only stdout
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or plain `python bench.py --gpus N`)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if world > ndev and "LSQ_BENCH_BACKEND" not in os.environ:
        os.environ["LSQ_BENCH_BACKEND"] = "gloo"      # more ranks than devices (a 1-GPU smoke run): RCCL cannot put two ranks on one device
    dev_index = local_rank % ndev                      # one rank per GPU in production; modulo only for 1-GPU smoke runs
    torch.cuda.set_device(dev_index)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LSQ_BENCH_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; gloo only for 1-GPU smoke tests
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    d, m, h = args.d, args.m, 256
    if args.scaling == "strong":                       # cfg4: one dataset, `splitarray` shards (src/utils.jl:152-177)
        goff, stop = lsq.split_ranges(args.total, world)[rank]
        n = stop - goff
        n_total = args.total
    else:                                              # weak: every rank encodes its own n vectors of a global index space
        n, goff, n_total = args.n, rank * args.n, args.n * world
    eng = lsq.Engine(dev_index, profile=True, chunk=(args.chunk or None), tuning=args.tuning)
    if args.schedule >= 0:
        eng.set_option("schedule", args.schedule)
    eng.set_option("skip", args.skip)
    if args.tuning:
        eng.set_option("ablation", args.ablation)
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    lib_path = lsq._lib.TUNING_LIB_PATH if args.tuning else lsq._lib.LIB_PATH
    lib_sha = lib_hash(lib_path)

    data_tag = "synthetic"
    dX = None
    sift = os.path.join(os.environ.get("LSQ_DATA_DIR", ""), "sift", "sift_base.fvecs")
    if os.environ.get("LSQ_DATA_DIR") and os.path.exists(sift) and d == 128:
        # use the real SIFT1M base set when it is there (it is not in this image); rank r takes rows [goff, goff + n)
        rows_in_file = os.path.getsize(sift) // (4 + 4 * 128)
        if goff + n <= rows_in_file:
            X = lsq.fvecs_read((goff + 1, goff + n), sift)              # (d, n), 1-based inclusive bounds like the reference reader
            dX = torch.from_numpy(np.ascontiguousarray(X.T)).to("cuda:%d" % dev_index)
            data_tag = "sift1m_base rows %d..%d (%s)" % (goff, goff + n - 1, sift)
    if dX is None:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
        if d == 960:
            dX.mul_(0.3 / 255.0)                       # GIST-like range
    dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
    dK = eng.synth_codebooks_dev(4321, m, d) if rank == 0 else torch.zeros((m * h, d), dtype=torch.float32, device=dX.device)
    if d == 960 and rank == 0:
        dK.mul_(0.3 / 255.0)
    if args.trained_codebooks and world == 1:
        _, dK, _, _ = train_codebooks(lsq, eng, dX, dB0, n, d, m, args)
        data_tag += "; codebooks TRAINED by this package's train_lsq on the first 100 000 vectors (--trained-codebooks)"
    dBs = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
    torch.cuda.synchronize()

    def step(ils=args.ils):
        if dist is not None:
            dist.broadcast(dK, src=0)                  # RCCL over xGMI: the one data-path collective
        return eng.encode_icm_dev(dX, dB0, dK, m, [ils], args.icmiter, args.npert, True, seed=42,
                                  global_offset=goff, out=dBs)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    t_gpu0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    fence()
    eng.reset_timings()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, sums, stats = step()
    fence()
    dt_local = dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dX.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = eng.timings()

    # ---- per-rank roofline numbers (every rank computes its own; rank 0 reports all of them) -------------------------
    cs = 8 if m <= 8 else 16
    launches = max(tm["icm_launches"], 1)
    avg_launch_s = tm["icm_ms"] * 1e-3 / launches
    nu_per_launch = tm["icm_node_updates"] / launches
    # which walk did the work: the 16-bit filtered one (u16 unary row: 2h B) or the f32 one (4h B); + code record read + 1 code byte
    filtered = tm["filtered_blocks"] > tm["staged_blocks"]
    bytes_nu = (2 * h if filtered else 4 * h) + cs + 1
    hbm_bytes = nu_per_launch * bytes_nu              # M2 data-flow, per RECOMPUTED node update
    achieved = hbm_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    mine = {"rank": rank, "device": dev_index, "vectors": n, "global_offset": goff, "ms_per_step": dt_local / args.steps * 1e3,
            "vectors_per_s": n * args.steps / dt_local, "hbm_frac": achieved / HBM_PEAK_GBS, "icm_ms_per_step": tm["icm_ms"] / args.steps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "avg_launch_us": avg_launch_s * 1e6, "launches": int(tm["icm_launches"])}}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    if rank == 0:
        value = n_total * args.steps / dt
        table_bytes = nu_per_launch * (m - 1) * (2 if filtered else 4) * h      # (m-1) table rows per node update (u16 levels or f32): LDS reads (L2 gathers in light blocks)
        total_nu = n * args.ils * args.icmiter * m           # node updates one step resolves on this rank
        m1_bytes = n * (4 * d + 2 * m + 2 * m + 4 + 4 * m * h + 4 * m * h + 4 * d)      # SURVEY 8(d) model M1 per step
        step_s = dt / args.steps
        walk_ceiling = LDS_PEAK_GBS / walk_conflict_factor(m)      # the filtered walk's own pattern (schedule 6; the f32 walk keeps plain rows)
        roof = {
            "kernel": ("icm_walkq_kernel<%d,SLQ> (16-bit filtered walk, exact f32 refinement)" if filtered else "icm_walk_kernel<%d,SL> (f32 walk)") % m,
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_of_achievable_6300": achieved / HBM_ACHIEVABLE_GBS,
            "traffic": None,
            "avg_launch_us": avg_launch_s * 1e6, "launches": int(tm["icm_launches"]),
            "algorithmic_bytes_per_launch": hbm_bytes,
            "bytes_per_node_update": bytes_nu,
            "hbm_achieved": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
            "node_updates_recomputed_per_launch": nu_per_launch,
            "recomputed_fraction": tm["icm_node_updates"] / max(total_nu * args.steps, 1),
            "blocks": {"staged_f32": int(tm["staged_blocks"]), "light_f32": int(tm["light_blocks"]), "filtered_u16": int(tm["filtered_blocks"])},
            "filter": {"ambiguous_node_updates": int(tm["filter_refined"]), "ambiguous_fraction": tm["filter_refined"] / max(tm["icm_node_updates"], 1),
                       "exact_candidate_evaluations": int(tm["filter_exact"]), "out_of_sampled_range_node_updates": int(tm["filter_f32"]),
                       "note": "node updates whose two best 16-bit level sums lie within the rigorous window are re-decided in exact f32 (every candidate inside "
                               "the window); the others are decided by the level sums alone -- same codes as the f32 walk, bit for bit"},
            "data_flow": "M2 (SURVEY 8(d)): the unary row of a node is re-read from HBM at every recomputed node update -- as 16-bit levels (512 B) by "
                         "the filtered walk, as f32 (1 KiB) by the f32 walk; `achieved` counts only node updates that were actually recomputed (memoised "
                         "ones move no bytes).  It is a fraction of the traffic this design chose to create, not of the compulsory bytes -- see "
                         "m1_compulsory.  The filtered walk's slice loop was bound by VALU issue (removing every LDS read changed nothing, removing VALU work shortened it); after the instruction-count work of round 2 its measured traffic moves at ~5.1 TB/s of the ~6.3 TB/s a streaming kernel reaches on this part (traffic_rate below): the launch is HBM-bound on the bytes it creates, 1.5x the algorithmic ones (DESIGN 4.2).",
            "gather": {"achieved": table_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                       "frac": (table_bytes / avg_launch_s / 1e9 / LDS_PEAK_GBS) if avg_launch_s > 0 else 0.0,
                       "pattern_ceiling": walk_ceiling,
                       "frac_of_pattern_ceiling": (table_bytes / avg_launch_s / 1e9 / walk_ceiling) if avg_launch_s > 0 else 0.0,
                       "conflict_factor": walk_conflict_factor(m),
                       "note": "(m-1) pair-table rows per recomputed node update (512 B each as u16 levels, 1 KiB as f32), read from LDS-staged slices with "
                               "ds_read_b128; rotated placement (round 5): the vectors of a 16-lane group read different slots of their 256-byte lines -- conflict-free "
                               "for a full group of tables, ceil(slots / tables) cycles for the short second group; pattern_ceiling = 150 TB/s / conflict_factor "
                               "(the plain placement of earlier rounds: / 2.125)"},
            "m1_compulsory": {"bytes_per_step": m1_bytes, "achieved": m1_bytes / step_s / 1e9, "unit": "GB/s",
                              "frac": m1_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                              "note": "SURVEY 8(d) model M1 (X once, codes in/out, unaries written once and read once, X re-read for the cost) / whole step time"},
        }
        # which ceiling does this shape use more of?  HBM (the level stream) or the LDS gathers (against the ceiling of THEIR access pattern)
        util = {"hbm": roof["frac"], "lds": roof["gather"]["frac_of_pattern_ceiling"]}
        roof["utilisation"] = util
        if util["lds"] > util["hbm"]:
            roof.update({"bound": "lds", "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS},
                         "achieved": roof["gather"]["achieved"], "peak": walk_ceiling, "frac": util["lds"],
                         "bound_note": "m = %d: %d table rows per node update -- the LDS gathers run closer to their ceiling (150 TB/s / %.3f: the rotated placement's "
                                       "conflict factor) than the level stream to HBM's; achieved / peak / frac are the gather's, the HBM figures are under `hbm`"
                                       % (m, m - 1, walk_conflict_factor(m))})
        tr = pmc_traffic(lib_sha)
        if tr is not None:
            roof["traffic"] = tr["bytes_per_icm_launch"]
            roof["traffic_source"] = tr
            if avg_launch_s > 0:
                # the rate at which the kernel's MEASURED traffic moves (profiled launch bytes / live launch time): how close the launch as a
                # whole is to the memory system's limit, whatever fraction of those bytes was algorithmically necessary
                roof["traffic_rate"] = {"value": tr["bytes_per_icm_launch"] / avg_launch_s / 1e9, "unit": "GB/s",
                                        "frac_of_peak": tr["bytes_per_icm_launch"] / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                                        "frac_of_achievable_6300": tr["bytes_per_icm_launch"] / avg_launch_s / 1e9 / HBM_ACHIEVABLE_GBS}
        workload = ("BASELINE configs[1]" if (d, m, args.scaling, n) == (128, 8, "weak", 1_000_000) else
                    "BASELINE configs[2]" if (d, m, args.scaling, n) == (128, 16, "weak", 1_000_000) else
                    "BASELINE configs[3]" if (d, m, args.scaling) == (960, 8, "strong") else
                    "BASELINE configs[4]" if (d, m, args.scaling, n) == (128, 8, "weak", 12_500_000) else "custom")
        out = {
            "metric": "vectors encoded/sec (ICM, m=%d h=%d)" % (m, h),
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": data_tag,
            "dtype_note": "every code is the exact f32 argmin of the reference's arithmetic (bit-exact vs the oracle); the dominant kernel FILTERS on 16-bit "
                          "levels of the same sums (u16 planes written by the unary GEMM) and re-decides in exact f32 whatever the filter cannot prove",
            "config": {
                "workload": "%s: %s base encode, %s, m=%d, h=%d, %d ILS iters x %d ICM sweeps, npert=%d, randord, seed=42; inputs resident "
                            "in HBM; lsq_encode_icm_dev" % (workload, "SIFT1M-shaped" if d == 128 else "GIST1M-shaped" if d == 960 else "synthetic",
                                                            ("%d x %d f32 per GPU" % (n, d)) if args.scaling == "weak" else
                                                            ("%d x %d f32 in total, splitarray shards over %d GPUs" % (n_total, d, world)),
                                                            m, h, args.ils, args.icmiter, args.npert),
                "vectors_per_gpu": n if args.scaling == "weak" else [r["vectors"] for r in ranks], "vectors_total": n_total,
                "d": d, "m": m, "h": h, "ils_iters": args.ils, "icm_iters": args.icmiter, "npert": args.npert,
                "skip_unchanged_nodes": bool(args.skip),
                "parallelism": "%d x independent shards (one process per GPU), RCCL broadcast of codebooks, no collective in the sweep" % world,
                "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                "rccl_ranks": world if (world == 1 or backend == "nccl") else 0, "library": os.path.basename(lib_path), "lib_sha16": lib_sha,
            },
            "ranks": ranks,
            "objective": float(sums[0] / n), "last_ils_pct_better": float(100.0 * stats[-1, 1] / n),
            "roofline": roof,
            "time_breakdown_ms_per_step": {k: tm[k] / args.steps for k in ("tables_ms", "unaries_ms", "perturb_ms", "icm_ms", "cost_ms", "other_ms")},
        }
        if args.tuning and args.ablation:
            out["INVALID"] = "ablation %d: timing-only variant, results are garbage" % args.ablation

    # ---- extra legs: rank 0 of a single-GPU run only -----------------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_sample_parity:      # copies two 256-row slices only: also on the 12.5 M-vector cfg5 share
        ok, detail = sample_parity(eng, dX, dB0, dK, dBs, n, m, args, goff)
        out["sample_parity"] = ok
        out["sample_parity_detail"] = detail
    if rank == 0 and world == 1 and not args.no_extra_legs:
        # end to end through the host-buffer entry point (pageable numpy buffers): H2D of X / K / codes, encode, D2H of the codes
        if n * d * 4 <= 8 << 30:
            Xh, Kh = dX.cpu().numpy(), dK.cpu().numpy()
            Bh = dB0.cpu().numpy().astype(np.int16) + 1
            eng.encode_icm(Xh[:1000], Bh[:1000], Kh, m, [1], args.icmiter, args.npert, True, seed=42)       # staging buffers allocated
            best, e2e_tm, e2e_all = None, None, []
            for _ in range(3):
                eng.reset_timings()
                t0 = time.perf_counter()
                Bs_h, objs_h = eng.encode_icm(Xh, Bh, Kh, m, [args.ils], args.icmiter, args.npert, True, seed=42, global_offset=goff)
                e = time.perf_counter() - t0
                e2e_all.append(e * 1e3)
                if best is None or e < best:
                    best, e2e_tm = e, eng.timings()
            same = bool(np.array_equal(Bs_h[0].astype(np.int16) - 1, dBs[0].cpu().numpy().astype(np.int16)))      # dBs: output of the timed loop
            out["end_to_end"] = {"value": n / best, "unit": "vectors/s", "ms_per_call": best * 1e3,
                                 "ms_first_call": e2e_all[0], "ms_steady_state": min(e2e_all[1:]), "value_first_call": n / (e2e_all[0] * 1e-3),
                                 "note": "lsq_encode_icm on pageable host buffers: upload of X (%.0f MB), K and int16 codes, the whole encode, "
                                         "download of the int16 codes (X goes up panel by panel under its own unary GEMM: option upload_pipeline_min_bytes); `value` = the best of 3 "
                                         "calls (steady state); ms_first_call = the first full-size call of this context (first touch of the caller's pageable buffers, the workspace "
                                         "allocations) -- what a one-call process such as demos/demo_lsq_gpu.jl:50 sees, measured from a fresh process by tools/first_call.py "
                                         "(profiles/r06_first_call.txt); never reported as the top-level `value`" % (n * d * 4 / 1e6)}
            out["end_to_end"]["same_codes_as_device_path"] = same
            out["end_to_end"]["ms_all_calls"] = e2e_all
            out["end_to_end"]["device_ms_of_best_call"] = {k: e2e_tm[k] for k in ("tables_ms", "unaries_ms", "icm_ms", "cost_ms", "other_ms")}
            out["end_to_end"]["table_reuses"] = int(e2e_tm["table_reuses"])
            del Xh, Bh
        # the non-blocking form of the same call (option "async": no host synchronisation inside the call; both walks enqueued per ILS iteration, the idle one
        # returns at once): what a caller that pipelines work on the stream pays for it
        step_nb = lambda: eng.encode_icm_dev(dX, dB0, dK, m, [args.ils], args.icmiter, args.npert, True, seed=42, global_offset=goff, out=dBs, nonblocking=True)
        step_nb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            _, nb_sums, _ = step_nb()
        torch.cuda.synchronize()
        nb_dt = (time.perf_counter() - t0) / args.steps
        out["nonblocking_call"] = {"value": n / nb_dt, "unit": "vectors/s", "ms_per_step": nb_dt * 1e3, "same_objective": bool(abs(float(nb_sums[0].item()) - float(sums[0])) <= 1e-9 * abs(float(sums[0]))),
                                   "note": "lsq_encode_icm_dev with option async = 1: verdict and probe decided on the device, sums / counters written in stream order"}
        # BASELINE configs[0] on the GPU: the trainer's call -- n = 10 000, d = 128, m = 8, ONE ILS iteration per call (demo_lsq.jl:34),
        # chained through the host-buffer entry point lsq_encoding_icm (tables cached after the first call: same codebooks) and through lsq_encode_icm_dev
        if (d, m) == (128, 8):
            n1 = 10_000
            X1, B1 = dX[:n1].cpu().numpy(), dB0[:n1].cpu().numpy().astype(np.int16) + 1
            K1 = dK.cpu().numpy()
            with lsq.Engine(dev_index) as e1:
                Bc = B1
                for it in range(3):
                    Bc = e1.encoding_icm(X1, Bc, K1, m, args.icmiter, True, args.npert, seed=42, it=it)
                # (a) the tables of unchanged codebooks reused: 20 chained calls with the same K
                t0 = time.perf_counter()
                for it in range(3, 23):
                    Bc = e1.encoding_icm(X1, Bc, K1, m, args.icmiter, True, args.npert, seed=42, it=it)
                t_host = (time.perf_counter() - t0) / 20
                reuses = e1.timings()["table_reuses"]
                # (b) the TRAINER's ratio: the codebooks change after every `ilsiter` = 8 calls (LSQ.jl:53-66: update_codebooks, then ilsiter x encoding_icm),
                #     so every 8th call uploads K and rebuilds the tables; 24 calls = three outer iterations
                K2 = K1.copy()
                t0 = time.perf_counter()
                for it in range(24):
                    if it % 8 == 0:
                        K2[(it // 8) * 7 + 3, 5] += np.float32(0.25)           # a new codebook matrix, as after update_codebooks
                    Bc = e1.encoding_icm(X1, Bc, K2, m, args.icmiter, True, args.npert, seed=43, it=it)
                t_train = (time.perf_counter() - t0) / 24
                # (c) every call rebuilds (K changes every call)
                t0 = time.perf_counter()
                for it in range(8):
                    K2[it * 11 + 1, 7] += np.float32(0.25)
                    Bc = e1.encoding_icm(X1, Bc, K2, m, args.icmiter, True, args.npert, seed=44, it=it)
                t_rebuild = (time.perf_counter() - t0) / 8
                dX1, dB1 = dX[:n1].contiguous(), dB0[:n1].contiguous()
                o1 = torch.empty((1, n1, m), dtype=torch.uint8, device=dX.device)
                for _ in range(3):
                    e1.encode_icm_dev(dX1, dB1, dK, m, [1], args.icmiter, args.npert, True, seed=42, out=o1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    e1.encode_icm_dev(dX1, dB1, dK, m, [1], args.icmiter, args.npert, True, seed=42, out=o1)
                torch.cuda.synchronize()
                t_dev = (time.perf_counter() - t0) / 20
            out["cfg1_gpu"] = {"host_buffers": {"value": n1 / t_train, "ms_per_call": t_train * 1e3,
                                                "ms_per_call_tables_reused": t_host * 1e3, "ms_per_call_tables_rebuilt": t_rebuild * 1e3,
                                                "table_reuses_of_the_20_same_K_calls": int(reuses) - 2,
                                                "note": "`value` / ms_per_call: the trainer's ratio -- the codebooks change every 8th call (LSQ.jl:53-66), 24 calls; "
                                                        "tables_reused: 20 chained calls with unchanged codebooks; tables_rebuilt: the codebooks change every call"},
                               "device_buffers": {"value": n1 / t_dev, "ms_per_call": t_dev * 1e3,
                                                  "note": "lsq_encode_icm_dev rebuilds its tables in every call (it cannot know whether the caller's device K changed)"},
                               "unit": "vectors/s per encoding_icm call (1 ILS iteration, %d sweeps), blocking" % args.icmiter,
                               "note": "BASELINE configs[0] exactly (n = 10 000, d = 128, m = 8): the GPU counterpart of cpu_baseline.cfg1"}
        # north_star's own operating point: 4 ILS iterations
        ns_ils = 4
        step(ns_ils)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(ns_ils)
        torch.cuda.synchronize()
        ns_dt = (time.perf_counter() - t0) / args.steps
        out["north_star_point"] = {"ils_iters": ns_ils, "value": n / ns_dt, "unit": "vectors/s", "ms_per_step": ns_dt * 1e3,
                                   "note": "north_star quotes its >= 50x target at 4 ILS iterations; same workload otherwise"}
        if not args.no_workloads and n * d * 4 <= 2 << 30:
            out["workloads"] = workloads_leg(lsq, eng, dX, dB0, dK, n, d, m, args, goff)
            # The REPRESENTATIVE number beside `value`: the reference encodes a base set with TRAINED codebooks (LSQ.jl:10-88 ->
            # demo_lsq_gpu.jl:33-50); `value` keeps the synthetic codebooks of SURVEY 8(d).  Same vectors, same options, same timing method.
            tr = out["workloads"]["trained"]
            out["trained"] = {"value": tr["value"], "unit": tr["unit"], "ms_per_step": tr["ms_per_step"], "recomputed_fraction": tr["recomputed_fraction"],
                              "ambiguous_fraction": tr["ambiguous_fraction"], "sample_parity": tr["sample_parity"], "vs_value": tr["value"] / out["value"],
                              "note": "codebooks trained by this package's train_lsq on the first 100 000 of the same vectors; details in workloads.trained"}
            step()                                                       # dBs again holds the timed workload's codes
            torch.cuda.synchronize()
            out["search"] = search_leg(lsq, eng, dK, dBs[0].contiguous(), n, d, m)
        if args.multi_leg:
            devs = list(range(ndev)) if ndev > 1 else [0, 0]
            Xh, Kh = dX.cpu().numpy(), dK.cpu().numpy()
            Bh = dB0.cpu().numpy().astype(np.int16) + 1
            with lsq.MultiEngine(devs) as mg:
                mg.encode_icm(Xh[:4000], Bh[:4000], Kh, m, [1], args.icmiter, args.npert, True, seed=42)
                t0 = time.perf_counter()
                mg.encode_icm(Xh, Bh, Kh, m, [args.ils], args.icmiter, args.npert, True, seed=42, global_offset=goff)
                e = time.perf_counter() - t0
            out["lsq_multi"] = {"devices": devs, "value": n / e, "unit": "vectors/s", "ms_per_call": e * 1e3,
                                "note": "single-process multi-GPU entry point (one context + host thread per listed device, splitarray shards, host buffers)"}
            del Xh, Bh
        # keep the device busy long enough for an external utilisation sampler; doubles as a steady-state check of `value`
        extra, t1 = 0, time.perf_counter()
        while time.perf_counter() - t_gpu0 < args.min_gpu_seconds and extra < 400:
            step()
            extra += 1
        torch.cuda.synchronize()
        if extra:
            out["steady_state"] = {"extra_untimed_steps": extra, "ms_per_step": (time.perf_counter() - t1) / extra * 1e3}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
            if "north_star_point" in out:
                ns = out["north_star_point"]
                ns["cpu_baseline"] = {"value": cb["value"] * args.ils / ns["ils_iters"], "unit": "vectors/s", "cores": cb["cores"], "kind": "port",
                                      "sample": "same measurement as cpu_baseline (%.3f s per ILS iteration of the 100 000-vector sample), "
                                                "scaled to %d iterations" % (cb["seconds_per_ils_iteration"], ns["ils_iters"])}
                ns["speedup_vs_cpu_baseline"] = ns["value"] / ns["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
