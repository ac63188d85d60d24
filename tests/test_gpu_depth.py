"""The STATED depth under pytest: BASELINE configs[1] at its full size AND its full depth (16 ILS iterations x 4 sweeps),
with 512 + 64 rows of the output re-encoded by the oracle -- on the synthetic codebooks of SURVEY 8(d) and on codebooks trained by this package
(the representative workload: the reference encodes with trained codebooks, LSQ.jl:10-88 -> demo_lsq_gpu.jl:33-50); a fixed-seed slice of the
two randomised campaigns (tools/fuzz_filter.py, tools/fuzz_scan.py); and `python bench.py --gpus 2` in its plain form (the script launches its own ranks).

Valid as a parity check because results depend on (vector, global index) only: the oracle run on rows [a, b) with global_offset = a
must reproduce rows [a, b) of the full-size call bit for bit.  Parity stays "unpinned": the oracle is this repo's restatement."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = 256
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(n):
    rng = np.random.default_rng(5)
    blocks = [(0, 256), (n // 2 - 128, n // 2 + 128)]                                            # the two blocks bench.py's sample_parity uses
    singles = [(int(i), int(i) + 1) for i in np.sort(rng.choice(n, size=64, replace=False))]    # + 64 scattered rows
    return blocks + singles


@pytest.mark.parametrize("codebooks", ["synthetic", "trained"])
def test_cfg2_full_size_full_depth_rows_vs_oracle(lsq, oracle, codebooks):
    import torch
    n, d, m, ils, J, npert, seed = 1_000_000, 128, 8, [16], 4, 4, 42
    with lsq.Engine(0) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d)
        dB0 = eng.randinit_dev(7, n, m)
        if codebooks == "synthetic":
            dK = eng.synth_codebooks_dev(4321, m, d)
        else:
            ns = 100_000
            with lsq.Engine(0) as e2:
                dK, _, _, _, obj = lsq.train_lsq_dev(dX[:ns].contiguous(), m, H, dB0[:ns].contiguous(), 8, 4, J, True, npert, seed=42, engine=e2,
                                                     norm_codebook=False)
            assert obj[-1] < obj[0]
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed)
        torch.cuda.synchronize()
        tm = eng.timings()
        assert tm["filtered_blocks"] > 0, tm                                   # the default road at this size: the 16-bit filtered walk
        rows = _rows(n)
        idx = np.concatenate([np.arange(a, b) for a, b in rows])
        t_idx = torch.from_numpy(idx).to(dX.device)
        # only the checked rows travel to the host
        Xr, Br, got_r = dX[t_idx].cpu().numpy(), dB0[t_idx].cpu().numpy().astype(np.int16) + 1, dBs[0][t_idx].cpu().numpy().astype(np.int16) + 1
        K = dK.cpu().numpy()
    # re-index the gathered rows as contiguous slices
    bad, pos = 0, 0
    for a, b in rows:
        w = b - a
        ref, _ = oracle.encode_icm(Xr[pos:pos + w], Br[pos:pos + w], K, m, H, ils, J, npert, True, seed, global_offset=a)
        bad += int((ref[0] != got_r[pos:pos + w]).any(axis=1).sum())
        pos += w
    assert bad == 0, "%d of %d checked rows differ from the oracle (%s codebooks)" % (bad, len(idx), codebooks)
    assert stats.shape == (16, 2) and stats[0, 1] > 0.9 * n and np.all(np.diff(stats[1:, 1]) <= 0.02 * n)      # "% better" falls off (encode_icm_cuda.jl:199-204)
    assert np.isfinite(sums[0]) and sums[0] > 0


def _check_rows(oracle, dX, dB0, dBs0, dK, rows, m, ils, J, npert, seed, goff0=0):
    """Re-encode the listed row ranges with the oracle (global index = goff0 + row) and count the rows that differ."""
    import torch
    idx = np.concatenate([np.arange(a, b) for a, b in rows])
    t_idx = torch.from_numpy(idx).to(dX.device)
    Xr, Br, got_r = dX[t_idx].cpu().numpy(), dB0[t_idx].cpu().numpy().astype(np.int16) + 1, dBs0[t_idx].cpu().numpy().astype(np.int16) + 1
    K = dK.cpu().numpy()
    bad, pos = 0, 0
    for a, b in rows:
        w = b - a
        ref, _ = oracle.encode_icm(Xr[pos:pos + w], Br[pos:pos + w], K, m, H, ils, J, npert, True, seed, global_offset=goff0 + a)
        bad += int((ref[0] != got_r[pos:pos + w]).any(axis=1).sum())
        pos += w
    return bad, len(idx)


def _rows_small(n, width=48, scattered=32, extra=()):
    rng = np.random.default_rng(6)
    blocks = [(0, width), (n // 2 - width // 2, n // 2 + width // 2), (n - width, n)] + list(extra)
    singles = [(int(i), int(i) + 1) for i in np.sort(rng.choice(n, size=scattered, replace=False))]
    return blocks + singles


@pytest.mark.parametrize("codebooks", ["synthetic", "trained"])
def test_cfg3_full_size_full_depth_rows_vs_oracle(lsq, oracle, codebooks):
    """BASELINE configs[2] (10^6 x 128, m = 16) at its stated depth (16 ILS iterations x 4 sweeps): the only place the m = 16 memoisation state (validity
    bits, fall-back to the current tuple, 16-byte records) is aged over 16 iterations.  176 rows re-encoded by the oracle."""
    import torch
    n, d, m, ils, J, npert, seed = 1_000_000, 128, 16, [16], 4, 4, 42
    with lsq.Engine(0) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d)
        dB0 = eng.randinit_dev(7, n, m)
        if codebooks == "synthetic":
            dK = eng.synth_codebooks_dev(4321, m, d)
        else:
            ns = 50_000
            with lsq.Engine(0) as e2:
                dK, _, _, _, obj = lsq.train_lsq_dev(dX[:ns].contiguous(), m, H, dB0[:ns].contiguous(), 4, 4, J, True, npert, seed=42, engine=e2,
                                                     norm_codebook=False)
            assert obj[-1] < obj[0]
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed)
        torch.cuda.synchronize()
        tm = eng.timings()
        assert tm["filtered_blocks"] > 0, tm
        bad, cnt = _check_rows(oracle, dX, dB0, dBs[0], dK, _rows_small(n), m, ils, J, npert, seed)
    assert cnt >= 96 and bad == 0, "%d of %d checked rows differ from the oracle (m = 16, %s codebooks)" % (bad, cnt, codebooks)
    assert stats.shape == (16, 2) and np.isfinite(sums[0]) and sums[0] > 0


def test_cfg4_share_full_depth_rows_vs_oracle(lsq, oracle):
    """BASELINE configs[3]: rank 3's splitarray shard of the 10^6 x 960 GIST-shaped set (125 000 vectors at global offset 375 000), m = 8, at the stated
    depth of 16 ILS iterations: every block holds 489 vectors, so every node update runs in the sparse regime.  176 rows re-encoded by the oracle at
    their GLOBAL indices."""
    import torch
    ntot, d, m, ils, J, npert, seed = 1_000_000, 960, 8, [16], 4, 4, 42
    start, stop = lsq.distributed.shard_range(ntot, 8, 3)
    n = stop - start
    assert (start, n) == (375_000, 125_000)
    with lsq.Engine(0) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=start)
        dX.mul_(0.3 / 255.0)                                                  # GIST-like range, as bench.py
        dB0 = eng.randinit_dev(7, n, m, global_offset=start)
        dK = eng.synth_codebooks_dev(4321, m, d)
        dK.mul_(0.3 / 255.0)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=start)
        torch.cuda.synchronize()
        tm = eng.timings()
        assert tm["filtered_blocks"] > 0, tm
        bad, cnt = _check_rows(oracle, dX, dB0, dBs[0], dK, _rows_small(n), m, ils, J, npert, seed, goff0=start)
    assert cnt >= 96 and bad == 0, "%d of %d checked rows differ from the oracle (125 000 x 960 at offset 375 000)" % (bad, cnt)
    assert stats.shape == (16, 2) and np.isfinite(sums[0]) and sums[0] > 0


def test_cfg5_chunk_boundary_full_depth_rows_vs_oracle(lsq, oracle):
    """BASELINE configs[4] walks 13 resident chunks per GPU: one chunk boundary at the stated depth -- 1 115 808 vectors = one full default chunk
    (256 x 3968) + 100 000, at a non-zero global offset; rows on both sides of the boundary, in the short second chunk and scattered."""
    import torch
    d, m, ils, J, npert, seed, goff = 128, 8, [16], 4, 4, 42, 25_000_000
    chunk = 256 * 3968
    n = chunk + 100_000
    with lsq.Engine(0) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
        dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
        dK = eng.synth_codebooks_dev(4321, m, d)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        torch.cuda.synchronize()
        rows = _rows_small(n, extra=[(chunk - 24, chunk + 24)])
        bad, cnt = _check_rows(oracle, dX, dB0, dBs[0], dK, rows, m, ils, J, npert, seed, goff0=goff)
    assert cnt >= 96 and bad == 0, "%d of %d checked rows differ from the oracle (two chunks, offset %d)" % (bad, cnt, goff)
    assert stats.shape == (16, 2) and np.isfinite(sums[0]) and sums[0] > 0


def test_fuzz_filter_fixed_seed_slice():
    """50 cases of the randomised campaign for the filtered walk (random shapes / scales / offsets / duplicated codewords / heavy tails, every block
    staged, every vector compared with the oracle), fixed seed."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_filter
    bad, summary = fuzz_filter.run(50, 20260929, verbose=False)
    assert bad == 0, summary


def test_fuzz_scan_fixed_seed_slice():
    """50 cases of the randomised campaign for the device ADC scan against the host scan (itself pinned to the reference's build), fixed seed."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_scan
    bad, summary = fuzz_scan.run(50, 20260929, verbose=False)
    assert bad == 0, summary


def test_bench_plain_multi_gpu_form_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and NO WORLD_SIZE: the script starts two ranks itself (on a 1-GPU box they share the device and
    the collective backend falls back to gloo), rank 0 prints ONE JSON line with both ranks' figures.  The command the driver would run on 8 GPUs
    is the same with --gpus 8."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--vectors", "70000", "--ils", "2",
           "--no-cpu-baseline", "--no-extra-legs", "--no-sample-parity"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["ranks"]) == 2 and out["config"]["vectors_total"] == 140000
    assert {r["rank"] for r in out["ranks"]} == {0, 1} and all("roofline" in r for r in out["ranks"])
    assert out["value"] > 0 and out["scaling"] == "weak"
    import torch
    if torch.cuda.device_count() >= 2:
        assert out["config"]["rccl_ranks"] == 2 and out["config"]["collective_backend"] == "rccl"
    else:
        assert out["config"]["collective_backend"] == "gloo"


def test_bench_strong_scaling_two_rank_form():
    """The strong-scaling form of BASELINE configs[3] on two ranks: `python bench.py --gpus 2 --scaling strong --total 250000 --dim 960` -- a FIXED total
    split into splitarray shards of 125 000 (the per-GPU share of the 8-GPU run), the script starting its own ranks; rank 0 prints ONE line with
    "scaling": "strong" and the whole job's vectors (so that the first 8-GPU run records a curve, not an rc)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--total", "250000", "--dim", "960", "--steps", "1",
           "--warmup", "1", "--ils", "2", "--no-cpu-baseline", "--no-extra-legs", "--no-sample-parity"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["vectors_total"] == 250000 and len(out["ranks"]) == 2
    assert out["value"] > 0 and out["config"]["d"] == 960 and sorted(out["config"]["vectors_per_gpu"]) == [125000, 125000]


def test_chained_host_calls_reuse_the_tables_of_unchanged_codebooks(lsq, oracle):
    """The trainer's chain (demos/demo_lsq.jl:48-51: encoding_icm called ilsiter times with the SAME codebooks): from the second call on the context
    finds the caller's K unchanged (one memcmp) and neither uploads it nor rebuilds the tables; a K that differs in ONE float rebuilds them.  Codes
    equal the oracle's chain either way (cfg1's shape)."""
    from conftest import make_problem
    d, n, m, J, npert, seed = 128, 10_000, 8, 4, 4, 42
    X, K, B0 = make_problem(d, n, m, seed=3)
    with lsq.Engine(0) as eng:
        B, Bref = B0, B0
        for it in range(3):
            B = eng.encoding_icm(X, B, K, m, J, True, npert, seed=seed, it=it)
            Bref = oracle.encoding_icm_faithful(X, Bref, K, m, H, J, True, npert, seed, it)
            assert np.array_equal(B, Bref), "call %d" % it
        assert eng.timings()["table_reuses"] == 2
        K2 = K.copy()
        K2[5 * H + 17, 3] += np.float32(0.5)
        B = eng.encoding_icm(X, B, K2, m, J, True, npert, seed=seed, it=3)
        Bref = oracle.encoding_icm_faithful(X, Bref, K2, m, H, J, True, npert, seed, 3)
        assert np.array_equal(B, Bref)
        assert eng.timings()["table_reuses"] == 2                       # rebuilt, not reused
        # another entry point in between (it overwrites the staged codebooks): the next chained call must rebuild as well
        eng.veccost(X[:100], B[:100], K, m)
        B = eng.encoding_icm(X, B, K2, m, J, True, npert, seed=seed, it=4)
        Bref = oracle.encoding_icm_faithful(X, Bref, K2, m, H, J, True, npert, seed, 4)
        assert np.array_equal(B, Bref)
        assert eng.timings()["table_reuses"] == 2


@pytest.mark.parametrize("n,d,m,forced", [(70_001, 32, 8, True), (70_001, 32, 8, False), (9_000, 24, 5, False), (66_000, 16, 12, True)])
def test_host_upload_pipeline_gives_the_same_codes(lsq, oracle, n, d, m, forced):
    """The host-buffer entry point's first chunk uploaded panel by panel under its own unary GEMM (level sample first; a helper thread feeds the copy
    stream): every vector against the oracle, with the filtered walk forced onto the chunk (`forced`) and on the default roads, several ragged panels,
    heavy-tailed rows in the LAST panel (their |sigma| lies beyond what the early sample predicted: they must be flagged, not mis-filtered)."""
    from conftest import make_problem
    ils, J, npert, seed = [2], 3, 3, 5
    X, K, B0 = make_problem(d, n, m, seed=9, kind="gauss")
    X[-300:] *= np.float32(50.0)                                                   # not in the sample's first panels
    ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed)
    with lsq.Engine(0) as eng:
        eng.set_option("upload_pipeline_min_bytes", 1)
        eng.set_option("upload_panel_bytes", 4 * d * 128 * 37)                    # 37 tiles per panel: several panels, the last one ragged
        if forced:
            for k, v in (("q16_min", 0), ("light", 0), ("filter_probe_div", 0), ("filter_fallback_div", 0)):
                eng.set_option(k, v)
        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
        tm = eng.timings()
    assert np.array_equal(Bs, ref), "%d codes differ" % (Bs != ref).sum()
    assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
    if forced:
        assert tm["filtered_blocks"] > 0 and tm["filter_f32"] > 0, tm             # the scaled rows went through the f32 routine


@pytest.mark.parametrize("m", list(range(1, 17)))
def test_every_codebook_count_through_the_filtered_walk(lsq, oracle, m):
    """Every instantiation of the filtered walk kernel, one by one: m = 1 .. 8 use the rotated-rows slice table (lsq_q16.h, WalkqRot: one or two table groups, with
    and without the free slot that holds the smallest keys), m = 9 .. 16 the plain placement.  The filter is forced onto every block (no light blocks, no probe, no
    fallback); several blocks hold a ragged number of vectors; every vector against the oracle."""
    from conftest import make_problem
    d, n, ils, J, npert, seed = 24, 21_013, [2], 3, min(m, 3), 11
    X, K, B0 = make_problem(d, n, m, seed=100 + m, kind="gauss")
    ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed)
    with lsq.Engine(0) as eng:
        for k, v in (("q16_min", 0), ("light", 0), ("filter_probe_div", 0), ("filter_fallback_div", 0)):
            eng.set_option(k, v)
        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
        tm = eng.timings()
    assert np.array_equal(Bs, ref), "%d codes differ at m = %d" % ((Bs != ref).sum(), m)
    assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
    if m > 1:
        assert tm["filtered_blocks"] > 0 and tm["light_blocks"] == 0, tm


def test_more_ils_iterations_than_the_per_call_block_holds(lsq, oracle):
    """The per-call words (counters, sums, flags) live in one 12 KB block whose counter window holds 512 ILS iterations; a longer call gives the counters an
    allocation of their own and goes back to separate fills / copies.  600 iterations on a small problem: codes, objective and the per-iteration counters against
    the oracle, then a short call on the same context (the window must not be used half-way)."""
    from conftest import make_problem
    d, n, m, J, npert, seed = 16, 300, 4, 2, 2, 9
    X, K, B0 = make_problem(d, n, m, seed=21, kind="gauss")
    with lsq.Engine(0) as eng:
        for ils in ([600], [3]):
            ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed)
            Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
            assert np.array_equal(Bs, ref), "ils = %s: %d codes differ" % (ils, (Bs != ref).sum())
            assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
