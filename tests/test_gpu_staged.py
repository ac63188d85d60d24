"""GPU parity of the LDS-STAGED node-update path of the walk kernel, for every codebook count and every kernel instantiation
family, compared with the oracle on ALL vectors (the staged path used to be oracle-tested for m = 8 only,
because a block with <= 256 active vectors -- any n <= 65 536 -- takes the L2-gather path instead).

Two ways to make blocks stage:  option "light" = 0 (always stage, any n)  and  natural staging (n large enough that every
block holds more than 256 active vectors).  Each test asserts WHICH path ran through the device counters exported in
`lsq_timings` (staged_blocks / light_blocks / filtered_blocks).

Instantiation families (csrc/lsq_icm.hip, lsq_launch_icm_walk): m = 1..8 -> <M,16,DEPTH 3,1024 threads>;
m = 9..13 -> <M,8,DEPTH 2,1024>; m = 14..16 -> <M,8,DEPTH 4,512>.  The reference demos use m = 7 (demo_lsq_gpu.jl:15);
BASELINE cfg3 is m = 16.  Conditioning / argmin semantics under test: encode_icm.jl:72-125.
"""
import numpy as np
import pytest

from conftest import make_problem

pytestmark = pytest.mark.gpu
H = 256


def _paths(eng):
    t = eng.timings()
    return t["staged_blocks"], t["light_blocks"], t["filtered_blocks"]


@pytest.mark.parametrize("m", list(range(1, 17)))
def test_forced_staging_every_m(lsq, oracle, m):
    """light = 0: every block stages its table slices through LDS, whatever its active count.  n = 3000 over 256 CUs is
    ~12 vectors per block (ragged), J = 3 sweeps, random node order; the four skip x fallback combinations must all
    give the oracle's codes on every vector."""
    d, n, ils, J, npert, seed = 16, 3000 + 7 * m, [1, 2], 3, min(4, m), 500 + m
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    Bs_ref, objs_ref, st_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed, want_stats=True)
    for schedule in (6, 4, 3):
        for skip in (1, 0):
            for fb in (1, 0):
                if schedule == 3 and (skip, fb) != (1, 1):
                    continue
                with lsq.Engine(0, schedule=schedule, skip=skip) as eng:
                    eng.set_option("light", 0)
                    eng.set_option("fallback", fb)
                    eng.set_option("q16_min", 0)             # schedule 6: the 16-bit filtered walk even at this size
                    eng.set_option("filter_probe_div", 0)    # ... for every ILS iteration, whatever the first one's ambiguity
                    Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
                    staged, light, filt = _paths(eng)
                tag = "m=%d schedule=%d skip=%d fallback=%d" % (m, schedule, skip, fb)
                assert np.array_equal(Bs, Bs_ref), "%s: %d of %d codes differ" % (tag, (Bs != Bs_ref).sum(), Bs.size)
                assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0), tag
                assert light == 0, "%s: staged=%d light=%d filtered=%d" % (tag, staged, light, filt)
                assert (filt > 0 and staged == 0) if schedule == 6 else (staged > 0 and filt == 0), "%s: staged=%d filtered=%d" % (tag, staged, filt)


@pytest.mark.parametrize("m,n", [(7, 90_000), (12, 80_000), (16, 80_000), (3, 120_000)])
def test_natural_staging_per_family(lsq, oracle, m, n):
    """n >= 80 000: at least 313 vectors per block, above the light-block threshold, so the first sweeps stage naturally
    (late sweeps of a block may still go light: both paths then serve the same vectors within one call).  One case per
    instantiation family + the demos' m = 7."""
    d, ils, J, npert, seed = 16, [2], 2 if m >= 12 else 3, min(4, m), 900 + m
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed)
    for schedule in (6, 4):                        # 6 = the default: 16-bit filtered walk (n >= 65 536); 4 = the f32 walk
        with lsq.Engine(0, schedule=schedule) as eng:
            Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
            staged, light, filt = _paths(eng)
            t = eng.timings()
        assert np.array_equal(Bs, Bs_ref), "schedule %d: %d of %d codes differ" % (schedule, (Bs != Bs_ref).sum(), Bs.size)
        assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
        assert (filt if schedule == 6 else staged) > 0, "schedule %d: staged=%d light=%d filtered=%d" % (schedule, staged, light, filt)
        if schedule == 6:
            assert t["filter_refined"] < 0.25 * t["icm_node_updates"], t      # the filter decides most node updates on 16 bits


def test_cfg1_exact_shape_chained_calls(lsq, oracle):
    """BASELINE configs[0] at its exact shape: n = 10 000, d = 128, m = 8, h = 256, 4 ICM sweeps, npert 4, random order --
    what `train_lsq` runs (demos/demo_lsq.jl:34): `ilsiter` = 8 chained calls of encoding_icm, each with the accept rule
    (encode_icm.jl:131-189).  Every call's output is compared with the oracle's reference-loop-nest restatement on all vectors."""
    d, n, m, J, npert, seed = 128, 10_000, 8, 4, 4, 2024
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="sift")
    Bg, Bo = B0.copy(), B0.copy()
    with lsq.Engine(0) as eng:
        for it in range(8):
            Bg = eng.encoding_icm(X, Bg, K, m, J, True, npert, seed=seed, it=it)
            Bo = oracle.encoding_icm_faithful(X, Bo, K, m, H, J, True, npert, seed, it, nworkers=oracle.num_threads())
            assert np.array_equal(Bg, Bo), "ILS call %d: %d codes differ" % (it, (Bg != Bo).sum())
        q = eng.qerror(X, Bg, K, m)
    assert abs(q - oracle.qerror(X, Bo, K, m, H)) <= 1e-5 * q


def _sample_rows_vs_oracle(oracle, X, K, B0, got, m, ils, J, npert, seed, rows, goff=0):
    for i in rows:
        ref, _ = oracle.encode_icm(X[i:i + 1], B0[i:i + 1], K, m, H, ils, J, npert, True, seed, global_offset=goff + int(i))
        assert np.array_equal(ref[-1, 0], got[i]), "vector %d differs: %s vs %s" % (i, ref[-1, 0], got[i])


def test_cfg3_m16_staged_sample_and_full_size_properties(lsq, oracle):
    """BASELINE configs[2] (m = 16, d = 128): (a) 300 000 vectors -- 1172 per block, every first-sweep block stages with the
    <16,8,4,512> instantiation -- a random sample of vectors must equal the oracle run on exactly those vectors (results
    depend on the global index only, P8); (b) the full 10^6 vectors through the size-independent properties (monotone cost,
    strict accept, objective = mean cost, invariance to chunking / sharding)."""
    import torch
    d, m, ils, J, npert, seed = 128, 16, [1, 2], 4, 4, 16
    with lsq.Engine(0) as eng:
        n = 300_000
        dX = eng.synth_data_u8_dev(31, n, d)
        dB0 = eng.randinit_dev(32, n, m)
        dK = eng.synth_codebooks_dev(33, m, d)
        dBs, sums, _ = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed)
        staged, light, team = _paths(eng)
        assert staged + team > 0
        X, K = dX.cpu().numpy(), dK.cpu().numpy()
        B0 = dB0.cpu().numpy().astype(np.int16) + 1
        got = dBs[1].cpu().numpy().astype(np.int16) + 1
        rng = np.random.default_rng(3)
        rows = np.sort(np.concatenate([rng.choice(n, size=40, replace=False), [0, n - 1]]))
        _sample_rows_vs_oracle(oracle, X, K, B0, got, m, ils, J, npert, seed, rows)
        del dX, dB0, dBs
        # (b) full size
        n = 1_000_000
        dX = eng.synth_data_u8_dev(31, n, d)
        dB0 = eng.randinit_dev(32, n, m)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed)
        torch.cuda.synchronize()
        X = dX.cpu().numpy()
        B0 = dB0.cpu().numpy().astype(np.int16) + 1
        B1 = dBs[0].cpu().numpy().astype(np.int16) + 1
        B2 = dBs[1].cpu().numpy().astype(np.int16) + 1
        c0, c1, c2 = (eng.veccost(X, B, K, m) for B in (B0, B1, B2))
        assert np.all(c1 <= c0) and np.all(c2 <= c1)
        assert np.array_equal(B1[c1 == c0], B0[c1 == c0]) and np.array_equal(B2[c2 == c1], B1[c2 == c1])
        assert abs(sums[1] / n - c2.astype(np.float64).mean()) <= 1e-6 * sums[1] / n
        assert stats[0, 1] == int((c1 < c0).sum()) and stats[1, 1] == int((c2 < c1).sum())
        rows = np.sort(rng.choice(n, size=16, replace=False))
        _sample_rows_vs_oracle(oracle, X, K, B0, B2, m, ils, J, npert, seed, rows)
        h1 = 333_337
        with lsq.Engine(0, chunk=250_000) as e2:
            a, sa, _ = e2.encode_icm_dev(dX[:h1].contiguous(), dB0[:h1].contiguous(), dK, m, ils, J, npert, True, seed=seed, global_offset=0)
            b, sb, _ = e2.encode_icm_dev(dX[h1:].contiguous(), dB0[h1:].contiguous(), dK, m, ils, J, npert, True, seed=seed, global_offset=h1)
        assert torch.equal(torch.cat([a, b], dim=1), dBs)
        assert np.allclose(sa + sb, sums, rtol=1e-9)


def test_cfg5_chunk_walk(lsq, oracle):
    """BASELINE configs[4] per-GPU mechanics at reduced length: 3 x 10^6 + 17 vectors generated ON THE DEVICE from the global
    Philox stream (rank offset 12.5 M, as rank 1 of the 8-GPU job would use), walked in resident chunks of 2^20 (the default).
    Checks: sampled rows around every chunk boundary and at random == the oracle on exactly those vectors; objective = mean cost;
    identical codes with another chunk size (P8)."""
    import torch
    d, m, ils, J, npert, seed = 128, 8, [2], 4, 4, 5
    n, goff = 3_000_017, 12_500_000
    with lsq.Engine(0) as eng, lsq.Engine(0, chunk=700_001) as e2:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
        dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
        dK = eng.synth_codebooks_dev(4321, m, d)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        staged, light, team = _paths(eng)
        assert staged + team > 0
        dBs2, sums2, stats2 = e2.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        assert torch.equal(dBs, dBs2) and np.allclose(sums, sums2, rtol=1e-9) and np.array_equal(stats, stats2)
        K = dK.cpu().numpy()
        rng = np.random.default_rng(9)
        rows = [0, n - 1] + [c * (256 * 3968) + o for c in (1, 2) for o in (-1, 0, 1)] + list(rng.choice(n, size=20, replace=False))
        rows = np.array(sorted(set(int(r) for r in rows)))
        idx = torch.from_numpy(rows).to(dX.device)
        Xs = dX[idx].cpu().numpy()
        B0s = dB0[idx].cpu().numpy().astype(np.int16) + 1
        gots = dBs[0][idx].cpu().numpy().astype(np.int16) + 1
        for q, i in enumerate(rows):
            assert np.array_equal(Xs[q], oracle.synth_data_u8(1234, 1, d, global_offset=goff + int(i))[0])      # the generator keys on the global index
            ref, _ = oracle.encode_icm(Xs[q:q + 1], B0s[q:q + 1], K, m, H, ils, J, npert, True, seed, global_offset=goff + int(i))
            assert np.array_equal(ref[0, 0], gots[q]), "vector %d differs" % i
        # objective = mean cost of the returned codes (P10), on the first million
        q = 1_000_000
        c = eng.veccost(dX[:q].cpu().numpy(), dBs[0][:q].cpu().numpy().astype(np.int16) + 1, K, m)
        dq, sq, _ = eng.encode_icm_dev(dX[:q].contiguous(), dB0[:q].contiguous(), dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        assert torch.equal(dq[0], dBs[0][:q])
        assert abs(sq[0] / q - c.astype(np.float64).mean()) <= 1e-6 * sq[0] / q



def test_cfg5_full_per_gpu_share(lsq, oracle):
    """BASELINE configs[4] exactly as ONE GPU of the 8-GPU weak-scaling job runs it (missing #4): 12.5 M x 128 vectors generated ON THE
    DEVICE at rank 1's offset, default chunking (13 resident chunks of 256 x 3968: X = 6.4 GB, byte offsets pass 2^32), 2 ILS iterations.  Rows on both sides
    of EVERY chunk boundary + 32 random rows == the oracle on exactly those vectors (P8); every chunk ran the filtered walk; objective = mean cost
    on one whole chunk (the last full one)."""
    import torch
    d, m, ils, J, npert, seed = 128, 8, [2], 4, 4, 5
    n, goff, chunk = 12_500_000, 12_500_000, 256 * 3968
    with lsq.Engine(0) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
        assert dX.numel() * 4 > (1 << 32)
        dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
        dK = eng.synth_codebooks_dev(4321, m, d)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        torch.cuda.synchronize()
        t = eng.timings()
        assert t["filtered_blocks"] > 0 and t["staged_blocks"] == 0 and t["filter_fallback_chunks"] == 0, t
        K = dK.cpu().numpy()
        rng = np.random.default_rng(12)
        nchunks = (n + chunk - 1) // chunk
        assert nchunks == 13
        rows = [0, n - 1] + [c * chunk + o for c in range(1, nchunks) for o in (-1, 0)] + [int(r) for r in rng.choice(n, size=32, replace=False)]
        rows = np.array(sorted(set(rows)))
        idx = torch.from_numpy(rows).to(dX.device)
        Xs = dX[idx].cpu().numpy()
        B0s = dB0[idx].cpu().numpy().astype(np.int16) + 1
        gots = dBs[0][idx].cpu().numpy().astype(np.int16) + 1
        for q, i in enumerate(rows):
            ref, _ = oracle.encode_icm(Xs[q:q + 1], B0s[q:q + 1], K, m, H, ils, J, npert, True, seed, global_offset=goff + int(i))
            assert np.array_equal(ref[0, 0], gots[q]), "vector %d (chunk %d) differs" % (i, i // chunk)
        lo = (nchunks - 2) * chunk                                   # the last FULL chunk (byte offset of its X rows > 4 GiB)
        c = eng.veccost(dX[lo:lo + chunk].cpu().numpy(), dBs[0][lo:lo + chunk].cpu().numpy().astype(np.int16) + 1, K, m)
        dq, sq, _ = eng.encode_icm_dev(dX[lo:lo + chunk].contiguous(), dB0[lo:lo + chunk].contiguous(), dK, m, ils, J, npert, True, seed=seed,
                                       global_offset=goff + lo)
        assert torch.equal(dq[0], dBs[0][lo:lo + chunk])
        assert abs(sq[0] / chunk - c.astype(np.float64).mean()) <= 1e-6 * sq[0] / chunk


def test_cfg4_per_gpu_share_runs_the_filtered_walk(lsq, oracle):
    """BASELINE configs[3] as ONE GPU of the 8-GPU job runs it: 125 000 x 960 (GIST-like range), m = 8, 16-bit filtered walk by default
    (n >= q16_min), Q16 GEMM epilogue with Kd = 960, the 4096-vector range sample (lsq_icmq.hip: d > 512), global offset of rank 3.
    Default options.  >= 40 random rows + the first / last row + rows around block boundaries (489 vectors per block) vs the oracle run
    on exactly those vectors (results depend on the global index only, P8); counters prove which path produced them."""
    import torch
    d, m, ils, J, npert, seed = 960, 8, [1, 2], 4, 4, 42
    n, goff = 125_000, 375_000
    with lsq.Engine(0, profile=True) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d, global_offset=goff)
        dX.mul_(0.3 / 255.0)
        dB0 = eng.randinit_dev(7, n, m, global_offset=goff)
        dK = eng.synth_codebooks_dev(4321, m, d)
        dK.mul_(0.3 / 255.0)
        dBs, sums, stats = eng.encode_icm_dev(dX, dB0, dK, m, ils, J, npert, True, seed=seed, global_offset=goff)
        torch.cuda.synchronize()
        t = eng.timings()
        assert t["filtered_blocks"] > 0 and t["staged_blocks"] == 0, t
        assert t["filter_refined"] < 0.25 * t["icm_node_updates"], t
        K = dK.cpu().numpy()
        rng = np.random.default_rng(960)
        per = -(-n // 256)
        rows = [0, n - 1] + [b * per + o for b in (1, 2, 128, 255) for o in (-1, 0, 1)] + list(rng.choice(n, size=44, replace=False))
        rows = np.array(sorted(set(int(r) for r in rows if 0 <= r < n)))
        idx = torch.from_numpy(rows).to(dX.device)
        Xs, B0s = dX[idx].cpu().numpy(), dB0[idx].cpu().numpy().astype(np.int16) + 1
        for r, which in ((0, [1]), (1, [1, 2])):
            gots = dBs[r][idx].cpu().numpy().astype(np.int16) + 1
            for q, i in enumerate(rows):
                ref, _ = oracle.encode_icm(Xs[q:q + 1], B0s[q:q + 1], K, m, H, which, J, npert, True, seed, global_offset=goff + int(i))
                assert np.array_equal(ref[-1, 0], gots[q]), "snapshot %d, vector %d differs" % (r, i)
        # objective = mean cost of the returned codes (P10) on the whole shard
        c = eng.veccost(dX.cpu().numpy(), dBs[1].cpu().numpy().astype(np.int16) + 1, K, m)
        assert abs(sums[1] / n - c.astype(np.float64).mean()) <= 1e-6 * sums[1] / n


def test_filter_forced_at_d960_all_vectors(lsq, oracle):
    """d = 960 through the forced filter (q16_min = 0, light = 0) on ALL vectors: the long chains (Kd = 960: 60 K chunks of the GEMM, odd row counts)
    feeding the u16 epilogue, and the exact refinement with d > 257."""
    d, n, m, ils, J, npert, seed = 960, 3001, 8, [2], 3, 4, 96
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    X = (X * np.float32(0.1)).astype(np.float32)
    t = _filter_case(lsq, oracle, X, K, B0, m, ils, J, npert, seed)
    assert t["light_blocks"] == 0, t


# ---- the 16-bit filter under adversarial value distributions -------------------------------------------------------------------
def _filter_case(lsq, oracle, X, K, B0, m, ils, J, npert, seed, expect_filter=True, **options):
    Bs_ref, objs_ref, st_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed, want_stats=True)
    with lsq.Engine(0, schedule=6) as eng:
        eng.set_option("q16_min", 0)
        eng.set_option("light", 0)
        eng.set_option("filter_probe_div", 0)          # the first-iteration probe is tested on its own
        for k, v in options.items():
            eng.set_option(k, v)
        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
        t = eng.timings()
    assert np.array_equal(Bs, Bs_ref), "%d of %d codes differ (%r)" % ((Bs != Bs_ref).sum(), Bs.size, t)
    assert np.array_equal(np.isnan(objs), np.isnan(objs_ref)) and np.allclose(objs[~np.isnan(objs)], objs_ref[~np.isnan(objs)], rtol=1e-5, atol=0)
    if expect_filter:
        assert t["filtered_blocks"] > 0 and t["staged_blocks"] == 0 and t["filter_fallback_chunks"] == 0, t
    else:
        assert t["filtered_blocks"] == 0 and t["staged_blocks"] > 0 and t["filter_fallback_chunks"] > 0, t      # unusable bounds / too many flagged vectors: the f32 walk did the work
    return t


def test_filter_exact_ties_and_near_ties(lsq, oracle):
    """Duplicate codewords (exact ties: equal 16-bit sums AND equal f32 sums -> the lowest index must win through the exact
    refinement), long runs of copies (third candidate in reach -> full-f32 redo), and near-duplicates (gaps far below the
    filter's step)."""
    d, n, m, seed = 32, 6000, 8, 55
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    K = K.reshape(m, H, d).copy()
    K[:, 1::2] = K[:, 0::2]                              # every codeword twice
    K[1, 200:] = K[1, 3]                                 # a run of 56 copies
    K[2, 1::2] = K[2, 0::2] * np.float32(1 + 1e-6)      # near-duplicates
    K = K.reshape(m * H, d)
    t = _filter_case(lsq, oracle, X, K, B0, m, [1, 3], 3, 4, seed)
    assert t["filter_refined"] > 0 and t["filter_exact"] > 2 * t["filter_refined"], t      # some windows hold more than two candidates


@pytest.mark.parametrize("d,m,offset", [(32, 8, 50.0), (30, 8, 1000.0), (33, 16, 5.0), (7, 3, 2.0e4)])
def test_filter_shift_invariant_levels(lsq, oracle, d, m, offset):
    """The levels are taken of U + sigma_i (sigma_i = 2 <x_i, mean codeword>) and of table rows minus their own minimum -- shifts every candidate of
    a node update shares.  Data and codebooks with a LARGE common component (every vector and every codeword moved by `offset` along one direction)
    make those shifts orders of magnitude larger than the differences that decide the argmin: the f32 rounding of the shifts must be inside the
    filter's bound, or codes would differ from the oracle's.  d not a multiple of 4: the scalar path of the shift kernel."""
    n, seed = 20_000, 77
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    u = np.random.default_rng(5).standard_normal(d).astype(np.float32)
    u /= np.linalg.norm(u)
    X = np.ascontiguousarray(X + np.float32(offset) * u)
    K = np.ascontiguousarray(K + np.float32(offset / m) * u)
    t = _filter_case(lsq, oracle, X, K, B0, m, [2], 2, 4, seed)
    assert t["filter_f32"] < 0.05 * t["icm_node_updates"], t      # the shifted range was sampled well: few vectors fall outside it


def test_filter_more_ambiguous_vectors_than_records(lsq, oracle):
    """Every codeword duplicated: EVERY node update is an exact tie, so all ~1200 vectors of a block are ambiguous at once -- more than the 1024
    refinement records a block holds: the overflow takes the one-wave-per-vector f32 routine.  All vectors vs the oracle."""
    d, n, m, seed = 16, 300_000, 8, 56
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    K = K.reshape(m, H, d).copy()
    K[:, 1::2] = K[:, 0::2]
    K = K.reshape(m * H, d)
    t = _filter_case(lsq, oracle, X, K, B0, m, [1], 2, 4, seed)
    assert t["filter_refined"] + t["filter_f32"] > 0.95 * t["icm_node_updates"], t
    assert t["filter_refined"] > 0.5 * t["icm_node_updates"] and t["filter_f32"] > 0, t      # the records were full: some vectors overflowed


def test_filter_light_blocks(lsq, oracle):
    """Light blocks of the filtered walk (one wave per vector, f32 rows gathered from L2): default `light`, a chunk small enough that every block
    is light, with ties, outliers beyond the sampled level range and the four skip x fallback rules."""
    d, n, m, seed = 24, 40_000, 8, 57                      # 157 vectors per block: every block is light; the range sample takes every other 128-vector panel
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    K = K.reshape(m, H, d).copy()
    K[3, 1::2] = K[3, 0::2]                                # one codebook with duplicated codewords: ties at node 3 only
    K = K.reshape(m * H, d)
    X = X.copy()
    hot = np.array([i for i in range(0, n, 53) if (i // 128) % 2 == 1])      # ~1 % of the vectors, in panels the sample does not see
    X[hot] *= np.float32(6.0)                              # far outside the level range: flagged by the GEMM epilogue
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, [1, 3], 3, 4, True, seed)
    for skip in (1, 0):
        for fb in (1, 0):
            with lsq.Engine(0, schedule=6, skip=skip, profile=True) as eng:
                eng.set_option("q16_min", 0)
                eng.set_option("filter_probe_div", 0)
                eng.set_option("fallback", fb)
                eng.set_option("wave_max", 0)
                eng.set_option("filter_fallback_div", 0)
                Bs, objs = eng.encode_icm(X, B0, K, m, [1, 3], 3, 4, True, seed=seed)
                t = eng.timings()
            assert np.array_equal(Bs, Bs_ref), "skip=%d fallback=%d: %d codes differ (%r)" % (skip, fb, (Bs != Bs_ref).sum(), t)
            assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
            assert t["light_blocks"] > 0 and t["filtered_blocks"] == 0 and t["staged_blocks"] == 0, t


def test_filter_offsets_scales_and_single_codebook(lsq, oracle):
    """Value ranges that stress the bound: data far from the origin (|s| >> range: the f32 rounding term dominates the window),
    tiny and huge scales, m = 1 (no tables), m = 16."""
    rng = np.random.default_rng(8)
    for (d, n, m, shift, scale) in ((16, 5000, 8, 300.0, 1.0), (16, 5000, 8, 0.0, 1e-12), (16, 5000, 8, 0.0, 1e12), (24, 4000, 1, 5.0, 1.0), (16, 3000, 16, 50.0, 3.0)):
        X = ((rng.standard_normal((n, d)) + shift) * scale).astype(np.float32)
        K = (((rng.standard_normal((m * H, d)) + shift) * scale) / m).astype(np.float32)
        B0 = oracle.randinit(5, n, m, H)
        _filter_case(lsq, oracle, X, K, B0, m, [2], 2, min(4, m), 31 + m)


def test_filter_steps_aside_for_nonfinite_data(lsq, oracle):
    """NaN / Inf anywhere in the chunk or the codebooks: the bounds are unusable, the filtered launch idles and the f32 walk (with the
    reference's strict-< scan semantics for NaN) does the work -- decided on the device, no host round trip."""
    d, n, m, seed = 16, 5000, 8, 91
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    X[17, 0] = np.nan
    X[99, 3] = np.inf
    _filter_case(lsq, oracle, X, K, B0, m, [2], 3, 4, seed, expect_filter=False)
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    K = K.copy()
    K[300, 2] = -np.inf
    _filter_case(lsq, oracle, X, K, B0, m, [1], 2, 4, seed, expect_filter=False)
    # and a finite chunk after one holding a NaN vector (per-chunk, per-vector decisions): the NaN vector's unaries are flagged by the GEMM
    # epilogue and take the f32 path (or its whole chunk does, when the range sample caught it)
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    X[10, 1] = np.nan
    Bs_ref, _ = oracle.encode_icm(X, B0, K, m, H, [2], 3, 4, True, seed)
    with lsq.Engine(0, schedule=6, chunk=2500) as eng:
        eng.set_option("q16_min", 0)
        eng.set_option("light", 0)
        Bs, _ = eng.encode_icm(X, B0, K, m, [2], 3, 4, True, seed=seed)
        t = eng.timings()
    assert np.array_equal(Bs, Bs_ref)
    assert t["filtered_blocks"] > 0 and (t["staged_blocks"] > 0 or t["filter_f32"] > 0), t


def test_filter_outliers_beyond_the_sampled_range(lsq, oracle):
    """The level range comes from a sample of the chunk; vectors whose unaries leave it are flagged by the GEMM epilogue and take the
    f32 routine one by one.  A few mild outliers (outside the sample for one of the two panel parities), a heavy tail (inside it: coarse step, many refinements),
    and a chunk whose second half is scaled up."""
    d, n, m, seed = 16, 40_000, 8, 77
    rng = np.random.default_rng(seed)
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    ta = {"filter_f32": 0}
    for parity in (0, 1):                                # the sample takes every other 128-vector panel at this n: one parity is unseen
        rows = np.array([i for i in rng.choice(n, size=400, replace=False) if (i // 128) % 2 == parity][:9])
        Xa = X.copy()
        Xa[rows] *= np.float32(3.0)
        t = _filter_case(lsq, oracle, Xa, K, B0, m, [2], 2, 4, seed)
        ta["filter_f32"] += t["filter_f32"]
    Xb = (X * rng.standard_cauchy((n, 1)).astype(np.float32)).astype(np.float32)
    tb = _filter_case(lsq, oracle, Xb, K, B0, m, [2], 2, 4, seed, filter_fallback_div=0, filter_probe_div=0)      # never hand the chunk over
    Xc = X.copy()
    Xc[n // 2:] *= np.float32(1.7)
    tc = _filter_case(lsq, oracle, Xc, K, B0, m, [1], 2, 4, seed)
    assert ta["filter_f32"] + tb["filter_f32"] + tc["filter_f32"] > 0, (ta, tb, tc)      # at least one case exercised the per-vector flags
    # the chunk-level decision: a third of the vectors scaled far outside the sampled range.  Above 1/64 of the pairs the whole chunk goes
    # to the f32 walk; with the hand-over disabled the same chunk runs the exact-256 routine on every flagged pair.  Same codes both ways.
    Xd = X.copy()
    hot = np.array([i for i in range(n) if (i // 128) % 2 == 1 and i % 3 == 0])            # never in the sample (odd panels)
    Xd[hot] *= np.float32(40.0)
    td = _filter_case(lsq, oracle, Xd, K, B0, m, [1], 2, 4, seed, expect_filter=False)
    te = _filter_case(lsq, oracle, Xd, K, B0, m, [1], 2, 4, seed, filter_fallback_div=0, filter_probe_div=0)
    assert te["filter_f32"] > 1000, te


def test_filter_probe_hands_scale_mixture_chunks_to_the_f32_walk(lsq, oracle):
    """Cauchy-scaled vectors: the sampled level range is blown up by a few extreme vectors, the level step is too coarse for the bulk and most node
    updates come out ambiguous.  The first ILS iteration runs filtered, its counters are read back, and the remaining iterations run as the f32 walk
    (option filter_probe_div, default 8): both kernels serve the same call.  Codes == the oracle on all vectors either way; with the probe off the
    whole call stays on the filtered walk."""
    d, n, m, seed = 16, 40_000, 8, 79
    rng = np.random.default_rng(seed)
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    X = (X * rng.standard_cauchy((n, 1)).astype(np.float32)).astype(np.float32)
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, [1, 3], 3, 4, True, seed)
    seen = {}
    for probe in (8, 0):
        with lsq.Engine(0, schedule=6, profile=True) as eng:
            eng.set_option("q16_min", 0)
            eng.set_option("light", 0)
            eng.set_option("filter_fallback_div", 0)
            eng.set_option("filter_probe_div", probe)
            Bs, objs = eng.encode_icm(X, B0, K, m, [1, 3], 3, 4, True, seed=seed)
            t = eng.timings()
        assert np.array_equal(Bs, Bs_ref), "probe=%d: %d codes differ (%r)" % (probe, (Bs != Bs_ref).sum(), t)
        assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0)
        seen[probe] = t
    assert seen[8]["filter_fallback_chunks"] == 1 and seen[8]["filtered_blocks"] > 0 and seen[8]["staged_blocks"] > 0, seen[8]
    assert seen[0]["filter_fallback_chunks"] == 0 and seen[0]["staged_blocks"] == 0, seen[0]
    assert seen[8]["icm_node_updates"] == seen[0]["icm_node_updates"]          # same memoisation on both roads
    # several resident chunks, only the SECOND half of the data heavy-tailed: each chunk decides for itself (the well-conditioned chunks stay filtered)
    Xm = X.copy()
    Xg, _, _ = make_problem(d, n, m, seed=seed, kind="gauss")
    Xm[: n // 2] = Xg[: n // 2]
    ref, _ = oracle.encode_icm(Xm, B0, K, m, H, [3], 3, 4, True, seed)
    with lsq.Engine(0, schedule=6, profile=True, chunk=n // 4) as eng:
        eng.set_option("q16_min", 0)
        eng.set_option("light", 0)
        eng.set_option("filter_fallback_div", 0)
        Bs, _ = eng.encode_icm(Xm, B0, K, m, [3], 3, 4, True, seed=seed)
        t = eng.timings()
    assert np.array_equal(Bs, ref), "%d codes differ (%r)" % ((Bs != ref).sum(), t)
    assert t["filter_fallback_chunks"] == 2, t


def test_single_iteration_calls_probe_their_first_sweep(lsq, oracle):
    """The trainer's pattern -- chained encoding_icm calls, ONE ILS iteration each: the call's launch is split after its first sweep, whose counters are
    the probe; the other sweeps run on the road the probe chose.  Nothing is remembered between calls (round 3 kept a per-shape verdict that could not
    tell two data sets of one shape apart).  Scale-mixture data, 4 chained calls == the oracle's chained calls, every call hands over to the f32 walk
    after its first sweep; a well-conditioned data set never leaves the filtered walk."""
    d, n, m, seed = 16, 40_000, 8, 80
    rng = np.random.default_rng(seed)
    Xg, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    Xc = (Xg * rng.standard_cauchy((n, 1)).astype(np.float32)).astype(np.float32)
    for X, expect_fallbacks in ((Xc, 4), (Xg, 0)):
        with lsq.Engine(0, schedule=6, profile=True) as eng:
            eng.set_option("q16_min", 0)
            eng.set_option("light", 0)
            eng.set_option("filter_fallback_div", 0)
            Bg, Bo = B0.copy(), B0.copy()
            for it in range(4):
                Bg = eng.encoding_icm(X, Bg, K, m, 3, True, 4, seed=seed)                   # the context counts the iterations
                Bo = oracle.encoding_icm_faithful(X, Bo, K, m, H, 3, True, 4, seed, it, nworkers=oracle.num_threads())
                assert np.array_equal(Bg, Bo), "call %d: %d codes differ" % (it, (Bg != Bo).sum())
            t = eng.timings()
        assert t["filter_fallback_chunks"] == expect_fallbacks, t
        assert t["filtered_blocks"] > 0 and (t["staged_blocks"] > 0) == (expect_fallbacks > 0), t


def test_filter_degenerate_ranges(lsq, oracle):
    """All-zero codebooks (every conditioned sum is 0: step 0 -> unusable bounds -> the f32 walk; index 0 wins every argmin), constant
    data, and a single distinct codeword per codebook."""
    d, n, m, seed = 16, 5000, 8, 12
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    _filter_case(lsq, oracle, X, np.zeros_like(K), B0, m, [1], 2, 4, seed, expect_filter=False)
    Kc = np.repeat(K.reshape(m, H, d)[:, :1], H, axis=1).reshape(m * H, d).copy()       # h copies of one codeword: all sums tie
    oracle_B, _ = oracle.encode_icm(X, B0, Kc, m, H, [1], 2, 4, True, seed)
    with lsq.Engine(0, schedule=6) as eng:
        eng.set_option("q16_min", 0)
        eng.set_option("light", 0)
        Bs, _ = eng.encode_icm(X, B0, Kc, m, [1], 2, 4, True, seed=seed)
    assert np.array_equal(Bs, oracle_B)
    Xk = np.full_like(X, 3.25)
    with lsq.Engine(0, schedule=6) as eng:
        eng.set_option("q16_min", 0)
        Bs, _ = eng.encode_icm(Xk, B0, K, m, [2], 2, 4, True, seed=seed)
    ref, _ = oracle.encode_icm(Xk, B0, K, m, H, [2], 2, 4, True, seed)
    assert np.array_equal(Bs, ref)


def test_filter_mild_outliers_are_flagged_before_levels_can_wrap(lsq, oracle):
    """A unary just above the widened sampled range still has a 16-bit level, but with the table levels on top the sum could pass 65535 and wrap
    into a small key.  Such vectors must be flagged (lsq_q16_node::hiq), not only the ones whose own level leaves 16 bits."""
    d, n, m, seed = 16, 40_000, 8, 78
    rng = np.random.default_rng(seed)
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    K = (K * np.float32(2.5)).astype(np.float32)         # table ranges comparable to the unary range: the 16-bit span is much wider than the unaries'
    flagged = 0
    for scale in (1.2, 1.45, 1.8):
        for parity in (0, 1):                            # the sample takes every other 128-vector panel at this n
            rows = np.array([i for i in rng.choice(n, size=3000, replace=False) if (i // 128) % 2 == parity][:300])
            Xa = X.copy()
            Xa[rows] *= np.float32(scale)
            flagged += _filter_case(lsq, oracle, Xa, K, B0, m, [1], 2, 4, seed)["filter_f32"]
    assert flagged > 0


@pytest.mark.parametrize("m", [1, 4, 8, 9, 13, 16])
def test_small_chunk_wave_kernel_every_rule(lsq, oracle, m):
    """Chunks of at most 256 x wave_max vectors run icm_wave_kernel (a wave owns its vectors through the launch).  It must give the oracle's
    codes under the four skip x fallback combinations, for both schedules that can reach it, with a ragged last wave, and agree with the
    block-structured light path (wave_max = 0) in codes AND in the number of node updates recomputed (same memoisation rules)."""
    d, n, ils, J, npert, seed = 16, 2501, [1, 3], 3, min(4, m), 900 + m
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    Bs_ref, objs_ref = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, True, seed)
    for schedule in (6, 4):
        for skip in (1, 0):
            for fb in (1, 0):
                got = {}
                for wave_max in (64, 0):
                    with lsq.Engine(0, schedule=schedule, skip=skip) as eng:
                        eng.set_option("fallback", fb)
                        eng.set_option("wave_max", wave_max)
                        Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
                        t = eng.timings()
                    tag = "m=%d schedule=%d skip=%d fallback=%d wave_max=%d" % (m, schedule, skip, fb, wave_max)
                    assert np.array_equal(Bs, Bs_ref), "%s: %d of %d codes differ" % (tag, (Bs != Bs_ref).sum(), Bs.size)
                    assert np.allclose(objs, objs_ref, rtol=1e-5, atol=0), tag
                    assert t["staged_blocks"] == 0 and t["filtered_blocks"] == 0 and t["light_blocks"] > 0, (tag, t)
                    got[wave_max] = t["icm_node_updates"]
                assert got[64] == got[0], (m, schedule, skip, fb, got)
