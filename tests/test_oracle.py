"""CPU tests of the oracle (oracle/lsq_oracle.c): the RNG against the published Random123
known-answer vectors, and the ILS/ICM restatement against (a) the properties P1-P10 that follow
from the reference sources and (b) an independent numpy implementation."""
import numpy as np
import pytest

import pyref
from conftest import make_problem

H = 256


# ---- Philox4x32-10: Random123 kat_vectors (philox4x32 10 rounds) ------------------------------
@pytest.mark.parametrize("ctr,key,expect", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(oracle, ctr, key, expect):
    assert oracle.philox4x32_10(ctr, key).tolist() == expect


def test_rng_word_layout(oracle):
    # word w of stream (seed; idx, it, domain) = philox(ctr=(idx_lo, idx_hi, it, domain<<16 | w>>2), key=seed)[w & 3]
    seed, idx, it, dom = 0x0123456789abcdef, 0xfedcba9876543210, 7, 3
    for w in (0, 1, 5, 18, 31):
        blk = oracle.philox4x32_10([idx & 0xffffffff, idx >> 32, it, (dom << 16) | (w >> 2)], [seed & 0xffffffff, seed >> 32])
        assert oracle.rng_word(seed, idx, it, dom, w) == int(blk[w & 3])


@pytest.mark.parametrize("m", [1, 2, 7, 8, 16])
def test_perm_is_permutation(oracle, m):
    seen = set()
    for it in range(20):
        p = oracle.perm(42, it, m, True)
        assert sorted(p.tolist()) == list(range(m))
        seen.add(tuple(p.tolist()))
    assert oracle.perm(42, 3, m, False).tolist() == list(range(m))
    if m >= 7:
        assert len(seen) > 10        # P7: a fresh order per ILS iteration


@pytest.mark.parametrize("m,npert", [(8, 4), (7, 4), (16, 4), (4, 4), (3, 5), (8, 0), (8, 1)])
def test_perturb_P5(oracle, m, npert):
    """P5: exactly min(npert, m) distinct positions are rewritten, values uniform over 0..h-1."""
    base = np.full(m, 255, dtype=np.uint8)
    counts = np.zeros(m, dtype=np.int64)
    vals = []
    N = 3000
    for i in range(N):
        marker = oracle.perturb(9, i, 2, base, 255, npert)         # h=255 -> new values < 255 mark the touched slots
        touched = marker != 255
        assert touched.sum() == min(npert, m)
        counts += touched
        vals.extend(oracle.perturb(9, i, 2, base, H, npert)[touched].tolist())
    if 0 < npert < m:
        expect = N * npert / m
        assert np.all(np.abs(counts - expect) < 6 * np.sqrt(expect)), counts      # every position equally likely
        hist = np.bincount(np.array(vals), minlength=H)
        assert hist.min() > 0 and hist.max() < 5 * hist.mean()


def test_perturb_matches_pyref(oracle):
    for m, npert in [(8, 4), (16, 3), (5, 5)]:
        for i in range(50):
            code = (np.arange(m) * 17 % 256).astype(np.uint8)
            ref = pyref.perturb(oracle.rng_word, 5, 1000 + i, 3, code.astype(np.int64), H, npert)
            assert oracle.perturb(5, 1000 + i, 3, code, H, npert).tolist() == ref.tolist()
        assert oracle.perm(5, 9, m, True).tolist() == pyref.perm(oracle.rng_word, 5, 9, m, True)


def test_randinit_range_and_offset(oracle):
    B = oracle.randinit(3, 1000, 8, H)
    assert B.dtype == np.int16 and B.min() >= 1 and B.max() <= H and len(np.unique(B)) > 200
    assert np.array_equal(oracle.randinit(3, 400, 8, H, global_offset=600), B[600:])          # shard-invariant


# ---- tables ---------------------------------------------------------------------------------
def test_tables_exact_on_integer_data(oracle):
    """With small-integer inputs every product and partial sum is exact in f32, so the fmaf chain
    must equal integer arithmetic: checks indexing/layout of sqnorms, tables and unaries."""
    rng = np.random.default_rng(0)
    d, n, m = 24, 20, 3
    K = rng.integers(-8, 9, size=(m * H, d)).astype(np.float32)
    X = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    K64, X64 = K.astype(np.float64), X.astype(np.float64)
    assert np.array_equal(oracle.sqnorms(K), (K64 * K64).sum(1).astype(np.float32))
    G = 2.0 * K64 @ K64.T                                   # G[(j,a),(k,b)]
    T = oracle.tables(K, m, H)
    for j in range(m):
        for k in range(m):
            assert np.array_equal(T[j, k], G[j * H:(j + 1) * H, k * H:(k + 1) * H].T.astype(np.float32))   # T[j,k,b,a]
    U = oracle.unaries(X, K, m, H)
    Uref = (-2.0 * X64 @ K64.T + (K64 * K64).sum(1)[None, :]).astype(np.float32)                           # [i, (j,a)]
    assert np.array_equal(U, Uref.reshape(n, m, H).transpose(1, 0, 2))


def test_tables_close_to_float64(oracle):
    X, K, _ = make_problem(128, 50, 4, seed=1, kind="gauss")
    T = oracle.tables(K, 4, H)
    G = 2.0 * K.astype(np.float64) @ K.astype(np.float64).T
    assert np.allclose(T[1, 2], G[H:2 * H, 2 * H:3 * H].T, rtol=0, atol=1e-5 * np.abs(G).max())
    assert np.array_equal(T[1, 2], T[2, 1].T)               # bitwise symmetric (products commute)


def test_P1_energy_identity(oracle):
    """||x - sum c||^2 = ||x||^2 + sum_j U_j[b_j] + sum_{j<k} Bin_jk[b_j,b_k]   (utils.jl:107-118,134-141)."""
    X, K, B0 = make_problem(64, 40, 5, seed=2, kind="gauss")
    m = 5
    U, T = oracle.unaries(X, K, m, H), oracle.tables(K, m, H)
    codes = (B0 - 1).astype(np.int64)
    cost = oracle.veccost(X, K, codes.astype(np.uint8), H).astype(np.float64)
    for i in range(X.shape[0]):
        e = float((X[i].astype(np.float64) ** 2).sum())
        e += sum(float(U[j, i, codes[i, j]]) for j in range(m))
        e += sum(float(T[j, k, codes[i, k], codes[i, j]]) for j in range(m) for k in range(j + 1, m))
        assert abs(e - cost[i]) <= 2e-5 * max(1.0, abs(cost[i]))


def test_P2_P3_node_update(oracle):
    """P2: a node update never increases the energy; P3: it returns the lowest-index argmin."""
    X, K, B0 = make_problem(32, 30, 4, seed=3, kind="gauss")
    m = 4
    K = K.reshape(m, H, -1).copy()
    K[:, 1::2] = K[:, 0::2]                                  # exact duplicates -> ties
    K = K.reshape(m * H, -1)
    U, T = oracle.unaries(X, K, m, H), oracle.tables(K, m, H)
    codes = (B0 - 1).astype(np.uint8)
    for i in range(X.shape[0]):
        c = codes[i].copy()
        e_prev = float(oracle.veccost(X[i:i + 1], K, c[None], H)[0])
        for j in range(m):
            b = oracle.icm_node(U[j, i], T, c, j)
            assert b == pyref.node_update(U[j, i], T, c.astype(np.int64), j)
            assert b % 2 == 0                                # the lower of the two tied duplicates
            c[j] = b
            e = float(oracle.veccost(X[i:i + 1], K, c[None], H)[0])
            assert e <= e_prev * (1 + 1e-5) + 1e-4
            e_prev = e


def test_veccost_matches_pyref(oracle):
    for d in (7, 64, 100, 128, 200):
        X, K, B0 = make_problem(d, 6, 3, seed=d, kind="gauss")
        c = oracle.veccost(X, K, (B0 - 1).astype(np.uint8), H)
        ref = np.array([pyref.cost(X[i], K, (B0[i] - 1).astype(np.int64), H) for i in range(6)], dtype=np.float32)
        assert np.array_equal(c, ref)


# ---- the whole call ---------------------------------------------------------------------------
@pytest.mark.parametrize("d,n,m,ils,J,npert,randord,seed", [
    (16, 24, 4, [1, 2], 2, 2, True, 1),
    (32, 16, 7, [2], 2, 4, True, 2),
    (24, 12, 8, [1, 3], 1, 4, False, 3),
    (8, 10, 1, [2], 2, 1, True, 4),
])
def test_encode_matches_independent_numpy(oracle, d, n, m, ils, J, npert, randord, seed):
    X, K, B0 = make_problem(d, n, m, seed=seed, kind="gauss")
    U, T = oracle.unaries(X, K, m, H), oracle.tables(K, m, H)
    Bs_ref, objs_ref, st_ref = pyref.encode(oracle.rng_word, X, B0, K, U, T, H, ils, J, npert, randord, seed, global_offset=5)
    Bs, objs, st = oracle.encode_icm(X, B0, K, m, H, ils, J, npert, randord, seed, global_offset=5, want_stats=True)
    assert np.array_equal(Bs, Bs_ref)
    assert np.allclose(objs, objs_ref, rtol=1e-6)
    assert np.array_equal(st.astype(np.int64), st_ref)


def test_P6_accept_rule_and_monotone_cost(oracle):
    X, K, B0 = make_problem(32, 200, 4, seed=11, kind="gauss")
    ils = [1, 2, 3, 4, 5, 6]
    Bs, objs = oracle.encode_icm(X, B0, K, 4, H, ils, 2, 2, True, 7)
    prev = oracle.veccost(X, K, (B0 - 1).astype(np.uint8), H)
    prevB = B0
    for r in range(len(ils)):
        c = oracle.veccost(X, K, (Bs[r] - 1).astype(np.uint8), H)
        assert np.all(c <= prev)                             # per-vector cost never increases
        same = c == prev
        assert np.array_equal(Bs[r][same], prevB[same])      # not strictly better -> input column bit for bit
        assert abs(objs[r] - c.astype(np.float64).mean()) <= 1e-6 * objs[r]      # P10
        prev, prevB = c, Bs[r]
    assert objs[-1] < objs[0]


def test_P8_sharding_and_thread_invariance(oracle):
    X, K, B0 = make_problem(32, 101, 4, seed=12, kind="gauss")
    full, objs = oracle.encode_icm(X, B0, K, 4, H, [3], 2, 2, True, 99)
    a, _ = oracle.encode_icm(X[:37], B0[:37], K, 4, H, [3], 2, 2, True, 99, global_offset=0)
    b, _ = oracle.encode_icm(X[37:], B0[37:], K, 4, H, [3], 2, 2, True, 99, global_offset=37)
    assert np.array_equal(np.concatenate([a, b], axis=1), full)
    other, _ = oracle.encode_icm(X, B0, K, 4, H, [3], 2, 2, True, 100)
    assert not np.array_equal(other, full)                   # the seed matters


def test_faithful_loop_nest_is_bit_identical(oracle):
    """The structure-faithful restatement of encoding_icm (whole-array sweeps, per-worker shards,
    unaries recomputed every call) that bench.py times as the CPU baseline == the per-vector oracle."""
    X, K, B0 = make_problem(48, 150, 5, seed=13, kind="gauss")
    Bs, _ = oracle.encode_icm(X, B0, K, 5, H, [1, 2, 3], 3, 2, True, 21)
    B = B0
    for it in range(3):
        B = oracle.encoding_icm_faithful(X, B, K, 5, H, 3, True, 2, 21, it, nworkers=1 + 2 * it)      # 1, 3, 5 workers
        assert np.array_equal(B, Bs[it])


def test_bad_arguments(oracle):
    X, K, B0 = make_problem(8, 4, 2, seed=1, kind="gauss")
    with pytest.raises(ValueError):
        oracle.encode_icm(X, B0 * 0, K, 2, H, [1], 1, 1, True, 1)      # codes must be 1-based
    with pytest.raises(ValueError):
        oracle.encode_icm(X, B0, K, 2, H, [0], 1, 1, True, 1)
    Bs, objs = oracle.encode_icm(X[:0], B0[:0], K, 2, H, [1], 1, 1, True, 1)   # empty input
    assert Bs.shape == (1, 0, 2)


def test_oracle_side_trainer_follows_the_reference_step_order(oracle):
    """oracle/train_oracle.py (the independent checker of SURVEY 8(f)-4, LSQ.jl:36-66): update -> ilsiter x encode -> qerror; the objective it records is
    the qerror of the codes the previous encode returned, and alternating minimisation must not increase it; unused codewords stay at LSQR's minimum-norm 0."""
    from oracle import train_oracle
    rng = np.random.default_rng(4)
    n, d, m, h = 1500, 12, 3, 256
    cen = rng.standard_normal((40, d)).astype(np.float32) * 3
    X = (cen[rng.integers(40, size=n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    B0 = oracle.randinit(5, n, m, h)
    K, B, obj = train_oracle.train_lsq(X, m, h, B0, 3, 2, 3, True, 2, seed=9)
    assert obj.shape == (3,) and np.all(np.diff(obj) <= 1e-6 * obj[:-1])
    assert abs(oracle.qerror(X, B, K, m, h) - obj[-1]) <= obj[-1]          # same scale (one more update + encode after obj[-1])
    S = train_oracle.sparsify_codes(B0, h)
    assert S.shape == (n, m * h) and np.all(np.asarray(S.sum(1)).ravel() == m)
    K0 = train_oracle.update_codebooks(X, B0, h)
    used = np.asarray(S.sum(0)).ravel() > 0
    assert np.all(K0[~used] == 0)
