"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol that
include/lsq_mi355x.h declares, its host-only functions agree with the oracle, and -- with no GPU --
every compute entry point fails LOUDLY (there is no CPU fallback in the product)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 256


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lsq_mi355x.h")).read()
    return sorted(set(re.findall(r"LSQ_API\s+[\w\s\*]*?\b(lsq_\w+)\s*\(", hdr)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    for must in ("lsq_encode_icm", "lsq_encode_icm_dev", "lsq_encoding_icm", "lsq_encode_icm_fully", "lsq_get_unaries",
                 "lsq_get_binaries", "lsq_veccost", "lsq_qerror", "lsq_perturb", "lsq_randinit", "lsq_splitarray",
                 "lsq_create", "lsq_destroy", "lsq_last_error"):
        assert must in syms
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol(lsq):
    lib = C.CDLL(lsq._lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), "liblsq_mi355x.so does not export %s" % name
    # and the ctypes table covers exactly the header
    assert sorted(lsq._lib.SIGNATURES) == _declared_symbols()
    assert lsq._lib.load().lsq_version() >= 200
    # the tuning build (-DLSQ_TUNING: ablations, environment knobs, clock stamps) has the same ABI
    tun = lsq._lib.load(tuning=True)
    assert tun.lsq_version() == lsq._lib.load().lsq_version()


def test_product_library_carries_no_tuning_code(lsq):
    """The shipped .so must not contain the timing-only ablation variants, the environment knobs or the earlier
    schedules: they are compiled into liblsq_mi355x_tuning.so only."""
    blob = open(lsq._lib.LIB_PATH, "rb").read()
    tun = open(lsq._lib.TUNING_LIB_PATH, "rb").read()
    for marker in (b"LSQ_WALK_DIRECT", b"LSQ_WALK_SL", b"LSQ_COST_V2", b"LSQ_GEMM_BK", b"LSQ_Q16_ABL", b"lsq_tuning_set_walkq_debug", b"lsq_tuning_set_walkq_block_clock", b"LSQ_COST_ABL", b"LSQ_GEMM_STAGGER"):
        assert marker not in blob, "%s found in the product library" % marker.decode()
    assert b"lsq_tuning_set_walkq_debug" in tun and b"LSQ_WALK_DIRECT" in tun


def test_no_torch_types_and_no_oracle_in_the_product():
    """The product must not import / link the oracle, and the C-ABI header must stay plain C."""
    hdr = open(os.path.join(ROOT, "include", "lsq_mi355x.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                   # prototypes only, comments stripped
    assert "torch" not in code and "at::" not in code and "std::" not in code and "#include <hip" not in code
    pkg = os.path.join(ROOT, "local-search-quantization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), "%s imports the oracle" % f
                assert "lsq_oracle" not in src.replace("oracle/lsq_oracle.c", ""), "%s links the oracle" % f


def test_host_functions_match_oracle(lsq, oracle):
    assert np.array_equal(lsq.randinit_rows(500, 7, H, seed=11, global_offset=3), oracle.randinit(11, 500, 7, H, global_offset=3))
    assert lsq.randinit(20, 8, H, seed=1).shape == (8, 20)           # Julia shape m x n
    for m in (1, 2, 7, 8, 16):
        for it in range(5):
            assert np.array_equal(lsq.node_order(42, it, m, True), oracle.perm(42, it, m, True))
        assert lsq.node_order(42, 0, m, False).tolist() == list(range(m))


def test_splitarray_matches_reference_semantics(lsq):
    """src/utils.jl:152-177: contiguous parts, the first n mod p parts get one extra element."""
    assert lsq.split_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert lsq.split_ranges(9, 3) == [(0, 3), (3, 6), (6, 9)]
    assert lsq.split_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    parts = lsq.splitarray(range(1, 1000001), 8)                     # Julia-style 1:n
    assert [len(p) for p in parts] == [125000] * 8 and parts[0][0] == 1 and parts[-1][-1] == 1000000
    n = 1000003
    r = lsq.split_ranges(n, 8)
    assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert [e - s for s, e in r] == [125001] * 3 + [125000] * 5


def test_fails_loudly_without_gpu(lsq):
    if lsq.device_count() > 0:
        pytest.skip("a GPU is present; the loud-failure path is for GPU-less hosts")
    with pytest.raises(lsq._lib.LsqError) as e:
        lsq.Engine(0)
    assert e.value.code == lsq._lib.LSQ_ENODEV
    X = np.zeros((4, 8), np.float32)
    with pytest.raises(lsq._lib.LsqError):
        lsq.encode_icm_cuda(X.T, np.ones((2, 4), np.int16), [np.zeros((8, H), np.float32)] * 2, [1], 1, 1, True)


def test_bad_arguments_rejected_on_host(lsq):
    L = lsq._lib.load()
    assert L.lsq_randinit(1, 0, 4, 0, H, None) == lsq._lib.LSQ_EINVAL
    assert L.lsq_node_order(1, 0, 17, 1, None) == lsq._lib.LSQ_EINVAL
    s, ln = C.c_int64(), C.c_int64()
    assert L.lsq_splitarray(10, 0, 0, C.byref(s), C.byref(ln)) == lsq._lib.LSQ_EINVAL
    assert b"splitarray" in L.lsq_last_error()


def test_bench_matches_pmc_profiles_by_workload():
    """bench.py reports `roofline.traffic` only from a committed PMC profile of the SAME build and the SAME workload flags: the key ignores
    measurement flags (steps, warmup, legs) and keeps every shape-defining one."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    k = b.workload_key
    assert k([]) == k("--steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-sample-parity".split()) == ""
    assert k("--codebooks 16 --steps 1".split()) == "--codebooks=16" != k([])
    assert k("--scaling strong --total 125000 --dim 960".split()) == k("--dim 960 --total=125000 --scaling strong --warmup 0".split())
    assert k("--vectors 12500000".split()) != k("--vectors 1000000".split())


def test_bench_plain_multi_gpu_form_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start its own ranks (missing #2) -- here, without a GPU, every rank stops at
    "needs a GPU": what is checked is that the plain form reaches the ranks instead of asking for a launcher."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (the GPU-side one is tests/test_gpu_depth.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "launch with torch.distributed.run" not in p.stderr, p.stderr[-1500:]
