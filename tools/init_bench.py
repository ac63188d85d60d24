"""tools/init_bench.py [n] [d] [m] -- run ON THE GPU BOX: the initialisers' two device kernels (csrc/lsq_init.hip) timed on device-resident data:
Viterbi chain encode (encode_chain.jl:2-123) and the all-sub-spaces nearest-codeword assignment (PQ.jl:12-41), with the unary GEMM they both start with."""
import importlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
lsq = importlib.import_module("local-search-quantization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8
with lsq.Engine(0, profile=True) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d)
    dK = eng.synth_codebooks_dev(4321, m, d)
    out = {"n": n, "d": d, "m": m}
    for name, fn in (("viterbi", lambda: eng.encode_viterbi_dev(dX, dK, m)), ("assign_codewords", lambda: eng.assign_codewords_dev(dX, dK, m))):
        fn(); torch.cuda.synchronize()
        eng.reset_timings()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        tm = eng.timings()
        out[name] = {"ms_per_call": ms, "unary_gemm_ms": tm["unaries_ms"] / 3, "tables_ms": tm["tables_ms"] / 3, "kernel_ms": tm["other_ms"] / 3,
                     "vectors_per_s": n / ms * 1e3, "ns_per_vector_kernel": tm["other_ms"] / 3 * 1e6 / n}
    # viterbi: (m - 1) x 65 536 (add, compare, 2 selects) per vector, 4 VALU lane-operations each; peak = 256 CUs x 64 lanes x clock
    out["viterbi"]["valu_lane_ops"] = 4.0 * (m - 1) * 65536 * n
    out["viterbi"]["frac_of_valu_issue_peak_at_2.1GHz"] = out["viterbi"]["valu_lane_ops"] / (out["viterbi"]["kernel_ms"] * 1e-3) / (256 * 64 * 2.1e9)
    # assignment: reads the m x n x 1 KiB f32 unaries once
    out["assign_codewords"]["hbm_GBs"] = m * n * 1024 / (out["assign_codewords"]["kernel_ms"] * 1e-3) / 1e9
print(json.dumps(out))
