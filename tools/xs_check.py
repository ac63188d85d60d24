"""First-light / A-B tool for schedule 7 (csrc/lsq_icmx.hip: the slices of a node spread over the CUs of an XCD).

  python tools/xs_check.py parity          small shapes, every vector against the oracle, schedule 7 forced at any n
  python tools/xs_check.py ab [n] [m] [d]  schedule 6 vs 7 on one device-generated problem: identical codes, ICM ms per step of both

Prints one JSON object per line (gpurun_out/xs_check.jsonl collects them)."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
lsq = importlib.import_module("local-search-quantization_amd")


def emit(obj):
    print(json.dumps(obj), flush=True)


def parity():
    import oracle as O
    from conftest import make_problem
    O.build()
    cases = [(16, 3000, 8, "gauss"), (16, 70000, 8, "gauss"), (16, 5000, 16, "gauss"), (16, 4000, 7, "gauss"), (128, 20000, 8, "sift"),
             (16, 3000, 2, "gauss"), (16, 3000, 12, "gauss"), (32, 300, 8, "gauss"), (16, 40000, 16, "gauss")]
    for d, n, m, kind in cases:
        ils, J, npert, seed = [1, 2], 3, min(4, m), 77 + m
        X, K, B0 = make_problem(d, n, m, seed=seed, kind=kind)
        ref, objs_ref = O.encode_icm(X, B0, K, m, 256, ils, J, npert, True, seed)
        for skip in (1, 0):
            with lsq.Engine(0, schedule=7, skip=skip, tuning=True) as eng:
                for k, v in (("q16_min", 0), ("xs_min", 0), ("filter_probe_div", 0), ("filter_fallback_div", 0)):
                    eng.set_option(k, v)
                t0 = time.time()
                try:
                    Bs, objs = eng.encode_icm(X, B0, K, m, ils, J, npert, True, seed=seed)
                    err = None
                except Exception as e:      # noqa: BLE001 -- report and go on: the next case may tell more
                    Bs, objs, err = None, None, str(e)
                t = eng.timings()
            emit({"case": [d, n, m, kind], "skip": skip, "error": err, "differ": None if Bs is None else int((Bs != ref).sum()), "of": int(ref.size),
                  "obj_ok": None if objs is None else bool(np.allclose(objs, objs_ref, rtol=1e-5, atol=0)), "xs_launches": t["xs_launches"],
                  "xs_fallback": t["xs_fallback_launches"], "refined": t["filter_refined"], "f32": t["filter_f32"], "nodes": t["icm_node_updates"],
                  "s": round(time.time() - t0, 3)})


def ab(n=1_000_000, m=8, d=128, ils=16):
    import torch
    res = {}
    codes = {}
    for schedule in (6, 7, 6, 7):
        with lsq.Engine(0, schedule=schedule, profile=True, tuning=True) as eng:
            dX = eng.synth_data_u8_dev(1234, n, d)
            dB0 = eng.randinit_dev(7, n, m)
            dK = eng.synth_codebooks_dev(99, m, d)
            eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=42)      # warm-up
            eng.reset_timings()
            torch.cuda.synchronize()
            t0 = time.time()
            steps = 3
            for _ in range(steps):
                dBs, sums, _ = eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=42)
            torch.cuda.synchronize()
            wall = (time.time() - t0) / steps
            t = eng.timings()
            codes[schedule] = dBs.cpu().numpy()
            res[schedule] = {"schedule": schedule, "n": n, "m": m, "d": d, "ms_per_step": round(wall * 1e3, 3), "icm_ms": round(t["icm_ms"] / steps, 3),
                             "unaries_ms": round(t["unaries_ms"] / steps, 3), "cost_ms": round(t["cost_ms"] / steps, 3), "xs_launches": t["xs_launches"],
                             "xs_fallback": t["xs_fallback_launches"], "nodes": t["icm_node_updates"] // steps, "refined": t["filter_refined"] // steps,
                             "Mvec_s": round(n / wall / 1e6, 3), "obj": float(sums[0] / n)}
            emit(res[schedule])
    emit({"ab": [n, m, d], "codes_equal": bool(np.array_equal(codes[6], codes[7])), "differ": int((codes[6] != codes[7]).sum())})


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        parity()
    else:
        args = [int(a) for a in sys.argv[2:]]
        ab(*args)
