"""tools/knob_run.py n d m [key=value ...] -- one timed encode configuration in a fresh process (the tuning build reads its
environment knobs once per process): prints one JSON line with the step time and the per-class timings.
    LSQ_WALKQ_BPC=2 python tools/knob_run.py 125000 960 8 tuning=1
"""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")

n, d, m = (int(x) for x in sys.argv[1:4])
opts = dict(kv.split("=") for kv in sys.argv[4:])
tuning = int(opts.pop("tuning", 0))
ils, J, steps = int(opts.pop("ils", 16)), int(opts.pop("J", 4)), int(opts.pop("steps", 5))
with lsq.Engine(0, profile=True, tuning=bool(tuning)) as eng:
    for k, v in opts.items():
        eng.set_option(k, int(v))
    dX = eng.synth_data_u8_dev(1234, n, d)
    if d == 960:
        dX.mul_(0.3 / 255.0)
    dB0 = eng.randinit_dev(7, n, m)
    dK = eng.synth_codebooks_dev(4321, m, d)
    if d == 960:
        dK.mul_(0.3 / 255.0)
    out = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
    eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out)
    torch.cuda.synchronize()
    eng.reset_timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        _, sums, _ = eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timings()
    knobs = {k: v for k, v in os.environ.items() if k.startswith("LSQ_")}
    print(json.dumps(dict(n=n, d=d, m=m, tuning=tuning, opts=opts, knobs=knobs, ms=round(dt * 1e3, 3), Mvps=round(n / dt / 1e6, 3),
                          icm_ms=round(tm["icm_ms"] / steps, 3), unaries_ms=round(tm["unaries_ms"] / steps, 3), cost_ms=round(tm["cost_ms"] / steps, 3),
                          tables_ms=round(tm["tables_ms"] / steps, 3), light=tm["light_blocks"] // steps, filtered=tm["filtered_blocks"] // steps,
                          staged=tm["staged_blocks"] // steps, node_updates_per_launch=tm["icm_node_updates"] / max(tm["icm_launches"], 1), obj=float(sums[0] / n), codes_sum=int(out.to(torch.int64).sum().item()))), flush=True)
