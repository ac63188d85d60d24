import importlib, json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
lsq = importlib.import_module("local-search-quantization_amd")
res = []
def run(n, d, m, sched, opts, ils=16, J=4, steps=3):
    with lsq.Engine(0, profile=True, schedule=sched) as eng:
        for k, v in opts.items(): eng.set_option(k, v)
        dX = eng.synth_data_u8_dev(1234, n, d); dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, d)
        out = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
        eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out); torch.cuda.synchronize()
        eng.reset_timings(); t0 = time.perf_counter()
        for _ in range(steps): _, sums, _ = eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        tm = eng.timings()
        r = dict(n=n, d=d, m=m, sched=sched, opts=opts, ms=dt * 1e3, Mvps=n / dt / 1e6, icm_ms=tm["icm_ms"] / steps, unaries_ms=tm["unaries_ms"]/steps, cost_ms=tm["cost_ms"]/steps,
                 staged=tm["staged_blocks"] // steps, light=tm["light_blocks"] // steps, team=tm["filtered_blocks"] // steps, obj=float(sums[0] / n))
        print(json.dumps(r), flush=True); res.append(r)
import os
os.makedirs("gpurun_out/r02g", exist_ok=True)
for n, d, m in ((1_000_000, 128, 8), (1_000_000, 128, 16), (125_000, 960, 8), (1_000_000, 128, 7), (1_000_000, 128, 4), (500_000, 128, 8), (250_000, 128, 8), (100_000, 128, 8), (10_000, 128, 8), (1_000_000, 128, 12)):
    run(n, d, m, 6, {})
    run(n, d, m, 4, {})
json.dump(res, open("gpurun_out/r02g/sched_cmp.json", "w"), indent=1)
