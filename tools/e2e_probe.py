"""tools/e2e_probe.py [n] [d] -- host-buffer entry point at full size with the per-class device timings: where does an end-to-end call spend its time?"""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, ".")
lsq = importlib.import_module("local-search-quantization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = 8
with lsq.Engine(0, profile=True) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d)
    if d == 960: dX.mul_(0.3 / 255.0)
    dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, d)
    if d == 960: dK.mul_(0.3 / 255.0)
    Xh, Kh = dX.cpu().numpy(), dK.cpu().numpy(); Bh = dB0.cpu().numpy().astype(np.int16) + 1
    for mode in ("piped", "one piece", "piped"):
        eng.set_option("upload_pipeline_min_bytes", (64 << 20) if mode == "piped" else 0)
        eng.encode_icm(Xh, Bh, Kh, m, [16], 4, 4, True, seed=42)
        eng.reset_timings()
        t0 = time.perf_counter()
        Bs, objs = eng.encode_icm(Xh, Bh, Kh, m, [16], 4, 4, True, seed=42)
        dt = time.perf_counter() - t0
        tm = eng.timings()
        print("%-10s %.2f ms  %.2f M vectors/s" % (mode, dt * 1e3, n / dt / 1e6), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items() if v}, flush=True)
