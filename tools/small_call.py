"""tools/small_call.py -- run ON THE GPU BOX.  Per-call time at BASELINE cfg1's shape (n = 10 000, d = 128, m = 8): one ILS iteration per call as
train_lsq makes it (demo_lsq.jl:34), device-buffer and host-buffer entry points."""
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, ".")
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m = 10000, 128, 8
with lsq.Engine(0, profile=True) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d); dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, d)
    out = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
    for ils in (1, 16):
        for _ in range(3): eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=42, out=out)
        torch.cuda.synchronize(); eng.reset_timings(); t0 = time.perf_counter()
        K = 50
        for _ in range(K): eng.encode_icm_dev(dX, dB0, dK, m, [ils], 4, 4, True, seed=42, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        tm = eng.timings()
        print("device entry n=%d ils=%d: %.1f us per call  (%.2f M vectors/s)" % (n, ils, dt * 1e6, n / dt / 1e6), {k: round(tm[k] / K * 1e3, 1) for k in ("tables_ms", "unaries_ms", "icm_ms", "cost_ms", "other_ms")}, "us")
    Xh, Kh = dX.cpu().numpy(), dK.cpu().numpy(); Bh = dB0.cpu().numpy().astype(np.int16) + 1
    for _ in range(3): eng.encode_icm(Xh, Bh, Kh, m, [1], 4, 4, True, seed=42)
    t0 = time.perf_counter()
    for _ in range(50): eng.encode_icm(Xh, Bh, Kh, m, [1], 4, 4, True, seed=42)
    dt = (time.perf_counter() - t0) / 50
    print("host entry n=%d ils=1: %.1f us per call (%.2f M vectors/s per call)" % (n, dt * 1e6, n / dt / 1e6))
