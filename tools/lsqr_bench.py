"""tools/lsqr_bench.py [n d m] -- the codebook update (update_codebooks, src/codebook_update.jl:52-86): host LSQR (std::thread workers over the
dimensions) against the device LSQR (all dimensions at once) on the same synthetic problem; prints one JSON line."""
import importlib, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m = (int(x) for x in (sys.argv[1:4] + ["100000", "128", "8"][len(sys.argv) - 1:]))
H = 256
rng = np.random.default_rng(0)
B = rng.integers(1, H + 1, size=(m, n)).astype(np.int16)
Ct = [rng.standard_normal((d, H)).astype(np.float32) for _ in range(m)]
X = (sum(Ct[j][:, B[j] - 1] for j in range(m)) + 0.05 * rng.standard_normal((d, n))).astype(np.float32)
t0 = time.perf_counter(); Ch = lsq.update_codebooks(X, B, H); th = time.perf_counter() - t0
with lsq.Engine(0) as eng:
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    dB = torch.from_numpy(np.ascontiguousarray((B.T - 1).astype(np.uint8))).cuda()
    eng.update_codebooks_dev(dX, dB, m); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dK, iters = eng.update_codebooks_dev(dX, dB, m)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter(); Cd = lsq.update_codebooks(X, B, H, engine=eng); tg = time.perf_counter() - t0
Kh, Kd = np.concatenate(Ch, axis=1), np.concatenate(Cd, axis=1)
print(json.dumps(dict(n=n, d=d, m=m, host_s=round(th, 4), host_threads=os.cpu_count(), device_ms=round(td * 1e3, 3), device_host_buffers_ms=round(tg * 1e3, 3),
                      lsqr_iterations=iters, ms_per_iteration=round(td * 1e3 / max(iters, 1), 4), speedup_device_resident=round(th / td, 1),
                      rel_diff_host_vs_device=float(np.linalg.norm(Kd - Kh) / np.linalg.norm(Kh)))))
