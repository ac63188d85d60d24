"""tools/linscan_bench.py [n nq d m knn] -- the device ADC scan (csrc/lsq_adc.hip) on synthetic codes: queries/s, table lookups/s,
the breakdown (tables / sample + thresholds / scan / selection) and, on a few queries, the host scan as the CPU figure and checker."""
import importlib, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")

n, nq, d, m, knn = (int(x) for x in (sys.argv[1:6] + ["1000000", "10000", "128", "8", "1000"][len(sys.argv) - 1:]))
H = 256
rng = np.random.default_rng(1)
K = (rng.standard_normal((m * H, d)) * 0.5).astype(np.float32)
codes = rng.integers(0, H, size=(n, m), dtype=np.uint8)
Q = rng.standard_normal((nq, d)).astype(np.float32)
dev = torch.device("cuda:0")
dK, dC, dQ = torch.from_numpy(K).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(Q).to(dev)
recon = torch.zeros((n, d), device=dev)
for j in range(m):
    recon += dK[j * H + dC[:, j].long()]
dN = (recon.double() ** 2).sum(1).float().contiguous()
del recon
with lsq.Engine(0, profile=True) as eng:
    eng.linscan_dev(dC, dQ, dK, dN, m, knn)                      # warm-up at full size (allocates the scan's work buffers)
    torch.cuda.synchronize()
    eng.reset_timings()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        dd, di = eng.linscan_dev(dC, dQ, dK, dN, m, knn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = eng.linscan_stats()
    # host scan on a few queries: checker + CPU figure
    nh = min(nq, 32)
    t1 = time.perf_counter()
    hd, hi = lsq.linscan_lsq(codes.T, Q[:nh].T, [np.ascontiguousarray(K[j * H:(j + 1) * H].T) for j in range(m)], dN.cpu().numpy(),
                             np.eye(d, dtype=np.float32), knn)
    th = time.perf_counter() - t1
    same = bool(np.array_equal(hi.T, di[:nh].cpu().numpy()) and np.array_equal(hd.T, dd[:nh].cpu().numpy()))
lookups = float(n) * nq * m
print(json.dumps(dict(n=n, nq=nq, d=d, m=m, knn=knn, ms=round(dt * 1e3, 3), queries_per_s=round(nq / dt, 1), lookups_per_s=lookups / dt,
                      lds_gather_TBps=round(lookups * 4 / dt / 1e12, 2),
                      scan_only_lds_gather_TBps=round(lookups * 4 * reps / (st["scan_ms"] * 1e-3) / 1e12, 2) if st["scan_ms"] > 0 else None,
                      breakdown_ms={k: round(st[k] / reps, 3) for k in ("lut_ms", "sample_ms", "scan_ms", "select_ms")},
                      candidates_per_query=round(st["candidates"] / max(st["queries"], 1), 1), fallback_queries=st["fallback_queries"],
                      threshold_rank=st["threshold_rank"], list_capacity=st["list_capacity"], batches=st["batches"] // reps, exhaustive=st["exhaustive"],
                      host_scan=dict(queries=nh, s=round(th, 3), queries_per_s=round(nh / th, 1), threads=os.cpu_count(), same_results=same))), flush=True)
