"""tools/train_bench.py [n d m niter ilsiter] -- the training loop (LSQ.jl:10-88) three ways on one synthetic problem: host LSQR + GPU encode through host
buffers (train_lsq), device LSQR + GPU encode through host buffers (device_update=True), everything resident in HBM (train_lsq_dev).  One JSON line."""
import importlib, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m, niter, ilsiter = (int(x) for x in (sys.argv[1:6] + ["100000", "128", "8", "8", "4"][len(sys.argv) - 1:]))
H = 256
with lsq.Engine(0) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d)
    dB0 = eng.randinit_dev(7, n, m)
    X = np.ascontiguousarray(dX.cpu().numpy().T)
    B0 = np.ascontiguousarray((dB0.cpu().numpy().astype(np.int16) + 1).T)
    R = np.eye(d, dtype=np.float32)
    out = {"n": n, "d": d, "m": m, "niter": niter, "ilsiter": ilsiter}
    lsq.train_lsq_dev(dX, m, H, dB0, 1, 1, 4, True, 4, seed=42, engine=eng, norm_codebook=False)      # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); C1, B1, _, _, o1 = lsq.train_lsq(X, m, H, R, B0, None, niter, ilsiter, 4, True, 4, seed=42, engine=eng); out["host_lsqr_host_buffers_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter(); C2, B2, _, _, o2 = lsq.train_lsq(X, m, H, R, B0, None, niter, ilsiter, 4, True, 4, seed=42, engine=eng, device_update=True); out["device_lsqr_host_buffers_s"] = round(time.perf_counter() - t0, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); dK, dB, _, _, o3 = lsq.train_lsq_dev(dX, m, H, dB0, niter, ilsiter, 4, True, 4, seed=42, engine=eng, norm_codebook=False); torch.cuda.synchronize(); out["resident_s"] = round(time.perf_counter() - t0, 3)
    K2 = np.concatenate([np.asarray(c, dtype=np.float32).T for c in C2], axis=0)
    out["resident_equals_device_lsqr_run"] = bool(np.array_equal(dK.cpu().numpy(), K2) and np.array_equal(dB.cpu().numpy().astype(np.int16) + 1, np.asarray(B2).T))
    out["objective_first_last"] = [float(o3[0]), float(o3[-1])]
    out["note"] = "the first two include the k-means of the norm codebook on the host (n scalars, <= 100 sweeps); the resident run was asked not to (norm_codebook=False)"
print(json.dumps(out))
