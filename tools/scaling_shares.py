"""tools/scaling_shares.py [out.json] -- run ON THE GPU BOX.  PREDICTED multi-GPU scaling from single-GPU shares (no 8-GPU node is available to the
builder: VERDICT r2 #5).  The sweep has no cross-GPU dependency: an N-GPU job's wall time is the time of the largest `splitarray`
shard on one GPU plus one RCCL broadcast of the codebooks (m*256*d floats; bounded below by the measured 2-rank gloo/1-GPU run, taken as 0.2 ms
over xGMI).  So the curve is predicted by timing, on ONE GPU, the share each rank would hold:
    cfg4 (BASELINE configs[3]) strong scaling: 10^6 x 960 over N = 1, 2, 4, 8  ->  shares of 1 000 000 / 500 000 / 250 000 / 125 000 vectors
    cfg2-shaped weak scaling: 10^6 x 128 per GPU (every rank holds the same share: the prediction is flat by construction; listed for the table)
Each share: `bench.py --scaling strong --total <share> --dim <d>` semantics (device-resident inputs, lsq_encode_icm_dev, 16 ILS x 4 sweeps).
Output: shares with ms / vectors/s, predicted job time, speed-up and efficiency vs N = 1 -- labelled "predicted from single-GPU shares"."""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")

BCAST_MS = 0.2


def share_ms(eng, n, d, m=8, ils=16, J=4, steps=4):
    dX = eng.synth_data_u8_dev(1234, n, d)
    dB0 = eng.randinit_dev(7, n, m)
    dK = eng.synth_codebooks_dev(4321, m, d)
    if d == 960:
        dX.mul_(0.3 / 255.0)
        dK.mul_(0.3 / 255.0)
    out = torch.empty((1, n, m), dtype=torch.uint8, device=dX.device)
    eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out)
    torch.cuda.synchronize()
    eng.reset_timings()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timings()
    del dX, dB0, dK, out
    torch.cuda.empty_cache()
    return dt * 1e3, {k: round(tm[k] / steps, 3) for k in ("tables_ms", "unaries_ms", "icm_ms", "cost_ms")}


res = {"label": "PREDICTED from single-GPU shares -- unmeasured on multi-GPU hardware", "broadcast_ms_assumed": BCAST_MS, "strong_cfg4": [], "weak_cfg2": []}
with lsq.Engine(0, profile=True) as eng:
    total = 1_000_000
    base = None
    for N in (1, 2, 4, 8):
        share = -(-total // N)                       # the largest splitarray shard
        ms, parts = share_ms(eng, share, 960)
        job = ms + (BCAST_MS if N > 1 else 0.0)
        base = base or job
        res["strong_cfg4"].append({"gpus": N, "share_vectors": share, "share_ms": round(ms, 3), "share_Mvps": round(share / ms / 1e3, 3), "parts_ms": parts,
                                   "predicted_job_ms": round(job, 3), "predicted_Mvps": round(total / job / 1e3, 3),
                                   "predicted_speedup": round(base / job, 3), "predicted_efficiency": round(base / job / N, 3)})
        print(res["strong_cfg4"][-1], flush=True)
    ms, parts = share_ms(eng, 1_000_000, 128)
    for N in (1, 2, 4, 8):
        job = ms + (BCAST_MS if N > 1 else 0.0)
        res["weak_cfg2"].append({"gpus": N, "share_vectors": 1_000_000, "share_ms": round(ms, 3), "parts_ms": parts, "predicted_job_ms": round(job, 3),
                                 "predicted_Mvps": round(N * 1_000_000 / job / 1e3, 3), "predicted_efficiency": round(ms / job, 3)})
    print(res["weak_cfg2"][-1], flush=True)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scaling_shares.json"
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
