"""tools/neighbours_run.py [n d m] -- one call each of the two device neighbours of the path on a device-generated problem:
the LSQR codebook update (lsq_update_codebooks_dev) and the norm quantisation (lsq_quantize_norms_dev).  Run under rocprofv3 by
tools/profile_neighbours.sh; prints one JSON line (wall times, LSQR iterations)."""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m = (int(x) for x in (sys.argv[1:4] + ["1000000", "128", "8"][len(sys.argv) - 1:]))
H = 256
with lsq.Engine(0) as eng:
    dX = eng.synth_data_u8_dev(1234, n, d)
    dB = eng.randinit_dev(7, n, m)
    dK0 = eng.synth_codebooks_dev(99, m, d)
    # codes that mean something for X: a short encode (the LSQR problem of a training iteration)
    dB = eng.encode_icm_dev(dX, dB, dK0, m, [2], 4, 4, True, seed=42)[0][0].contiguous()
    out = {"n": n, "d": d, "m": m}
    for rep in range(2):            # the second call is the one the profile's averages are dominated by; both are reported
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dK, iters = eng.update_codebooks_dev(dX, dB, m)
        torch.cuda.synchronize(); out["lsqr_ms_%d" % rep] = round((time.perf_counter() - t0) * 1e3, 3)
    out["lsqr_iterations"] = iters
    dcb = torch.linspace(0.0, float((dX[:4096].float() ** 2).sum(1).max().item()), 256, device=dX.device)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, dbn, nrm = eng.quantize_norms_dev(dB, dK, dcb, m)
        torch.cuda.synchronize(); out["norms_ms_%d" % rep] = round((time.perf_counter() - t0) * 1e3, 3)
print(json.dumps(out))
