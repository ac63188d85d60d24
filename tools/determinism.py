"""tools/determinism.py -- run ON THE GPU BOX.  The same 10^6-vector encode several times per schedule: every run must return bit-identical codes
(any data race in the walk / refinement / fused perturbation would show up as a run-to-run difference), and schedules 6 and 4 must agree."""
import hashlib, importlib, sys
import numpy as np, torch
sys.path.insert(0, ".")
lsq = importlib.import_module("local-search-quantization_amd")
n, d, m = 1_000_000, 128, 8
ref = None
for sched, reps in ((6, 6), (4, 2)):
    with lsq.Engine(0, schedule=sched) as eng:
        dX = eng.synth_data_u8_dev(1234, n, d); dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, d)
        for r in range(reps):
            out, sums, _ = eng.encode_icm_dev(dX, dB0, dK, m, [4, 8], 4, 4, True, seed=42)
            torch.cuda.synchronize()
            h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
            print("schedule %d run %d: codes %s objective sums %r" % (sched, r, h, [float(x) for x in sums]), flush=True)
            if ref is None:
                ref = h
            assert h == ref, "codes differ between runs / schedules"
print("deterministic: every run returned the same codes")
