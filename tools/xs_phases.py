"""tools/xs_phases.py [n] [m] [d] -- run ON THE GPU BOX with the tuning library.  Per-task clock stamps (wall_clock64, 100 MHz) of ONE block
(group 0, slice 0) of the LAST icm_xs_kernel launch (schedule 7, csrc/lsq_icmx.hip): when its walkers waited for a list / walked / signalled, when its
listers waited for their dependencies / scanned, when its merger saw the walk done / every CU's partial keys / finished.  One line per task + sums."""
import ctypes as C, importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
lsq = importlib.import_module("local-search-quantization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
L = lsq._lib.load(tuning=True)
buf = torch.zeros((1024, 16), dtype=torch.int64, device="cuda")
L.lsq_tuning_set_xs_debug.restype = C.c_int
L.lsq_tuning_set_xs_debug.argtypes = [C.c_void_p]
assert L.lsq_tuning_set_xs_debug(buf.data_ptr()) == 0
with lsq.Engine(0, schedule=7, tuning=True) as eng:
    eng.set_option("q16_min", 0)
    eng.set_option("xs_min", 0)
    dX = eng.synth_data_u8_dev(1234, n, d); dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, d)
    eng.encode_icm_dev(dX, dB0, dK, m, [3], 4, 4, True, seed=42)
    print(eng.timings())
torch.cuda.synchronize()
t = buf.cpu().numpy()
nt = int((t[:, 1] != 0).sum())
print("tasks recorded", nt, "(the launch's third ILS iteration)")
t0 = min(int(x) for x in t[:nt, [0, 5]].reshape(-1) if x)
us = lambda a, b: (int(b) - int(a)) / 100.0 if a and b else float("nan")
tot = dict(wait_list=0.0, walk=0.0, stage=0.0, ldep=0.0, lscan=0.0, mwait=0.0, merge=0.0)
Q = nt // (4 * m) if nt >= 4 * m else 1
for k in range(nt):
    r = t[k]
    row = dict(at=us(t0, r[0]), wait_list=us(r[0], r[1]), walk=us(r[1], r[2]), stage=us(r[3], r[4]) if r[3] else 0.0,
               ldep=us(r[5], r[6]), lscan=us(r[6], r[7]), nact=int(r[8]), wdone_to_all=us(r[9], r[10]), merge=us(r[10], r[11]), publish=us(r[11], r[12]),
               list_ready_at=us(t0, r[7]), merged_at=us(t0, r[12]), namb=int(r[13]))
    for key in ("wait_list", "walk", "stage", "ldep", "lscan", "merge"):
        if row[key] == row[key]:
            tot[key] += row[key]
    tot["mwait"] += row["wdone_to_all"] if row["wdone_to_all"] == row["wdone_to_all"] else 0.0
    if k < 6 * Q or k % (Q * m) < Q or k >= nt - 2 * Q:
        print("task %3d node %2d q %d at %7.1f | walker: wait list %5.1f walk %6.1f stage %4.1f | lister: dep %5.1f scan %5.1f (ready at %7.1f) nact %5d | "
              "merger: all CUs +%5.1f merge %5.1f publish %4.1f (at %7.1f) amb %d" %
              (k, k // Q, k % Q, row["at"], row["wait_list"], row["walk"], row["stage"], row["ldep"], row["lscan"], row["list_ready_at"], row["nact"],
               row["wdone_to_all"], row["merge"], row["publish"], row["merged_at"], row["namb"]))
end = max(int(x) for x in t[:nt, [2, 12]].reshape(-1))
print("launch span of this block %.1f us; sums: %s" % ((end - t0) / 100.0, {k: round(v, 1) for k, v in tot.items()}))
# per sweep
for sw in range(4):
    ks = range(sw * m * Q, min((sw + 1) * m * Q, nt))
    if len(ks) == 0:
        continue
    a = int(t[ks[0], 0]); b = int(t[ks[-1], 2])
    print("sweep %d: %.1f us  (%.1f per node)  active %d" % (sw, (b - a) / 100.0, (b - a) / 100.0 / m, int(t[list(ks), 8].sum())))
