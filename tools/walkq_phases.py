"""tools/walkq_phases.py [n] -- run ON THE GPU BOX with the tuning library.  Phase timestamps (wall_clock64, 100 MHz) of block 0 of
every icm_walkq_kernel launch (option per_node = 1: one launch per node update): compaction / per-slice walk / decide / redo."""
import ctypes as C, importlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
lsq = importlib.import_module("local-search-quantization_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
per_node = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = int(sys.argv[4]) if len(sys.argv) > 4 else 8
L = lsq._lib.load(tuning=True)
W = 24 + 2 * 65
buf = torch.zeros((4096, W), dtype=torch.int64, device="cuda")
L.lsq_tuning_set_walkq_debug.restype = C.c_int
L.lsq_tuning_set_walkq_debug.argtypes = [C.c_void_p]
assert L.lsq_tuning_set_walkq_debug(buf.data_ptr()) == 0
with lsq.Engine(0, schedule=6, tuning=True) as eng:
    eng.set_option("per_node", per_node)
    eng.set_option("q16_min", 0)
    eng.set_option("light", int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    dX = eng.synth_data_u8_dev(1234, n, 128); dB0 = eng.randinit_dev(7, n, m); dK = eng.synth_codebooks_dev(4321, m, 128)
    eng.encode_icm_dev(dX, dB0, dK, m, [2], 4, 4, True, seed=42)
torch.cuda.synchronize()
t = buf.cpu().numpy()
rows = t[(t[:, 0] != 0)]
print("launches recorded", len(rows))
names = ["start->compact", "compact->walk", "slice0", "slice1", "slice2", "slice3", "slice4", "slice5", "slice6", "slice7->walk end", "decide", "redo"]
for li, r in enumerate(rows[:72]):
    if r[13] == 0:
        print(li, "no staged work (light or idle)", "nact", r[14]); continue
    ts = [r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13]]
    d = [(ts[i + 1] - ts[i]) / 100.0 for i in range(len(ts) - 1)]       # us at 100 MHz
    st = [int(r[k]) for k in range(16, 21)]
    rf = [(st[k + 1] - st[k]) / 100.0 if st[k] and st[k + 1] else -1 for k in range(4)]
    if st[0] and st[4]:
        rf[3] = (st[4] - st[0]) / 100.0          # whole q16_refine
        rf[0] = (st[0] - int(r[12])) / 100.0     # from the end of the decide phase to the function entry
    # round 6: compaction split (scan done / first barrier passed / list written) and slice 4 split (barrier -> LDS stores + barrier -> prefetch issued -> items done -> next barrier)
    cs = [(int(r[k]) - int(r[0])) / 100.0 if r[k] else -1 for k in (16, 17, 18)]
    s4 = [(int(r[k]) - int(r[7])) / 100.0 if r[k] else -1 for k in (20, 21, 22)]
    print("%3d nact %5d amb %3d total %7.1f us | compact %5.1f (scan @%4.1f barrier @%4.1f list @%4.1f) pre %5.1f | slices %s (slice 4: stores+barrier @%4.1f prefetch issued @%4.1f items @%4.1f) | decide %5.1f refine+ %5.1f" %
          (li, r[14], r[15], (ts[-1] - ts[0]) / 100.0, d[0], cs[0], cs[1], cs[2], d[1], " ".join("%5.1f" % x for x in d[2:11]), s4[0], s4[1], s4[2], d[11], d[12]))

if not per_node:
    # one launch per ILS iteration: block 0's clock at the start of every node update of its first pass (the production schedule)
    for li, r in enumerate(rows[:4]):
        cl = [int(x) for x in r[24:24 + 65]]
        na = [int(x) for x in r[24 + 65:24 + 65 + 64]]
        k = max(i for i in range(64) if cl[i]) + 1
        ends = cl[1:k] + [cl[64]]
        print("launch %d: %d nodes, total %.1f us" % (li, k, (cl[64] - cl[0]) / 100.0))
        print("   " + " ".join("%d:%.0f" % (na[i], (ends[i] - cl[i]) / 100.0) for i in range(k)))
        # the LAST node update of the launch in detail (stamps 0..23 keep the last node's phases): offsets from the node's start
        t0 = cl[k - 1]
        def at(kx): return (int(r[kx]) - t0) / 100.0 if r[kx] else -1.0
        print("   last node (nact %d): scan @%.1f barrier @%.1f list @%.1f compacted @%.1f keys reset @%.1f | slice starts %s | slice 4: stores+barrier +%.1f prefetch +%.1f items +%.1f | walk end @%.1f decide @%.1f refine @%.1f node end @%.1f" % (
              na[k - 1], at(16), at(17), at(18), at(1), at(2), " ".join("%.1f" % at(3 + q) for q in range(8)),
              (int(r[20]) - int(r[7])) / 100.0, (int(r[21]) - int(r[7])) / 100.0, (int(r[22]) - int(r[7])) / 100.0, at(11), at(12), at(13), (cl[64] - t0) / 100.0))
