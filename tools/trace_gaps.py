"""tools/trace_gaps.py KERNEL_TRACE_CSV -- per-kernel durations and the idle gaps between consecutive kernels of a rocprofv3
--kernel-trace run (same stream): where the time of a launch-heavy schedule goes."""
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", n); return (m.group(1) + (m.group(2) or "")) if m else n[:50]
dur = defaultdict(list); gap = defaultdict(list)
prev_end = None; prev_name = None
for r in rows:
    s, e, nm = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    dur[nm].append((e - s) / 1e3)
    if prev_end is not None: gap[(prev_name, nm)].append((s - prev_end) / 1e3)
    prev_end, prev_name = e, nm
print("%-48s %7s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "total_ms"))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-48s %7d %10.2f %10.2f %10.3f" % (k[:48], len(v), sum(v) / len(v), min(v), sum(v) / 1e3))
print("\n%-70s %7s %10s %10s" % ("gap after -> before", "count", "avg_us", "total_ms"))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-70s %7d %10.2f %10.3f" % ((k[0][:32] + " -> " + k[1][:32]), len(v), sum(v) / len(v), sum(v) / 1e3))
tot = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("\nspan %.3f ms, kernels %.3f ms, gaps %.3f ms" % (tot / 1e6, sum(sum(v) for v in dur.values()) / 1e3, sum(sum(v) for v in gap.values()) / 1e3))

# per-position averages of the walk / apply kernels (position in the ILS iteration's launch sequence), when asked
if len(sys.argv) > 2:
    period = int(sys.argv[2])
    for key in ("icm_walk_kernel", "icm_apply_scan_kernel"):
        seqs = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])) for r in rows if key in r["Kernel_Name"] and ("true" in short(r["Kernel_Name"]) or key != "icm_walk_kernel")]
        if not seqs: continue
        per = period + (1 if key == "icm_apply_scan_kernel" else 0)
        print("\n%s per position (period %d, %d launches):" % (key, per, len(seqs)))
        for pos in range(per):
            v = [d for i, (d, _) in enumerate(seqs) if i % per == pos]
            if v: print("  pos %2d  avg %8.2f us  min %8.2f  (n=%d)" % (pos, sum(v) / len(v), min(v), len(v)))
