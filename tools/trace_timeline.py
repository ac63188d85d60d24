"""tools/trace_timeline.py KERNEL_TRACE_CSV [first] [count] -- the dispatches of a rocprofv3 --kernel-trace run in start order: offset, duration, idle gap before."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
def short(n):
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", n); return (m.group(1) + (m.group(2) or "")) if m else n[:50]
t0 = int(rows[first]["Start_Timestamp"]); pe = None
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  dur %9.1f  gap %7.1f  grid %8s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if pe is None else (s - pe) / 1e3, r.get("Grid_Size", "?"), short(r["Kernel_Name"])[:70]))
    pe = e
