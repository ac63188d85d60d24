"""tools/pmc_summary.py DIR TAG [bench args...] -- condense the rocprofv3 passes of tools/profile_round.sh into the files kept under profiles/:
   TAG_bench_kernel_stats.csv  (rocprofv3's own --stats summary, copied)
   TAG_pmc_per_kernel.json     (mean counter value per dispatch per kernel; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)
   TAG_bench_line.json         (bench.py's JSON line of the same build)"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {}
    for sub in ("fetch", "write", "tcc", "sq"):
        for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
            for k, cs in acc.items():
                for c, (tot, cnt) in cs.items():
                    out.setdefault(k, {})[c] = {"mean_per_dispatch": tot / cnt, "dispatches": cnt}
    # per-kernel average durations from the stats pass; for the walk kernel also the spread
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(d, "%s_bench_kernel_stats.csv" % tag))
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                out.setdefault(short(row["Name"]), {})["stats"] = {
                    "calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "pct": float(row["Percentage"]),
                    "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3}
    # which build the counters belong to: bench.py reports `traffic` only when this hash equals the hash of the library it loaded
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "local-search-quantization_amd", "liblsq_mi355x.so")
    out["_build"] = {"lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None,
                     "bench_args": " ".join(sys.argv[3:])}
    json.dump(out, open(os.path.join(d, "%s_pmc_per_kernel.json" % tag), "w"), indent=1, sort_keys=True)
    bl = os.path.join(d, "bench_line.json")
    if os.path.exists(bl):
        lines = [l for l in open(bl) if l.startswith("{")]
        if lines:
            open(os.path.join(d, "%s_bench_line.json" % tag), "w").write(lines[-1])
    print("wrote", sorted(os.listdir(d)))


if __name__ == "__main__":
    main()
