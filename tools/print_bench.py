"""tools/print_bench.py [file ...] (or < bench output): one compact line per bench JSON line (value, ms/step, breakdown, filter counters)."""
import json
import sys

import fileinput

for l in fileinput.input():
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l)
        print("%.2f M/s  %.2f ms" % (d["value"] / 1e6, d["ms_per_step"]), {k: round(v, 2) for k, v in d.get("time_breakdown_ms_per_step", {}).items()},
              "frac %.3f" % d["roofline"]["frac"] if d.get("roofline") else "", d.get("filter"))
