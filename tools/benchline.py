"""Condense bench.py's JSON line (stdin) to one short line: for same-box A/B runs.  Usage: python bench.py ... | python tools/benchline.py TAG"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    o = json.loads(line)
    tb = o["time_breakdown_ms_per_step"]
    print("%-14s %10.0f vec/s  %8.2f ms/step  icm %.2f  unary %.2f  cost %.2f  perturb %.2f  obj %.6f  frac %.3f" % (
        tag, o["value"], o["ms_per_step"], tb["icm_ms"], tb["unaries_ms"], tb["cost_ms"], tb["perturb_ms"], o["objective"], o["roofline"]["frac"]))
