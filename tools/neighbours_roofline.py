"""tools/neighbours_roofline.py DIR TAG [n d m] -- HBM roofline of the two device neighbours of the path from the rocprofv3 passes of
tools/profile_neighbours.sh (DIR/TAG_pmc_per_kernel.json, written by tools/pmc_summary.py): per kernel the ALGORITHMIC bytes of one launch
(formulas below), the average launch time of the stats pass, achieved TB/s, the fraction of 8 TB/s, and the counter traffic
(2 x FETCH_SIZE + WRITE_SIZE, KiB as rocprofv3 reports them on gfx950) -> DIR/TAG_neighbours_roofline.json"""
import json, os, sys

d_, tag = sys.argv[1], sys.argv[2]
n, d, m = (int(x) for x in (sys.argv[3:6] + ["1000000", "128", "8"][len(sys.argv) - 3:]))
H, TB = 256, 64
pk = json.load(open(os.path.join(d_, "%s_pmc_per_kernel.json" % tag)))
cols = m * H
dy = (d + TB - 1) // TB                  # block columns of 64 systems: codes / sorted keys are read once per block column
alg = {
    # u = S v - alpha u: U read + written, the codes once per block column; the V rows it gathers (m h d floats = 1 MiB at d = 128) stay in L2
    "lsqr_u_update": 2 * n * d * 4 + n * m * dy,
    # v = S'u - beta v: every row of U once per codebook through the sorted (column, row) keys, V read + written
    "lsqr_v_update": m * n * d * 4 + m * n * 8 * dy + 2 * cols * d * 4,
    "lsqr_xw_update": 5 * cols * d * 4,
    "lsqr_make_keys": n * m + n * m * 8,
    # norm quantisation: the codes in, index + quantised norm + norm out; the m codeword rows per vector (m d floats) are L2 gathers, not HBM
    "quantize_norms_kernel": n * m + n * (1 + 4 + 4),
}
gather = {"quantize_norms_kernel": n * m * d * 4, "lsqr_u_update": n * m * d * 4}      # on-chip (L2) gather bytes of one launch
rows = {}
for name, v in pk.items():
    key = next((k for k in alg if name.startswith(k)), None)
    if key is None and not name.startswith("DeviceRadixSort") and "Onesweep" not in name and "Histogram" not in name:
        continue
    st = v.get("stats")
    if not st:
        continue
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    traffic = (2 * f["mean_per_dispatch"] * 1024 if f else 0) + (w["mean_per_dispatch"] * 1024 if w else 0)
    row = {"calls": st["calls"], "avg_us": round(st["avg_us"], 2), "total_ms": round(st["avg_us"] * st["calls"] / 1e3, 3),
           "traffic_bytes_per_launch": int(traffic) if (f or w) else None}
    if key:
        a = alg[key]
        row.update(algorithmic_bytes_per_launch=a, achieved_TBps=round(a / (st["avg_us"] * 1e-6) / 1e12, 3),
                   frac_of_8TBps=round(a / (st["avg_us"] * 1e-6) / 8e12, 3), traffic_over_algorithmic=round(traffic / a, 2) if (f or w) else None)
        if key in gather:
            row["l2_gather_bytes_per_launch"] = gather[key]
            row["l2_gather_TBps"] = round(gather[key] / (st["avg_us"] * 1e-6) / 1e12, 2)
    rows[name] = row
out = {"shape": {"n": n, "d": d, "m": m}, "kernels": rows, "_build": pk.get("_build")}
rl = os.path.join(d_, "run_line.json")
if os.path.exists(rl):
    ls = [l for l in open(rl) if l.startswith("{")]
    if ls:
        out["run"] = json.loads(ls[-1])
json.dump(out, open(os.path.join(d_, "%s_neighbours_roofline.json" % tag), "w"), indent=1, sort_keys=True)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-44s calls %3d avg %9.1f us total %8.3f ms  alg %s  frac %s  traffic/alg %s" % (k[:44], r["calls"], r["avg_us"], r["total_ms"],
          r.get("algorithmic_bytes_per_launch"), r.get("frac_of_8TBps"), r.get("traffic_over_algorithmic")))
