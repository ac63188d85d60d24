"""tools/sweep_breakdown.py OUT_DIR [bench args] -- run ON THE GPU BOX.  Per-ICM-sweep breakdown of the walk kernel at one
launch per node update (schedule 3): time, HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes) and the number
of node updates actually recomputed, per sweep 1..icmiter (averaged over ILS iterations and nodes).

Answers "where do the bytes beyond the algorithmic 1033 B per recomputed node update come from" (VERDICT r1 weak #4):
dense sweeps vs sparse sweeps.  Writes OUT_DIR/sweep_breakdown.json and prints it.
"""
import csv
import glob
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def walk_rows(path, value_col, name_col):
    rows = []
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if ("icm_walkq_kernel" if os.environ.get("LSQ_SB_SCHEDULE", "3") == "6" else "icm_walk_kernel") in row[name_col]:
                rows.append(row)
    return rows


def main():
    out_dir = os.path.abspath(sys.argv[1])
    extra = sys.argv[2:]
    os.makedirs(out_dir, exist_ok=True)
    m, J, ils = 8, 4, 16
    for i, a in enumerate(extra):
        if a == "--codebooks":
            m = int(extra[i + 1])
        if a == "--icmiter":
            J = int(extra[i + 1])
        if a == "--ils":
            ils = int(extra[i + 1])
    sched = os.environ.get("LSQ_SB_SCHEDULE", "3")          # 3: f32 walk, one launch per node; 6: filtered walk with option per_node
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-legs", "--schedule", sched, "--steps", "1", "--warmup", "0"] + extra
    if sched == "6":
        bench += ["--option", "per_node=1"]
    env = dict(os.environ, TMPDIR="/tmp")
    res = {"bench_args": " ".join(extra), "per_sweep": []}
    passes = {"time": ["--kernel-trace"], "fetch": ["--pmc", "FETCH_SIZE", "--kernel-trace"], "write": ["--pmc", "WRITE_SIZE", "--kernel-trace"]}
    data = {}
    for name, flags in passes.items():
        d = os.path.join(out_dir, name)
        subprocess.run(["rocprofv3"] + flags + ["-d", d, "-o", "sb", "--output-format", "csv", "--"] + bench, cwd="/tmp", env=env,
                       stdout=open(os.path.join(out_dir, name + ".log"), "w"), stderr=subprocess.STDOUT, timeout=900)
        if name == "time":
            f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
            rows = walk_rows(f, None, "Kernel_Name")
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            data[name] = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        else:
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
            rows = walk_rows(f, None, "Kernel_Name")
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            data[name] = [float(r["Counter_Value"]) * 1024.0 for r in rows]      # KiB -> bytes
    # node updates recomputed per position of the node sequence (device trace counters, lsq_get_walk_trace)
    lsq = importlib.import_module("local-search-quantization_amd")
    n = 1_000_000
    for i, a in enumerate(extra):
        if a == "--vectors":
            n = int(extra[i + 1])
    with lsq.Engine(0, profile=True, schedule=int(sched)) as eng:
        dX = eng.synth_data_u8_dev(1234, n, 128)
        dB0 = eng.randinit_dev(7, n, m)
        dK = eng.synth_codebooks_dev(4321, m, 128)
        if "--trained-codebooks" in extra:      # the codebooks bench.py trains (bench.py: train_codebooks)
            ns = min(n, 100_000)      # the same training bench.py runs (train_codebooks): 0.1 s on the device, no file in between
            with lsq.Engine(0) as e2:
                dK, _, _, _, _ = lsq.train_lsq_dev(dX[:ns].contiguous(), m, 256, dB0[:ns].contiguous(), 8, 4, J, True, 4, seed=42, engine=e2, norm_codebook=False)
        eng.reset_timings()
        eng.encode_icm_dev(dX, dB0, dK, m, [ils], J, 4, True, seed=42)
        res["node_updates_total"] = eng.timings()["icm_node_updates"]
        trace = eng.walk_trace(64)
    res["active_fraction_per_position"] = [float(trace[q]) / (n * ils) for q in range(min(64, J * m))]
    nn = J * m
    for sw in range(J):
        idx = [i for i in range(len(data["time"])) if (i % nn) // m == sw]
        t = sum(data["time"][i] for i in idx) / max(len(idx), 1)
        fe = sum(data["fetch"][i] for i in idx) / max(len(idx), 1) if len(data["fetch"]) == len(data["time"]) else None
        wr = sum(data["write"][i] for i in idx) / max(len(idx), 1) if len(data["write"]) == len(data["time"]) else None
        act = sum(res["active_fraction_per_position"][sw * m:(sw + 1) * m]) / m if (sw + 1) * m <= 64 else None
        res["per_sweep"].append({"sweep": sw + 1, "launches": len(idx), "avg_us": t, "active_fraction": act,
                                 "algorithmic_bytes": (act * n * (521.0 if sched == "6" else 1033.0)) if act is not None else None, "fetch_bytes_raw": fe, "write_bytes": wr,
                                 "traffic_bytes_fetch_x2_plus_write": (2 * fe + wr) if fe is not None and wr is not None else None})
    res["launches_seen"] = {k: len(v) for k, v in data.items()}
    json.dump(res, open(os.path.join(out_dir, "sweep_breakdown.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
