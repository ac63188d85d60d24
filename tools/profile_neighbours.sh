#!/bin/bash
# tools/profile_neighbours.sh TAG [n d m] -- run ON THE GPU BOX (through gpurun), from the repo root: rocprofv3 passes over tools/neighbours_run.py
# (stats, FETCH_SIZE, WRITE_SIZE: separate passes, never with sys/runtime tracing), then tools/neighbours_roofline.py -> gpurun_out/prof_TAG/
set -u
TAG=${1:-rXXnb}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/neighbours_run.py $*"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o nb --output-format csv -- $B > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o nb --output-format csv -- $B > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o nb --output-format csv -- $B > "$OUT/write.log" 2>&1
cd "$R"
timeout 300 $B > "$OUT/run_line.json" 2> "$OUT/run.err"
python tools/pmc_summary.py "$OUT" "$TAG" $*
python tools/neighbours_roofline.py "$OUT" "$TAG" $*
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*_counter_collection.csv" -size +8M -delete
ls -la "$OUT"
