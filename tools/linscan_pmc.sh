#!/bin/bash
# tools/linscan_pmc.sh TAG -- run ON THE GPU BOX (through gpurun): the PMC passes of tools/profile_round.sh for the device ADC scan
# (tools/linscan_bench.py, SIFT1M-shaped search).  -> gpurun_out/prof_TAG/TAG_pmc_per_kernel.json (pmc_summary.py), copied to profiles/ by hand.
set -u
TAG=${1:-rXXlinscan}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/linscan_bench.py 1000000 10000 128 8 1000"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $B > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench --output-format csv -- $B > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench --output-format csv -- $B > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d "$OUT/tcc" -o bench --output-format csv -- $B > "$OUT/tcc.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d "$OUT/sq" -o bench --output-format csv -- $B > "$OUT/sq.log" 2>&1
cd "$R"
python tools/pmc_summary.py "$OUT" "$TAG" tools/linscan_bench.py 1000000 10000 128 8 1000
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*_counter_collection.csv" -size +8M -delete
python - "$OUT/${TAG}_pmc_per_kernel.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
for k,v in j.items():
    if k.startswith("adc_") : print(k, json.dumps(v)[:1200])
PY
