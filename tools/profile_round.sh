#!/bin/bash
# tools/profile_round.sh TAG [extra bench.py args]  -- run ON THE GPU BOX (through gpurun), from the repo root.
# Collects the rocprofv3 evidence DESIGN.md / bench.py cite, into gpurun_out/prof_TAG/ :
#   stats   : --kernel-trace --stats                      (per-kernel time)
#   fetch   : --pmc FETCH_SIZE            + kernel trace  (separate passes: TCC has 4 counter slots)
#   write   : --pmc WRITE_SIZE            + kernel trace
#   tcc     : --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
#   sq      : --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
# never --pmc together with sys/runtime/hip/hsa tracing.  Then: python tools/pmc_summary.py gpurun_out/prof_TAG TAG
set -u
TAG=${1:-rXX}; shift || true
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LSQ_BENCH_TRAIN_CACHE=1      # --trained-codebooks: trained once per box and build, so that the profiled processes hold no training launches
B="python $R/bench.py --no-cpu-baseline --no-extra-legs --no-sample-parity $*"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $B --steps 1 --warmup 1 > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench --output-format csv -- $B --steps 1 --warmup 0 > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench --output-format csv -- $B --steps 1 --warmup 0 > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d "$OUT/tcc" -o bench --output-format csv -- $B --steps 1 --warmup 0 > "$OUT/tcc.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d "$OUT/sq" -o bench --output-format csv -- $B --steps 1 --warmup 0 > "$OUT/sq.log" 2>&1
cd "$R"
python tools/pmc_summary.py "$OUT" "$TAG" $*
cp "$OUT/${TAG}_pmc_per_kernel.json" profiles/        # bench.py reads its `traffic` field from profiles/<PMC_FILE>: same build, same box
timeout 600 python bench.py $* > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python tools/pmc_summary.py "$OUT" "$TAG" $*
# keep the merge-back small: the raw per-dispatch csv files are large
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*_counter_collection.csv" -size +8M -delete
ls -la "$OUT"
