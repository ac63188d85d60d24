#!/bin/bash
# tools/final_round.sh TAG -- run ON THE GPU BOX (through gpurun), from the repo root: everything DESIGN.md / BASELINE.md quote for one build.
#   gpurun_out/final_TAG/  <- profile_round.sh (stats + PMC + bench line), the other BASELINE configs on one GPU, the shape sweep,
#                             the per-sweep breakdown, the in-kernel phase / per-node timings (tuning build), the LDS micro-benchmark
set -u
TAG=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out/final_$TAG
mkdir -p "$O"
bash tools/profile_round.sh "$TAG" > "$O/profile_round.log" 2>&1
cp gpurun_out/prof_$TAG/${TAG}_bench_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_pmc_per_kernel.json gpurun_out/prof_$TAG/${TAG}_bench_line.json "$O/" 2>/dev/null
# PMC passes for the other BASELINE shapes too: cfg3 (m = 16) and cfg4's per-GPU share
bash tools/profile_round.sh "${TAG}cfg3" --codebooks 16 > "$O/profile_cfg3.log" 2>&1
cp gpurun_out/prof_${TAG}cfg3/${TAG}cfg3_pmc_per_kernel.json gpurun_out/prof_${TAG}cfg3/${TAG}cfg3_bench_kernel_stats.csv "$O/" 2>/dev/null
bash tools/profile_round.sh "${TAG}cfg4" --scaling strong --total 125000 --dim 960 > "$O/profile_cfg4.log" 2>&1
cp gpurun_out/prof_${TAG}cfg4/${TAG}cfg4_pmc_per_kernel.json gpurun_out/prof_${TAG}cfg4/${TAG}cfg4_bench_kernel_stats.csv "$O/" 2>/dev/null
python tools/scaling_shares.py "$O/${TAG}_scaling_shares.json" > "$O/scaling.log" 2>&1
B="python bench.py --no-cpu-baseline --no-workloads --steps 5"
$B --codebooks 16 > "$O/${TAG}_bench_cfg3.json" 2> "$O/cfg3.err"
$B --scaling strong --total 125000 --dim 960 > "$O/${TAG}_bench_cfg4_share.json" 2> "$O/cfg4.err"
$B --vectors 12500000 --steps 2 --no-extra-legs > "$O/${TAG}_bench_cfg5_share.json" 2> "$O/cfg5.err"      # sample_parity stays on (two 256-row slices)
LSQ_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 \
    --scaling strong --total 250000 --dim 960 --steps 3 --no-cpu-baseline > "$O/${TAG}_bench_cfg4_2rank_1gpu_gloo.json" 2> "$O/gloo.err"
python tools/sched_cmp.py > "$O/sched_cmp.log" 2>&1 && cp gpurun_out/r02g/sched_cmp.json "$O/${TAG}_schedule6_vs_4_shapes.json"
LSQ_SB_SCHEDULE=6 python tools/sweep_breakdown.py "$O/sb6" > "$O/sb6.log" 2>&1 && cp "$O/sb6/sweep_breakdown.json" "$O/${TAG}_sweep_breakdown_schedule6_per_node.json"
{ echo "== one launch per node update (phases of block 0)"; python tools/walkq_phases.py 1000000 1 0 2>&1 | tail -66;
  echo "== production schedule (one launch per ILS iteration): active count : microseconds per node update of block 0"; python tools/walkq_phases.py 1000000 0 160 2>&1 | tail -4;
  echo "== 125 000 vectors, production schedule"; python tools/walkq_phases.py 125000 0 160 2>&1 | tail -4; } > "$O/${TAG}_walkq_phases.txt"
# the device ADC scan: SIFT1M-shaped search at m = 8 / 16 and three neighbour counts, then its kernel times
{ python tools/linscan_bench.py 1000000 10000 128 8 1000; python tools/linscan_bench.py 1000000 10000 128 8 100; python tools/linscan_bench.py 1000000 10000 128 8 1;
  python tools/linscan_bench.py 1000000 10000 128 16 1000; python tools/linscan_bench.py 1000000 10000 128 8 10000; python tools/linscan_bench.py 1000000 1000 960 8 1000; } > "$O/${TAG}_linscan.jsonl" 2> "$O/linscan.err"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$O/linscan_stats" -o k --output-format csv -- python "$R/tools/linscan_bench.py" > "$O/linscan_stats.log" 2>&1 )
cp "$(find "$O/linscan_stats" -name '*kernel_stats.csv' | head -1)" "$O/${TAG}_linscan_kernel_stats.csv" 2>/dev/null
# the codebook update on the device: host LSQR vs device LSQR on three shapes
{ python tools/lsqr_bench.py 100000 128 8; python tools/lsqr_bench.py 1000000 128 8; python tools/lsqr_bench.py 100000 960 8; } > "$O/${TAG}_lsqr.jsonl" 2> "$O/lsqr.err"
[ -x tools/bin/ubench_lds ] && tools/bin/ubench_lds > "$O/ubench_lds_${TAG}.txt" 2>&1
# round 5's micro-benchmarks: bare LDS read rates (the guide's figures reproduced), the cost pass's memory traffic replayed + the energy-identity filter's
# scattered reads, the unary GEMM's structure variants, HBM write patterns, H2D rates; the host-buffer entry point with / without the upload pipeline
[ -x tools/bin/ubench_lds2 ] && tools/bin/ubench_lds2 > "$O/${TAG}_ubench_lds2.txt" 2>&1
[ -x tools/bin/ubench_cost ] && { tools/bin/ubench_cost 1000000 128; tools/bin/ubench_cost 125000 960; } > "$O/${TAG}_ubench_cost.txt" 2>&1
[ -x tools/bin/ubench_gemm ] && tools/bin/ubench_gemm 1000064 128 > "$O/${TAG}_ubench_gemm.txt" 2>&1
[ -x tools/bin/ubench_write ] && tools/bin/ubench_write > "$O/${TAG}_ubench_write.txt" 2>&1
[ -x tools/bin/ubench_h2d ] && tools/bin/ubench_h2d > "$O/${TAG}_ubench_h2d.txt" 2>&1
# round 6: the sparse-read calibration of FETCH_SIZE (timings + one PMC pass), table staging by registers / LDS-DMA / direct gathers, the initialisers' device kernels,
# the first host-buffer call of a fresh process
[ -x tools/bin/ubench_sparse ] && { tools/bin/ubench_sparse 5 > "$O/ubench_sparse_timing.txt" 2>&1;
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/ubench_sparse_pmc" -o sp --output-format csv -- "$R/tools/bin/ubench_sparse" 1 > "$O/ubench_sparse_pmc.log" 2>&1 ); }
[ -x tools/bin/ubench_stage ] && timeout 300 tools/bin/ubench_stage 32 > "$O/${TAG}_ubench_stage_raw.txt" 2>&1
{ python tools/init_bench.py 100000 128 8; python tools/init_bench.py 1000000 128 8; python tools/init_bench.py 100000 960 8; python tools/init_bench.py 100000 128 16; } 2>&1 | grep "^{" > "$O/${TAG}_init_kernels.jsonl"
{ echo "== fresh process, cfg2 (10^6 x 128)"; python tools/first_call.py 1000000 128 0 1; echo "== fresh process, cfg4 share (125 000 x 960)"; python tools/first_call.py 125000 960 0 1; } 2>&1 | grep -v amdgpu.ids > "$O/${TAG}_first_call.txt"
{ python tools/e2e_probe.py 1000000 128; python tools/e2e_probe.py 125000 960; } 2>&1 | grep -v amdgpu.ids > "$O/${TAG}_end_to_end_probe.txt"
python tools/small_call.py 2>&1 | grep -v amdgpu.ids > "$O/${TAG}_small_call.txt"
R2=$(pwd); ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d "$O/timeline" -o bench --output-format csv -- python "$R2/bench.py" --no-cpu-baseline --no-extra-legs --no-sample-parity --steps 1 --warmup 1 > "$O/timeline.log" 2>&1 )
F=$(find "$O/timeline" -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && { N=$(python -c "import csv;print(len(list(csv.DictReader(open('$F')))))"); python tools/trace_timeline.py "$F" $((N-75)) 75 > "$O/${TAG}_timeline_one_step.txt"; python tools/trace_gaps.py "$F" > "$O/${TAG}_trace_gaps.txt"; }
find "$O" -name "*.csv" -size +4M -delete
ls -la "$O"
