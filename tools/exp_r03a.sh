mkdir -p gpurun_out/r03a; O=gpurun_out/r03a/knobs.jsonl; : > $O
R="python tools/knob_run.py"
for shape in "125000 960 8" "250000 960 8" "500000 960 8" "1000000 960 8" "125000 128 8" "1000000 128 8" "1000000 128 16"; do
  $R $shape >> $O 2>>gpurun_out/r03a/err.log
  $R $shape tuning=1 >> $O 2>>gpurun_out/r03a/err.log
  LSQ_WALKQ_BPC=2 $R $shape tuning=1 >> $O 2>>gpurun_out/r03a/err.log
done
for lt in 64 256 400 600; do $R 125000 960 8 light=$lt >> $O 2>>gpurun_out/r03a/err.log; done
$R 125000 960 8 schedule=4 >> $O 2>>gpurun_out/r03a/err.log
$R 125000 960 8 q16_min=1000000000 >> $O 2>>gpurun_out/r03a/err.log
cat $O
