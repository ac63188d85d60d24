cd /tmp && export TMPDIR=/tmp
R=/root/repo
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32"; do
timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/mfma_pmc -o x --output-format csv -- python $R/tools/knob_run.py 1000000 128 8 ils=1 steps=1 > $R/gpurun_out/mfma_pmc.log 2>&1
python - <<'PY'
import csv,glob,collections
for f in glob.glob('/root/repo/gpurun_out/mfma_pmc/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in agg.items():
        if 'chain_gemm' in k: print(k, dict(v))
import shutil; shutil.rmtree('/root/repo/gpurun_out/mfma_pmc', ignore_errors=True)
PY
done
