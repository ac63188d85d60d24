for sh in "1000000 128 8" "125000 960 8" "1000000 128 16"; do
  for rep in 1 2 3; do
    LSQ_LIB_PATH=tools/bin/old.so python tools/knob_run.py $sh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['n'], d['d'], d['m'], 'ms', d['ms'], 'icm', d['icm_ms'], 'codes', d['codes_sum'])"
    python tools/knob_run.py $sh 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['n'], d['d'], d['m'], 'ms', d['ms'], 'icm', d['icm_ms'], 'codes', d['codes_sum'])"
  done
done
