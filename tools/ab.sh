#!/bin/bash
# tools/ab.sh libA.so libB.so [bench args] -- run ON THE GPU BOX: alternate two builds of the library on the same box, print the bench line's timings
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in $A $B; do
    LSQ_LIB_PATH=$L python bench.py --no-cpu-baseline --no-extra-legs --no-sample-parity --steps 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['time_breakdown_ms_per_step']
print('$L', 'step %.2f ms  %.2f M/s | unaries %.2f icm %.2f cost %.2f tables %.2f' % (d['ms_per_step'], d['value']/1e6, t['unaries_ms'], t['icm_ms'], t['cost_ms'], t['tables_ms']))"
  done
done
