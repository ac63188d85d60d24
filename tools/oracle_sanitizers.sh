#!/bin/bash
# tools/oracle_sanitizers.sh -- the whole CPU test suite against an AddressSanitizer + UBSan build of oracle/lsq_oracle.c.
# The checker is C with hand-written index arithmetic: this is its memory-safety / UB gate.  Prints pytest's summary; non-zero on any report.
set -e
cd "$(dirname "$0")/.."
make -C oracle asan > /dev/null
export LSQ_ORACLE_LIB=$(pwd)/oracle/liblsq_oracle_asan.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:handle_segv=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
