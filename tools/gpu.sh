#!/bin/bash
# tools/gpu.sh '<command for the GPU box>' [timeout] -- builds first and refuses to spend GPU time on a broken build
set -e
cd "$(dirname "$0")/.."
make -C local-search-quantization_amd/csrc -j8 2>&1 | grep -E "error|Error" -A8 && { echo "BUILD FAILED"; exit 1; } || true
make -C local-search-quantization_amd/csrc -j8 > /dev/null
exec /usr/local/graft/bin/gpurun --timeout "${2:-1800}" -- "$1"
