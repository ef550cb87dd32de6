#!/bin/bash
# A/B of library builds on the GATv2 micro-benchmark (C2 tx-neighbors-tx layer, bf16):
#   tools/ab.sh default tools/ab_libs/variant.so ...     (ORDER=0 switches the degree-balanced row order off)
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SEGGER_AMD_LIB; else export SEGGER_AMD_LIB=$PWD/$lib; fi
  DROP=0.0 timeout -k 10 120 python tools/bench_gat.py 2>&1 | tail -1 || exit 1
  DROP=0.2 timeout -k 10 120 python tools/bench_gat.py 2>&1 | tail -1 || exit 1
done
