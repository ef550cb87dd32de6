#!/bin/bash
export BITS=1
for lib in default tools/ab_libs/dst_u2.so tools/ab_libs/fwd_u2.so; do
  if [ "$lib" = default ]; then unset SEGGER_AMD_LIB; else export SEGGER_AMD_LIB=$PWD/$lib; fi
  DROP=0.2 timeout -k 10 120 python3 tools/bench_gat.py 2>&1 | tail -1 || exit 1
done
