#!/bin/bash
# A/B of library builds on the GATv2 micro-benchmark: scratch/ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SEGGER_AMD_LIB; else export SEGGER_AMD_LIB=$PWD/$lib; fi
  DROP=0.0 timeout -k 10 120 python scratch/bench_gat.py 2>&1 | tail -1 || exit 1
  DROP=0.2 timeout -k 10 120 python scratch/bench_gat.py 2>&1 | tail -1 || exit 1
done
