#!/bin/bash
# bounding builds of linear_f32_split_kernel<384> (dX of the 128 -> 384 projections, fp32 storage) on one box
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-base onemfma nostore nosplit skeleton}; do
  [ $v = base ] && lib= || lib=$PWD/tools/ab_libs/fs_$v.so
  SEGGER_AMD_LIB=$lib N=${N:-1000000} ONLY=384 GATE_ONLY=1 python tools/bench_f32_split.py 2>&1 | grep "gate epilogue" | sed "s/^/$v: /"
done
