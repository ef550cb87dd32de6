#!/bin/bash
# in-step per-kernel durations, natural vs window-64 by-destination order
R=$PWD; OUT=$R/gpurun_out/s3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export ROUNDS=1 STEPS=8
export VARIANTS="natural:ROW_ORDER=0"
rocprofv3 --kernel-trace --output-format csv -d $OUT/nat -- python3 $R/tools/bench_step.py > $OUT/nat.log 2>&1 || exit 1
python3 $R/tools/prof_train_steps.py $OUT/nat 5 24 > $OUT/nat_breakdown.txt
export VARIANTS="win64:ROW_ORDER=64"
rocprofv3 --kernel-trace --output-format csv -d $OUT/w64 -- python3 $R/tools/bench_step.py > $OUT/w64.log 2>&1 || exit 1
python3 $R/tools/prof_train_steps.py $OUT/w64 5 24 > $OUT/w64_breakdown.txt
rm -rf $OUT/nat $OUT/w64
cat $OUT/nat_breakdown.txt; cat $OUT/w64_breakdown.txt
