#!/bin/bash
# profile of the captured 1M-edge step at fp32 storage (and bf16 for the round's records)
R=$PWD; OUT=$R/gpurun_out/s7; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/small32 -- python3 $R/tools/fov_stream.py --n-tx 10000000 --n-bd 100000 --train-batches 150 --train-epochs 3 --graphed-train --train-dtype f32 --score-dtypes f32 > $OUT/small32.log 2>&1 || { tail -20 $OUT/small32.log; exit 1; }
python3 $R/tools/prof_train_steps.py $OUT/small32 100 60 $OUT/r06_small_batch_step_f32_sequence.txt > $OUT/r06_small_batch_step_f32_graphed.txt
rm -rf $OUT/small32
cat $OUT/r06_small_batch_step_f32_graphed.txt
