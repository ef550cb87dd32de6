#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/s10; mkdir -p $OUT
python3 -m pytest tests/test_gpu_model.py -m gpu -x -q -k "polynomial or fp64 or oracle" > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log; [ $rc -eq 0 ] || exit $rc
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/small32 -- python3 $R/tools/fov_stream.py --n-tx 10000000 --n-bd 100000 --train-batches 150 --train-epochs 3 --graphed-train --train-dtype f32 --score-dtypes f32 > $OUT/small32.log 2>&1 || { tail -20 $OUT/small32.log; exit 1; }
python3 $R/tools/prof_train_steps.py $OUT/small32 100 40 $OUT/r06_small_batch_step_f32_sequence.txt > $OUT/r06_small_batch_step_f32_graphed.txt
rm -rf $OUT/small32
cat $OUT/r06_small_batch_step_f32_graphed.txt
