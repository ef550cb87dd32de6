#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/s9; mkdir -p $OUT
python3 -m pytest tests/test_gpu_model.py tests/test_gpu_train_graph.py tests/test_gpu_golden.py tests/test_gpu_fov.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -6 $OUT/pytest.log; [ $rc -eq 0 ] || exit $rc
DTYPE=f32 VARIANTS="old:POS_POLY_F32=0;poly:POS_POLY_F32=1" ROUNDS=2 STEPS=8 python3 tools/bench_step.py > $OUT/step_f32_ab.log 2>&1; grep round $OUT/step_f32_ab.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/small32 -- python3 $R/tools/fov_stream.py --n-tx 10000000 --n-bd 100000 --train-batches 150 --train-epochs 3 --graphed-train --train-dtype f32 --score-dtypes f32 > $OUT/small32.log 2>&1 || { tail -20 $OUT/small32.log; exit 1; }
python3 $R/tools/prof_train_steps.py $OUT/small32 100 30 $OUT/r06_small_batch_step_f32_sequence.txt > $OUT/r06_small_batch_step_f32_graphed.txt
rm -rf $OUT/small32
cat $OUT/r06_small_batch_step_f32_graphed.txt
