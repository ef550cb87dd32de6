#!/bin/bash
# tools/prof_round.sh rNN : the per-round rocprofv3 kernel traces (run on the GPU box from the repo root), reduced into
# gpurun_out/prof_rNN/ -- copy what is to be judged into profiles/.
#   rNN_bench_c2_kernel_stats.csv, rNN_step_breakdown.txt   bench.py's C2 part (--stats)
#   rNN_c2_step_breakdown.txt, rNN_c2_step_sequence.txt     steady state of the C2 training step (tools/bench_step.py), bf16
#   rNN_f32_step_breakdown.txt                               the same at fp32 storage
#   rNN_small_batch_step_graphed.txt, _sequence.txt          the replayed 1M-edge step (tools/fov_stream.py --graphed-train)
set -x
tag=$1; R=$PWD; OUT=$R/gpurun_out/prof_$tag; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $R/bench.py --no-strong --no-f32 --no-cpu-baseline > $OUT/c2.log 2>&1 || exit 1
cp $(ls $OUT/c2/*/*_kernel_stats.csv | head -1) $OUT/${tag}_bench_c2_kernel_stats.csv
python3 $R/tools/prof_steps.py $OUT/c2 13 75 > $OUT/${tag}_step_breakdown.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/step -- python3 $R/tools/bench_step.py > $OUT/step.log 2>&1 || exit 1
python3 $R/tools/prof_train_steps.py $OUT/step 5 80 $OUT/${tag}_c2_step_sequence.txt > $OUT/${tag}_c2_step_breakdown.txt
export DTYPE=f32
rocprofv3 --kernel-trace --output-format csv -d $OUT/step32 -- python3 $R/tools/bench_step.py > $OUT/step32.log 2>&1 || exit 1
unset DTYPE
python3 $R/tools/prof_train_steps.py $OUT/step32 3 60 > $OUT/${tag}_f32_step_breakdown.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/small -- python3 $R/tools/fov_stream.py --train-batches 150 --train-epochs 3 --graphed-train > $OUT/small.log 2>&1 || exit 1
python3 $R/tools/prof_train_steps.py $OUT/small 100 400 $OUT/${tag}_small_batch_step_sequence.txt > $OUT/${tag}_small_batch_step_graphed.txt
rm -rf $OUT/c2 $OUT/step $OUT/step32 $OUT/small          # (the raw traces are hundreds of MB)
ls -la $OUT
