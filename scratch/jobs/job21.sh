set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j21; mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_bench -- python3 bench.py --no-strong --no-f32 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python tools/prof_steps.py $O/tr_bench 13 75 > $O/step_breakdown.txt 2>&1
cp $(ls $O/tr_bench/*/*kernel_stats.csv | head -1) $O/bench_c2_kernel_stats.csv
rm -rf $O/tr_bench
rocprofv3 --kernel-trace --output-format csv -d $O/tr_c2 -- python3 tools/bench_step.py > $O/tr_c2.log 2>&1
python tools/prof_train_steps.py $O/tr_c2 5 80 $O/c2_step_sequence.txt > $O/c2_step_breakdown.txt 2>&1
rm -rf $O/tr_c2
rocprofv3 --kernel-trace --output-format csv -d $O/tr_small -- python3 tools/fov_stream.py --train-batches 150 --train-epochs 3 --graphed-train > $O/tr_small.log 2>&1
python tools/prof_train_steps.py $O/tr_small 100 400 $O/small_batch_step_sequence.txt > $O/small_batch_step_graphed.txt 2>&1
rm -rf $O/tr_small
DTYPE=f32 rocprofv3 --kernel-trace --output-format csv -d $O/tr_f32 -- python3 tools/bench_step.py > $O/tr_f32.log 2>&1
python tools/prof_train_steps.py $O/tr_f32 3 60 > $O/f32_step_breakdown.txt 2>&1
rm -rf $O/tr_f32
echo done
