set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j2; mkdir -p $O
VARIANTS="old:ONE_LAUNCH_LOSS_HEAD=0;new:ONE_LAUNCH_LOSS_HEAD=1" ROUNDS=2 python tools/bench_step.py > $O/ab_c2.txt 2>&1
VARIANTS="old:ONE_LAUNCH_LOSS_HEAD=0;new:ONE_LAUNCH_LOSS_HEAD=1" ROUNDS=2 python tools/ab_graphed.py > $O/ab_graphed.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_small -- python3 tools/fov_stream.py --n-tx 10000000 --n-bd 100000 --train-batches 150 --train-epochs 3 --graphed-train > $O/tr_small.log 2>&1
python tools/prof_train_steps.py $O/tr_small 100 400 $O/small_seq.txt > $O/small_breakdown.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr_c2 -- python3 tools/bench_step.py > $O/tr_c2.log 2>&1
python tools/prof_train_steps.py $O/tr_c2 5 80 $O/c2_seq.txt > $O/c2_breakdown.txt 2>&1
rm -rf $O/tr_small $O/tr_c2
echo done
