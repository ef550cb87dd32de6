set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -40 $O/gputest.txt; exit 1; }
tail -3 $O/gputest.txt
VARIANTS="old:ONE_LAUNCH_LOSS_HEAD=0;grouped:ONE_LAUNCH_LOSS_HEAD=1;chains:LOSS_HEAD_GROUPED_MIN_ROWS=1000000000" ROUNDS=2 python tools/bench_step.py > $O/ab_c2.txt 2>&1
grep round $O/ab_c2.txt
VARIANTS="nopair:LINEAR_PAIR=0;pair:LINEAR_PAIR=1" ROUNDS=2 python tools/ab_graphed.py > $O/ab_graphed.txt 2>&1
grep round $O/ab_graphed.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tr_c2 -- python3 tools/bench_step.py > $O/tr_c2.log 2>&1
python tools/prof_train_steps.py $O/tr_c2 5 80 $O/c2_seq.txt > $O/c2_breakdown.txt 2>&1
rm -rf $O/tr_c2
head -12 $O/c2_breakdown.txt
