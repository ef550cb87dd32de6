set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j3; mkdir -p $O
python -m pytest tests/test_gpu_heads.py -x -q -k "loss_head" > $O/test_heads.txt 2>&1 || { tail -30 $O/test_heads.txt; exit 1; }
tail -3 $O/test_heads.txt
VARIANTS="old:ONE_LAUNCH_LOSS_HEAD=0;new:ONE_LAUNCH_LOSS_HEAD=1;chains:LOSS_HEAD_GROUPED_MIN_ROWS=1000000000" ROUNDS=2 python tools/bench_step.py > $O/ab_c2.txt 2>&1
grep round $O/ab_c2.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tr_c2 -- python3 tools/bench_step.py > $O/tr_c2.log 2>&1
python tools/prof_train_steps.py $O/tr_c2 5 80 $O/c2_seq.txt > $O/c2_breakdown.txt 2>&1
rm -rf $O/tr_c2
head -30 $O/c2_breakdown.txt
