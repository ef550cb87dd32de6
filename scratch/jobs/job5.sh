set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j5; mkdir -p $O
python tools/bench_loss_head.py > $O/lh.txt 2>&1; cat $O/lh.txt | grep -v Warn
python -m pytest tests/test_gpu_linear.py tests/test_gpu_heads.py tests/test_gpu_train_graph.py tests/test_gpu_model.py -x -q > $O/gputest.txt 2>&1 || { tail -40 $O/gputest.txt; exit 1; }
tail -3 $O/gputest.txt
VARIANTS="nopair:LINEAR_PAIR=0,WGRAD_PAIR=0;fwdpair:WGRAD_PAIR=0;pair:WGRAD_PAIR=1" ROUNDS=2 python tools/ab_graphed.py > $O/ab_graphed.txt 2>&1
grep round $O/ab_graphed.txt
