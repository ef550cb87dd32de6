cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j23
python -m pytest tests -m gpu -x -q > gpurun_out/j23/gputest.txt 2>&1 || { tail -40 gpurun_out/j23/gputest.txt; exit 1; }
tail -2 gpurun_out/j23/gputest.txt
VARIANTS="before:DEFER_LOSS_FINISH=0;after:DEFER_LOSS_FINISH=1" ROUNDS=2 python tools/ab_graphed.py 2>&1 | grep round
