set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/j6; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1 || { tail -60 $O/gputest.txt; exit 1; }
tail -3 $O/gputest.txt
python tools/bench_loss_head.py 2>&1 | grep -v Warn | tail -3
N=44000 NB=450 E=17000 python tools/bench_loss_head.py 2>&1 | grep -v Warn | tail -3
VARIANTS="sep:MERGED_DRAWS=0;merged:MERGED_DRAWS=1" ROUNDS=2 python tools/ab_graphed.py > $O/ab_graphed.txt 2>&1
grep round $O/ab_graphed.txt
VARIANTS="a:;b:" ROUNDS=1 STEPS=20 python tools/bench_step.py 2>&1 | grep round
