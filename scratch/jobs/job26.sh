cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j26
python -m pytest tests -m gpu -x -q > gpurun_out/j26/gputest.txt 2>&1 || { tail -40 gpurun_out/j26/gputest.txt; exit 1; }
tail -2 gpurun_out/j26/gputest.txt
python tools/host_phases.py 2>&1 | grep -v Warn | head -12
python - <<'P' 2>&1 | grep -v Warn
import os, sys, time, torch
sys.path.insert(0, '.')
from segger_amd import ops
import subprocess
P
for v in 0 1; do echo "NODE_REDUCTIONS=$v"; python - <<P 2>&1 | grep -v Warn | tail -2
import runpy, sys
sys.argv=['tools/host_profile.py']
import os
os.environ['NO_CPROFILE']='1'
from segger_amd import ops
ops.NODE_REDUCTIONS=bool($v)
runpy.run_path('tools/host_profile.py', run_name='__main__')
P
done
VARIANTS="off:NODE_REDUCTIONS=0;on:NODE_REDUCTIONS=1" ROUNDS=2 STEPS=15 python tools/bench_step.py 2>&1 | grep round
