cd $GRAFT_REPO_ROOT
python tools/host_phases.py 2>&1 | grep -v Warn | tee gpurun_out/host_phases.txt
