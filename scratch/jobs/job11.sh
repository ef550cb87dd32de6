cd $GRAFT_REPO_ROOT
python tools/bench_f32_split.py 2>&1 | grep -v Warn | tee gpurun_out/f32_split.txt
