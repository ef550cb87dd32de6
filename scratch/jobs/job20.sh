cd $GRAFT_REPO_ROOT
for r in 1 2; do for ms in 32 16 24 48; do
  echo "min_stages $ms: $(SEGGER_WGRAD_MIN_STAGES=$ms VARIANTS='x:' ROUNDS=1 python tools/ab_graphed.py 2>&1 | grep round)"
done; done | tee gpurun_out/wgrad_stages.txt
