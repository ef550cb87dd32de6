cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_linear.py -x -q -k "split" 2>&1 | tail -3
DTYPE=f32 VARIANTS="exact:F32_SPLIT=0;split:F32_SPLIT=1" ROUNDS=2 STEPS=8 python tools/bench_step.py 2>&1 | grep round | tee gpurun_out/f32_split_step.txt
