cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j16
SEGGER_BENCH_WATCHDOG=50 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --strong-n-tx 4000000 --strong-n-bd 40000 > gpurun_out/j16/bench2.json 2> gpurun_out/j16/bench2.err; echo "bench rc $?"
grep -n "File\|Thread\|bench\|Error\|error" gpurun_out/j16/bench2.err | tail -50
