cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j18
SEGGER_BENCH_WATCHDOG=100 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --strong-n-tx 10000000 --strong-n-bd 100000 > gpurun_out/j18/bench2.json 2> gpurun_out/j18/bench2.err; echo "bench rc $?"
grep "fixed FOV\|Error" gpurun_out/j18/bench2.err | cut -c1-260
python - <<'P'
import json
d=json.loads(open('gpurun_out/j18/bench2.json').read().strip().splitlines()[-1])
s=d['strong']; print({k:s[k] for k in ('n_ranks_seen','census','ms_per_step','resident','peak_hbm_gib')}); print('graphed', s.get('graphed',{}))
print(d['value'], d['ms_per_step'], d['n_gpus'])
P
