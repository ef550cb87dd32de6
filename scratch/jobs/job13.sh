cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j13
python -m pytest tests/test_gpu_tiles.py -x -q -k "rank_local" 2>&1 | tail -15
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --strong-n-tx 10000000 --strong-n-bd 100000 > gpurun_out/j13/bench2.json 2> gpurun_out/j13/bench2.err; echo "bench rc $?"
grep "fixed FOV" gpurun_out/j13/bench2.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/j13/bench2.json').read().strip().splitlines()[-1])
s=d['strong']; print({k:s[k] for k in ('n_ranks_seen','census','ms_per_step','resident','peak_hbm_gib')}); print(s.get('graphed',{}).get('ms_per_step'))
P
