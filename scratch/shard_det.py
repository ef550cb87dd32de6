import sys, hashlib, torch
sys.path.insert(0, '/root/repo')
from segger_amd.fov import build_fov_shard, build_fov_batches, batch_weights
from segger_amd.synthetic import SyntheticSpec
from segger_amd.dp import rank_schedule
rank = int(sys.argv[1])
spec = SyntheticSpec(n_tx=10_000_000, n_bd=100_000, k_tx=15, seed=0)
dev = torch.device('cuda')
part, batches, lb, sched, aux, tiling, info = build_fov_shard(spec, dev, rank, 2)
h = hashlib.md5(repr(info['weights']).encode()).hexdigest()
print('rank', rank, 'n_batches', len(batches), 'weights md5', h, 'sched lens', [len(s) for s in sched], 'own', len(lb), flush=True)
if rank == 0:
    del part
    torch.cuda.empty_cache()
    p2, b2, a2, t2 = build_fov_batches(spec, dev)
    w2 = batch_weights(p2, b2)
    print('whole: n_batches', len(b2), 'md5', hashlib.md5(repr(w2).encode()).hexdigest(), 'equal', w2 == info['weights'], b2 == batches,
          [len(s) for s in rank_schedule(w2, 2)], flush=True)
