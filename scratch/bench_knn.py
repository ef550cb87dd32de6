import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd.neighbors import knn_grid, transcripts_graph
from segger_amd.synthetic import SyntheticSpec, make_graph
b = make_graph(SyntheticSpec(n_tx=1_000_000, n_bd=10_000, k_tx=15, seed=0))
pos = b['tx'].pos.cuda()
for _ in range(2): knn_grid(pos, 15)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): nbr,_ = knn_grid(pos, 15)
torch.cuda.synchronize(); print('knn 1M k=15: %.2f ms' % ((time.perf_counter()-t)/5*1e3))
ref = b[('tx','neighbors','tx')].edge_index[1].view(-1,15)
print('agree with scipy:', (nbr.cpu().long()==ref).float().mean().item())
