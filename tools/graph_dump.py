"""Dumps the captured training step of one 1M-edge batch as a DOT file (hipGraphDebugDotPrint) and prints its node kinds,
fan-out / fan-in nodes and the nodes that are not kernels:   python tools/graph_dump.py gpurun_out/step_graph.dot"""
import collections, os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segger_amd import LitISTEncoder
from segger_amd.fov import build_fov_batches
from segger_amd.synthetic import SyntheticSpec
from segger_amd import train_step_graph as tsg

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_graph.dot"
dev = torch.device("cuda:0")
spec = SyntheticSpec(n_tx=2_000_000, n_bd=20_000, k_tx=15, seed=0)
part, batches, aux, _ = build_fov_batches(spec, dev)
batches = [b for b in batches if len(b) == 1][:4]
torch.manual_seed(0)
m = LitISTEncoder(n_genes=spec.n_genes, in_channels=128)
m.model._materialize_bd(spec.bd_dim, "cpu")
m = m.to(dev)
m.set_similarities(aux["tx_similarity"].to(dev), aux["bd_similarity"].to(dev))
m._max_epochs_override, m.current_epoch = 20, 10
m.model.compute_dtype = torch.bfloat16
m.train()
orig = torch.cuda.CUDAGraph
def Dbg(*a, **k):                      # every graph the trainer creates keeps its hipGraph_t for the dump
    return orig(keep_graph=True)
torch.cuda.CUDAGraph = Dbg
tr = tsg.GraphedTrainer(m, m.configure_optimizers(capturable=True))
for ids in batches:
    tr.step(part.batch(ids))
torch.cuda.synchronize()
b = tr.buckets[0]
b.graph.debug_dump(out)
txt = open(out).read()
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
fan_out, fan_in = collections.Counter(a for a, _ in edges), collections.Counter(b_ for _, b_ in edges)
kinds = collections.Counter(re.split(r'[\\n ]', lab.strip())[0][:24] for lab in nodes.values())
print(len(nodes), "nodes", len(edges), "edges")
print("fan-out > 1:", [(n, nodes.get(n, '?')[:60], c) for n, c in fan_out.items() if c > 1][:20])
print("fan-in > 1:", [(n, nodes.get(n, '?')[:60], c) for n, c in fan_in.items() if c > 1][:20])
for n, lab in nodes.items():
    if not re.search(r'kernel|Kernel|segger|void|at::', lab):
        print("non-kernel node:", n, lab[:100].replace("\n", " "))
