import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_golden import load_encoder_golden, golden_model
dev = torch.device("cuda")
z, sd, b = load_encoder_golden()
bg = b.to(dev)
grads = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    m = golden_model(sd, dev, dt).eval()
    emb = m(bg)
    loss = m._segmentation_loss(emb, bg, torch.from_numpy(z["in::neg"]).to(dev))
    loss.backward()
    grads[dt] = {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}
ref = grads[torch.float32]
for k in ref:
    r = ref[k]; g = grads[torch.bfloat16][k]; h = grads[torch.float16][k]
    o = torch.from_numpy(z["grad::model." + k[6:]] if ("grad::" + k) not in z.files else z["grad::" + k]).double() if (("grad::" + k) in z.files) else None
    print(f"{k:70s} |g|2={r.norm():.3e} max={r.abs().max():.3e} bf16 relL2={(g-r).norm()/r.norm():.3e} relmax={(g-r).abs().max()/r.abs().max():.3e} f16 relL2={(h-r).norm()/r.norm():.3e}" + (f" f32-vs-oracle relL2={(r-o).norm()/o.norm():.2e}" if o is not None else ""))
