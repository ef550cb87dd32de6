import torch, time
dev='cuda'
N,M,K=1_000_000,384,256
gy=torch.randn(N,M,device=dev).bfloat16(); x=torch.randn(N,K,device=dev).bfloat16()
def t(fn,it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(True); b=torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
print('gy.t()@x', t(lambda: gy.t()@x))
print('(x.t()@gy).t()', t(lambda: (x.t()@gy).t()))
ref=(gy.t().float()@x.float())
for S in (16,64,128,256,512,1024):
    n=N//S*S
    f=lambda: torch.bmm(gy[:n].view(S,n//S,M).transpose(1,2), x[:n].view(S,n//S,K)).sum(0,dtype=torch.float32)
    out=f()
    err=(out-(gy[:n].t().float()@x[:n].float())).abs().max().item()
    print('bmm S',S, t(f), 'err',err, 'scale', ref.abs().max().item())
for K2,M2 in ((128,384),(128,64),(256,64)):
    gy2=torch.randn(N,M2,device=dev).bfloat16(); x2=torch.randn(N,K2,device=dev).bfloat16()
    print(K2,M2,'plain', t(lambda: gy2.t()@x2))
    S=256; n=N//S*S
    print(K2,M2,'bmm256', t(lambda: torch.bmm(gy2[:n].view(S,n//S,M2).transpose(1,2), x2[:n].view(S,n//S,K2)).sum(0,dtype=torch.float32)))
