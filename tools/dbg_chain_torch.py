"""Dependent chains of plain torch kernels on concurrent streams, with buffers rewritten inside the chain."""
import torch
dev = torch.device('cuda:0')
N = 64 * 1024 * 1024 // 4   # 64 MB fp32 buffers
def mk(seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return dict(x=torch.randn(N, generator=g).to(dev), a=torch.empty(N, device=dev), b=torch.empty(N, device=dev), y=torch.empty(N, device=dev))
def chain(d):
    torch.mul(d['x'], 2.0, out=d['a'])          # a = 2x          (write a)
    torch.add(d['a'], 1.0, out=d['b'])          # b = a + 1       (read a)
    torch.mul(d['b'], d['x'], out=d['a'])       # a = b * x       (REWRITE a)
    torch.sub(d['a'], d['b'], out=d['y'])       # y = a - b       (read the rewritten a)
jobs = [mk(i) for i in range(3)]
ref = []
for j in jobs:
    chain(j); torch.cuda.synchronize(); ref.append(j['y'].clone())
streams = [torch.cuda.Stream() for _ in jobs]
bad = 0
for it in range(30):
    for j in jobs: j['y'].fill_(float('nan'))
    torch.cuda.synchronize()
    for j, s in zip(jobs, streams):
        with torch.cuda.stream(s):
            for _ in range(3): chain(j)
    torch.cuda.synchronize()
    for ji, (j, r) in enumerate(zip(jobs, ref)):
        nb = int((j['y'] != r).sum())
        if nb: bad += 1; print("it", it, "job", ji, "mismatching elements", nb)
print("torch-kernel chains, bad:", bad)
