import ctypes as C, sys, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
p = lambda t: C.c_void_p(t.data_ptr())
def run(q, k, v, Lseq, H=1):
    M = q.shape[1]
    o = torch.empty(M, H * 64, device=dev, dtype=torch.int16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.wm_op_attention(0, p(q), p(k), p(v), p(o), H, M, Lseq, 1, 0, s) == 0
    torch.cuda.synchronize()
    return o.view(torch.bfloat16).float().reshape(M, H, 64).transpose(0, 1)
def ref(q, k, v):
    return torch.softmax(q.float() @ k.float().transpose(-1, -2), -1) @ v.float()
bf = lambda x: x.to(torch.bfloat16).to(dev)
g = torch.Generator().manual_seed(0)
for Ls in (64, 32, 128):
    print("==== L", Ls)
    z = bf(torch.zeros(1, Ls, 64)); V = bf(torch.randn(1, Ls, 64, generator=g))
    o = run(z, z, V, Ls); r = ref(z, z, V)
    print("case1 q=k=0 (uniform): err", float((o - r).abs().max()), "o[0,0,:4]", o[0, 0, :4].tolist(), "ref", r[0, 0, :4].tolist())
    Vk = bf(torch.arange(Ls).float()[None, :, None].expand(1, Ls, 64).contiguous())
    o = run(z, z, Vk, Ls); print("case2 V=key idx: o[0,0,:4]", o[0, 0, :4].tolist(), "expect", (Ls - 1) / 2)
    Vd = bf(torch.arange(64).float()[None, None, :].expand(1, Ls, 64).contiguous())
    o = run(z, z, Vd, Ls); print("case3 V=d idx: o[0,0,:8]", o[0, 0, :8].tolist(), " o[0,5,32:36]", o[0, 5, 32:36].tolist())
    Q = bf(torch.randn(1, Ls, 64, generator=g) * 0.3); K = bf(torch.randn(1, Ls, 64, generator=g))
    ones = bf(torch.ones(1, Ls, 64))
    o = run(Q, K, ones, Ls); print("case4 V=1: min/max", float(o.min()), float(o.max()))
    o = run(Q, K, Vk, Ls); r = ref(Q, K, Vk); print("case5 random qk, V=key idx: err", float((o - r).abs().max()), o[0, :4, 0].tolist(), r[0, :4, 0].tolist())
    o = run(Q, K, V, Ls); r = ref(Q, K, V); print("case6 full random: err", float((o - r).abs().max()))
