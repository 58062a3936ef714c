"""How many row bands the fused-LayerNorm fallback kernel had to recompute in real forwards (diagnostic build: make stamps, WM_HIP_LIB=...)."""
import sys, time, torch
sys.path.insert(0, '.')
from hunyuanworld_mirror_amd import WorldMirror, WMConfig, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
assert L.wm_set_tuning(b"ln_fuse", 1) == 0
m = WorldMirror(arch=WMConfig(), dtype="bf16").to(dev).init_synthetic_weights()
g = torch.Generator().manual_seed(1234)
v = {"img": torch.rand(1, 8, 3, 518, 518, generator=g).to(dev)}
m.reserve(8, 8, 518, 518)
m(v); torch.cuda.synchronize()
print("fallback bands after warm forward:", L.wm_debug_ln_fallback_count())
for _ in range(5): m(v)
torch.cuda.synchronize()
print("fallback bands in 5 forwards (of 5 x 144 x 64 bands):", L.wm_debug_ln_fallback_count())
