"""Host-side mirror of the reference's ``Rasterizer`` (src/models/models/rasterization.py:17-93) over the C ABI entry
``wm_rasterize_splats`` (hand-written HIP: projection, tile binning, radix sort, tile compositing).  Same method names,
argument order and return triples, so ``model.gs_renderer.rasterizer.rasterize_batches(...)`` as called by
``render_interpolated_video`` (src/utils/render_utils.py:242-312) and ``GaussianSplatRenderer.render``
(rasterization.py:221-241) keeps working.  No CPU fallback: tensors must live on a HIP device."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from . import _lib


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class Rasterizer:
    def __init__(self, rasterization_mode="classic", packed=True, abs_grad=True, with_eval3d=False, camera_model="pinhole",
                 sparse_grad=False, distributed=False, grad_strategy=None):
        if rasterization_mode != "classic" or camera_model != "pinhole" or with_eval3d or distributed:
            raise NotImplementedError("only the reference's configuration is built: classic / pinhole / no eval3d / single process")
        self.rasterization_mode, self.packed, self.abs_grad, self.camera_model = rasterization_mode, packed, abs_grad, camera_model
        self.sparse_grad, self.grad_strategy, self.distributed, self.with_eval3d = sparse_grad, grad_strategy, distributed, with_eval3d
        self._ws = None          # reusable workspace (torch uint8 tensor) and the pair capacity it was sized for
        self._cap = 0
        self.last_n_isects = 0

    # rasterization.py:29-66
    def rasterize_splats(self, means, quats, scales, opacities, colors, camtoworlds, Ks, width: int, height: int,
                         sh_degree=None, **kwargs) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if kwargs:
            raise TypeError(f"unsupported gsplat.rasterization arguments: {sorted(kwargs)}")
        dev = means.device
        if dev.type != "cuda":
            raise RuntimeError("the rasteriser runs in libwm_hip.so on the GPU: move the splats to a HIP device")
        L = _lib.lib()
        N, V = int(means.shape[0]), int(camtoworlds.shape[0])
        if colors.dim() == 3:            # SH coefficients [N, K, 3]
            if sh_degree is None or int(sh_degree) != 0:
                raise NotImplementedError("SH degree 0 only (the reference renders with sh_degree = min(self.sh_degree, 0))")
            cin, is_sh = colors[:, 0, :], 1
        else:                            # post-activation colours [N, 3]
            if sh_degree is not None:
                raise ValueError("colors [N, 3] go with sh_degree = None")
            cin, is_sh = colors, 0
        means, quats, scales, opacities, cin = _f32(means), _f32(quats), _f32(scales), _f32(opacities).reshape(-1), _f32(cin)
        viewmats = _f32(torch.linalg.inv(camtoworlds.to(torch.float32)))  # :48
        Ks = _f32(Ks)
        rgb = torch.empty((V, height, width, 3), device=dev, dtype=torch.float32)
        depth = torch.empty((V, height, width, 1), device=dev, dtype=torch.float32)
        alpha = torch.empty((V, height, width, 1), device=dev, dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        cap = max(self._cap, 8 * N * V, 1 << 16)
        n = C.c_ulonglong(0)
        for _ in range(2):
            need = L.wm_rasterize_workspace_bytes(N, V, width, height, cap)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, device=dev, dtype=torch.uint8)
            self._cap = cap
            st = L.wm_rasterize_splats(p(means), p(quats), p(scales), p(opacities), p(cin), is_sh, N, p(viewmats), p(Ks), V, width, height,
                                       p(rgb), p(depth), p(alpha), None, p(self._ws), self._ws.numel(), cap, C.byref(n), stream)
            if st == 0:
                break
            if st == 3 and n.value > cap:   # WM_ERR_STATE: more (Gaussian, tile) pairs than the workspace holds
                cap = int(n.value * 1.25) + 1024
                continue
            raise RuntimeError(f"wm_rasterize_splats failed with status {st}")
        else:
            raise RuntimeError("wm_rasterize_splats: workspace re-size did not converge")
        self.last_n_isects = int(n.value)
        return rgb, depth, alpha

    # rasterization.py:68-93 (NB: the reference passes what it calls `viewmats` on as `camtoworlds`)
    def rasterize_batches(self, means, quats, scales, opacities, colors, viewmats, Ks, width, height, **kwargs):
        rc, rd, ra = [], [], []
        for i in range(len(means)):
            c, d, a = self.rasterize_splats(means[i], quats[i], scales[i], opacities[i], colors[i], viewmats[i], Ks[i], width, height, **kwargs)
            rc.append(c); rd.append(d); ra.append(a)
        return torch.stack(rc, 0), torch.stack(rd, 0), torch.stack(ra, 0)
