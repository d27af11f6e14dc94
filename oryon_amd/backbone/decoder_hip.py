"""StandardDecoder.forward (models/decoder.py:82-108) through the C ABI's oryon_decoder_* entries (csrc/decoder.hip): the façade packs the
module's parameters once per (module, device) and keeps one workspace per input shape.  Inference only (torch.no_grad, fp32, CUDA)."""
import ctypes
from typing import Dict, List, Tuple

import torch
from torch import Tensor

from .. import _lib
from ..ops import LAYOUT_NCHW, LAYOUT_NHWC, check, lib, ptr, stream_ptr


class HipDecoder:
    """Handle + workspaces for one StandardDecoder module (decoder_dims [64, 32], extra upsampling, guidance: what get_decoder builds)."""

    def __init__(self, module: torch.nn.Module, device: torch.device):
        self.device = _lib.require_gpu(device)
        sd = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in module.state_dict().items()}
        w = _lib.DecoderWeights()
        for i in range(2):
            w.gp_w[i] = ptr(sd[f"decoder_guidance_projection.{i}.0.weight"])
            w.gp_b[i] = ptr(sd[f"decoder_guidance_projection.{i}.0.bias"])
        for i in range(3):
            p = f"decoder{i + 1}."
            w.up_w[i], w.up_b[i] = ptr(sd[p + "up.weight"]), ptr(sd[p + "up.bias"])
            w.c1_w[i] = ptr(sd[p + "conv.double_conv.0.weight"])
            w.n1_g[i], w.n1_b[i] = ptr(sd[p + "conv.double_conv.1.weight"]), ptr(sd[p + "conv.double_conv.1.bias"])
            w.c2_w[i] = ptr(sd[p + "conv.double_conv.3.weight"])
            w.n2_g[i], w.n2_b[i] = ptr(sd[p + "conv.double_conv.4.weight"]), ptr(sd[p + "conv.double_conv.4.bias"])
        w.head_w, w.head_b = ptr(sd["head.weight"]), ptr(sd["head.bias"])
        shapes = {"decoder_guidance_projection.0.0.weight": (32, 256, 3, 3), "decoder_guidance_projection.1.0.weight": (16, 128, 3, 3),
                  "decoder_guidance_projection.0.0.bias": (32,), "decoder_guidance_projection.1.0.bias": (16,),
                  "decoder1.up.weight": (128, 96, 2, 2), "decoder2.up.weight": (64, 48, 2, 2), "decoder3.up.weight": (32, 32, 2, 2),
                  "decoder1.up.bias": (96,), "decoder2.up.bias": (48,), "decoder3.up.bias": (32,),
                  "decoder1.conv.double_conv.0.weight": (64, 128, 3, 3), "decoder2.conv.double_conv.0.weight": (32, 64, 3, 3),
                  "decoder3.conv.double_conv.0.weight": (32, 32, 3, 3),
                  "decoder1.conv.double_conv.3.weight": (64, 64, 3, 3), "decoder2.conv.double_conv.3.weight": (32, 32, 3, 3),
                  "decoder3.conv.double_conv.3.weight": (32, 32, 3, 3), "head.weight": (1, 32, 3, 3), "head.bias": (1,)}
        for i, c in enumerate((64, 32, 32)):
            for j in (1, 4):
                shapes[f"decoder{i + 1}.conv.double_conv.{j}.weight"] = shapes[f"decoder{i + 1}.conv.double_conv.{j}.bias"] = (c,)
        for k, s in shapes.items():
            if k not in sd or tuple(sd[k].shape) != s:
                raise ValueError(f"oryon_decoder_*: {k} is {tuple(sd[k].shape) if k in sd else 'missing'}, the HIP decoder is built for {s} (get_decoder's module)")
        # the kernels hard-code GroupNorm(C / 16 groups, eps 1e-5) (models/decoder.py:14-19): refuse a module that normalises differently
        for name, mod in module.named_modules():
            if isinstance(mod, torch.nn.GroupNorm) and (mod.num_groups != mod.num_channels // 16 or abs(mod.eps - 1e-5) > 1e-12 or not mod.affine):
                raise ValueError(f"oryon_decoder_*: {name} is GroupNorm({mod.num_groups}, {mod.num_channels}, eps={mod.eps}); the HIP decoder "
                                 "implements GroupNorm(C // 16, C, eps=1e-5, affine)")
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().oryon_decoder_create(ctypes.byref(w), ctypes.byref(self._h), stream_ptr(self.device)), "oryon_decoder_create")
        self._ws: Dict[Tuple[int, int, int], Tensor] = {}

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().oryon_decoder_destroy(h)
            except Exception:
                pass

    def workspace(self, n: int, h: int, w: int) -> Tensor:
        key = (n, h, w)
        if key not in self._ws:
            nbytes = int(lib().oryon_decoder_workspace_bytes(n, h, w))
            if nbytes <= 0:
                raise ValueError(f"oryon_decoder_forward: unsupported shape n={n}, h={h}, w={w} (h, w multiples of 8)")
            self._ws.clear()                                  # one shape at a time: the buffers are ~14 MB per image at 24 x 24
            self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    def layout(self, n: int, h: int, w: int) -> List[int]:
        off = (ctypes.c_int64 * 3)()
        check(lib().oryon_decoder_workspace_layout(n, h, w, off), "oryon_decoder_workspace_layout")
        return [int(o) for o in off]

    def forward(self, x: Tensor, g2: Tensor, g3: Tensor, stop_after: int = 0) -> Tuple[Tensor, Tensor]:
        """x [n,128,h,w], g2 [n,256,2h,2w], g3 [n,128,4h,4w] fp32 -> (logits [n,8h,8w], featmap [n,32,8h,8w])."""
        n, c, h, w = x.shape
        if n == 0:                                          # empty batch: what the torch module returns, no launch
            return (torch.empty((0, 8 * h, 8 * w), dtype=torch.float32, device=self.device),
                    torch.empty((0, 32, 8 * h, 8 * w), dtype=torch.float32, device=self.device))
        assert c == 128 and tuple(g2.shape) == (n, 256, 2 * h, 2 * w) and tuple(g3.shape) == (n, 128, 4 * h, 4 * w), (x.shape, g2.shape, g3.shape)
        x = x.to(torch.float32).contiguous()
        # the Swin tower hands out permuted views of its [n, H, W, C] token maps (backbone/swin.py::guidance_embeds): read them in place
        nhwc = all(g.dtype == torch.float32 and g.permute(0, 2, 3, 1).is_contiguous() for g in (g2, g3))
        if nhwc:
            g2, g3 = g2.permute(0, 2, 3, 1), g3.permute(0, 2, 3, 1)              # the contiguous storage order
        else:
            g2, g3 = g2.to(torch.float32).contiguous(), g3.to(torch.float32).contiguous()
        ws = self.workspace(n, h, w)
        fm = torch.empty((n, 32, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        lg = torch.empty((n, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().oryon_decoder_forward(self._h, ptr(x), ptr(g2), ptr(g3), n, h, w, ptr(ws), ws.numel(), ptr(fm), ptr(lg),
                                              LAYOUT_NHWC if nhwc else LAYOUT_NCHW, int(stop_after),
                                              stream_ptr(self.device)), "oryon_decoder_forward")
        return lg, fm
