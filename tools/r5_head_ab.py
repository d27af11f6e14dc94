"""Round 5: the fused confidence head + normalisation (pdsc_head_x3_kernel) against the four launches it replaces
(ORYON_PDSC_FUSED_HEAD=0, development library): encoder outputs and poses saved for a bit-for-bit comparison.
Usage: python tools/r5_head_ab.py out.pt"""
import os, sys
import _devlib  # noqa: F401
import torch
from bench import build_solver
dev = torch.device("cuda", 0)
solver = build_solver(dev)
g = torch.Generator(device=dev).manual_seed(3)
B = 16
src = torch.rand(B, 512, 3, generator=g, device=dev)
tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
n = torch.tensor([500, 512, 64, 65, 1, 200, 448, 449, 300, 128, 127, 500, 33, 400, 512, 0][:B], dtype=torch.int32, device=dev)
feat, conf = solver.encode(src, tgt, n)
live = torch.arange(512, device=dev)[None, :] < ((n[:, None] + 63) // 64 * 64)       # rows of tiles holding a live row (the others keep stale conf)
T = solver.register(src, tgt, n, torch.zeros(B, dtype=torch.int32, device=dev))
T = T[0] if isinstance(T, (tuple, list)) else T
torch.save({"feat": feat.cpu(), "conf": torch.where(live, conf, torch.zeros_like(conf)).cpu(), "T": T.cpu()}, sys.argv[1])
print("saved", float(feat.double().abs().sum()), float(T.double().sum()))
