"""Minimal driver for PMC passes over mha_x3_kernel: three calls of oryon_mha_f16x3 at the cfg2 batch, nothing else."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oryon_amd import ops
torch.set_grad_enabled(False)
torch.manual_seed(0)
qkv = torch.randn(128, 577, 3 * 16 * 64, device="cuda")
for _ in range(3):
    out = ops.mha_f16x3(qkv, 16)
torch.cuda.synchronize()
print("ok", float(out[0, 0, 0]))
