"""Round 5: stand-alone time of the CLIP tower's fp16x3 attention (oryon_mha_f16x3) at the cfg2 batch: 128 images x 577 tokens x 16 heads.
Usage: python tools/r5_mha_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401  (ORYON_MHA_PIPE=0: the unpipelined kernel)
from oryon_amd import ops

torch.set_grad_enabled(False)
N, L, H = 128, 577, 16
torch.manual_seed(0)
qkv = torch.randn(N, L, 3 * H * 64, device="cuda")
for _ in range(3):
    out = ops.mha_f16x3(qkv, H)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.mha_f16x3(qkv, H)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
flop = 4.0 * N * H * L * L * 64
print(f"mha_f16x3 {N} x {L} x {H} heads: {ms:.3f} ms per call, {flop / ms / 1e9:.0f} TFLOP/s fp32-equivalent, {3 * flop / ms / 1e9:.0f} on the fp16 pipe")
q, k, v = qkv.double().view(N, L, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = torch.nn.functional.scaled_dot_product_attention(q[:8], k[:8], v[:8]).permute(0, 2, 1, 3).reshape(8, L, H * 64)
print("max rel err vs fp64 (8 images):", float((out[:8].double() - ref).abs().max() / ref.abs().max()))
if len(sys.argv) > 1:
    torch.save(out.cpu(), sys.argv[1])
