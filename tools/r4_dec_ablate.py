"""Round-4 probe: the HIP decoder's time up to a given stage (`stop_after` of oryon_decoder_forward: 7 = block 3's cat buffer, 8 / 9 = its first
/ second convolution, 0 = the whole module), 128 images.  The phase ablation recorded in DESIGN.md (no tile loads / no MFMA loop / no output
stores) used temporary switches inside dec_conv3x3_kernel (commit "Decoder: persistent register-weight variant ..."); they are not in the
library any more - no switch of the shipped kernels changes results."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oryon_amd
oryon_amd.configure()
from oryon_amd.backbone.fusion import StandardDecoder
from oryon_amd.backbone.decoder_hip import HipDecoder
torch.manual_seed(0)
dec = StandardDecoder("cuda", True, True, input_dim=128, decoder_dims=[64, 32]).eval()
n = 128
x = torch.randn(n, 128, 24, 24, device="cuda")
g2 = torch.randn(n, 48, 48, 256, device="cuda").permute(0, 3, 1, 2)
g3 = torch.randn(n, 96, 96, 128, device="cuda").permute(0, 3, 1, 2)
hip = HipDecoder(dec, x.device)
for stop in (0, 7, 8, 9):
    for _ in range(2):
        hip.forward(x, g2, g3, stop_after=stop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hip.forward(x, g2, g3, stop_after=stop)
    e1.record()
    torch.cuda.synchronize()
    print(f"mask {os.environ.get('ORYON_DEC_DEBUG', '0'):>5s} stop_after {stop}: {e0.elapsed_time(e1) / 5:.3f} ms")
