"""Time K0 gather+normalise at a BASELINE geometry: python tools/bench_gather.py [H C B]  (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd.synth import make_pair

H, C, B = (int(x) for x in (sys.argv[1:4] + ["224", "256", "64"][len(sys.argv) - 1:]))
dev = "cuda"
p = make_pair(0, H, H, C, device=dev)
feat = p["feat_q"][None].expand(B, -1, -1, -1).contiguous()
mask = p["mask_q"][None].expand(B, -1, -1).contiguous()
roi, n = ops.roi_compact(mask)
cap = ops.round_up(int(n.max()), 256)
cp = 128 if C <= 128 else 256 if C <= 256 else 512
for _ in range(3):
    out = ops.gather_normalise(feat, roi, n, cap, c_pad=cp, want_f16=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = ops.gather_normalise(feat, roi, n, cap, c_pad=cp, want_f16=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
rows = float(n.sum())
gb = rows * (4 * C + 6 * cp) / 1e9
print(f"gather H={H} C={C} B={B} NL={os.environ.get('ORYON_GATHER_NL', 'auto')}: {ms:.3f} ms, rows/map={rows / B:.0f}, {gb:.2f} GB -> {gb / ms:.2f} TB/s")
