"""Time the int8 screening launch alone at cfg2 (development aid; ORYON_SCREEN8_VARIANT / ORYON_SCREEN8_ABLATE select kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd._lib import lib
from oryon_amd.synth import make_pair
H, C, B = 224, 256, 64
dev = "cuda"
pairs = [make_pair(i, H, H, C, device=dev) for i in range(B)]
st = lambda k: torch.stack([p[k] for p in pairs])
feat_a, feat_q, mask_a, mask_q = st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q")
del pairs
roi_a, na = ops.roi_compact(mask_a); roi_q, nq = ops.roi_compact(mask_q); ops.roi_subsample_(roi_a, na, 5000, seed=1)
cap_a, cap_q = 5120, ops.round_up(H * H, 256)
a8, a_sc, _, _, a_hat = ops.gather_q8(feat_a, roi_a, na, cap_a, 256, want_f32=True)
q8, q_sc, q_eps, q_norm, _ = ops.gather_q8(feat_q, roi_q, nq, cap_q, 256)
ops_total = 2.0 * float((na.double() * nq.double()).sum()) * C
ts = []
for it in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    lib().oryon_profile_events(e0.cuda_event, e1.cuda_event)
    ops.match_screened8_raw(a_hat, a8, a_sc, feat_q, roi_q, q_norm, q8, q_sc, q_eps, na, nq, 0.25)
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[2:])[len(ts[2:]) // 2]
print(f"variant={os.environ.get('ORYON_SCREEN8_VARIANT','2')} ablate={os.environ.get('ORYON_SCREEN8_ABLATE','0')}: screen launch {t:.3f} ms = {ops_total / t / 1e9:.0f} TOP/s")
