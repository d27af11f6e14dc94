#!/usr/bin/env python3
"""Timing of the screened matcher (K1s) vs the exact fp32 matcher (K1) on cfg2-sized data (development aid)."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd.synth import make_pair

ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=64); ap.add_argument("--C", type=int, default=256)
ap.add_argument("--H", type=int, default=224); ap.add_argument("--iters", type=int, default=3); a = ap.parse_args()
dev = "cuda"
pairs = [make_pair(i, a.H, a.H, a.C, device=dev) for i in range(a.B)]
st = lambda k: torch.stack([p[k] for p in pairs])
feat_a, feat_q, mask_a, mask_q = st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q"); del pairs
roi_a, na = ops.roi_compact(mask_a); roi_q, nq = ops.roi_compact(mask_q); ops.roi_subsample_(roi_a, na, 5000, seed=1)
cap_q = ops.round_up(a.H * a.H, 256)
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(a.iters):
    e0 = ev(); a_hat, a16 = ops.gather_normalise(feat_a, roi_a, na, 5120, want_f16=True); q_hat, q16 = ops.gather_normalise(feat_q, roi_q, nq, cap_q, want_f16=True)
    e1 = ev(); md1, am1, va1 = ops.match_screened(a_hat, q_hat, a16, q16, na, nq, 0.25)
    e2 = ev(); md0, am0, va0 = ops.match(a_hat, q_hat, na, nq, 0.25)
    e3 = ev(); torch.cuda.synchronize()
    fl = 2.0 * (na.double() * nq.double()).sum().item() * a.C
    t1, t0 = e1.elapsed_time(e2), e2.elapsed_time(e3)
    print(f"iter {it}: gather(f32+f16) {e0.elapsed_time(e1):.2f} ms | screened {t1:.2f} ms ({2*fl/t1/1e9:.0f} TF/s fp16 over 2 passes) | exact fp32 {t0:.2f} ms ({fl/t0/1e9:.0f} TF/s)")
ok = all(torch.equal(va0[b, :int(na[b])], va1[b, :int(na[b])]) and torch.equal(am0[b, :int(na[b])][va0[b, :int(na[b])].bool()], am1[b, :int(na[b])][va0[b, :int(na[b])].bool()]) for b in range(a.B))
print("screened == exact on valid rows:", ok)
