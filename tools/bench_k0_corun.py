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
sb = torch.cuda.Stream()
def screen():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    lib().oryon_profile_events(e0.cuda_event, e1.cuda_event)
    ops.match_screened8_raw(a_hat, a8, a_sc, feat_q, roi_q, q_norm, q8, q_sc, q_eps, na, nq, 0.25)
    return e0, e1
def k0():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record(); ops.gather_q8(feat_q, roi_q, nq, cap_q, 256); e1.record()
    return e0, e1
med = lambda v: sorted(v)[len(v) // 2]
for _ in range(3): screen(); k0()
torch.cuda.synchronize()
ts, tk, bs, bk = [], [], [], []
for _ in range(5):
    a = screen(); torch.cuda.synchronize(); ts.append(a[0].elapsed_time(a[1]))
    p = k0(); torch.cuda.synchronize(); tk.append(p[0].elapsed_time(p[1]))
for _ in range(5):
    p = k0(); a = screen(); torch.cuda.synchronize(); bs.append(a[0].elapsed_time(a[1])); bk.append(p[0].elapsed_time(p[1]))
print(f"V5={os.environ.get('ORYON_GATHER8_V5','0')} W={os.environ.get('ORYON_SCREEN8_WAVES','4')} screen alone {med(ts):.3f} | K0(query) alone {med(tk):.3f} | together: screen {med(bs):.3f}, K0 {med(bk):.3f} ms")
