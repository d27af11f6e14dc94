"""Premise check for a register-light K0: the int8 screening launch of cfg2 alone and beside a small-footprint streaming kernel on another stream
(build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probe_coresidency.hip -o build_probe/libprobe_cores.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd._lib import lib
from oryon_amd.synth import make_pair
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libprobe_cores.so"))
P.launch_stream_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
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
src = torch.randn(3 << 28, device=dev)          # 3 GiB to read
dst = torch.empty(3 << 26, device=dev)
sb = torch.cuda.Stream()
def screen():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    lib().oryon_profile_events(e0.cuda_event, e1.cuda_event)
    ops.match_screened8_raw(a_hat, a8, a_sc, feat_q, roi_q, q_norm, q8, q_sc, q_eps, na, nq, 0.25)
    return e0, e1
def probe(groups, lds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        e0.record()
        rc = P.launch_stream_probe(src.data_ptr(), dst.data_ptr(), src.numel() * 4, groups, lds, 4, sb.cuda_stream); assert rc == 0, rc
        e1.record()
    return e0, e1
med = lambda v: sorted(v)[len(v) // 2]
for _ in range(3): screen()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a = screen(); torch.cuda.synchronize(); ts.append(a[0].elapsed_time(a[1]))
print(f"screen alone: {med(ts):.3f} ms   (ORYON_SCREEN8_WAVES={os.environ.get('ORYON_SCREEN8_WAVES', '4')})")
for groups, lds in ((256, 81920), (512, 32768), (1024, 16384), (2048, 0)):
    tp = []
    for _ in range(3):
        p = probe(groups, lds); torch.cuda.synchronize(); tp.append(p[0].elapsed_time(p[1]))
    both_s, both_p = [], []
    for _ in range(5):
        p = probe(groups, lds); a = screen(); torch.cuda.synchronize()
        both_s.append(a[0].elapsed_time(a[1])); both_p.append(p[0].elapsed_time(p[1]))
    print(f"probe {groups} x 64 threads, {lds >> 10} KB LDS: alone {med(tp):.3f} ms ({3.75 * 1.0737 / med(tp):.2f} TB/s) | together: screen {med(both_s):.3f} ms, probe {med(both_p):.3f} ms")
