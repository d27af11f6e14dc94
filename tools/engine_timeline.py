"""Per-step timeline of the native step engine on the cfg2 workload (HIP events of oryon_engine_set_timing): for the last 12 steps of
a 20-step window the absolute start / end of the gather, match and registration sections (ms since the first event), so that
bubbles between sections and the overlap between steps can be read off.  `python tools/engine_timeline.py [steps]`."""
import os
import sys
import ctypes

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _devlib  # noqa: E402,F401  (the development library: ORYON_* switches are live)
import bench  # noqa: E402
import oryon_amd
oryon_amd.configure()
from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
B, H, C = int(os.environ.get("ENG_B", 64)), int(os.environ.get("ENG_H", 224)), int(os.environ.get("ENG_C", 256))
inp = bench.make_inputs(B, H, C, 0, dev)
inp["cam"] = inp["cam"].reshape(B, 9).float().contiguous()
ser = bool(os.environ.get("ENG_SERIAL"))
eng = MatchPoseEngine(bench.build_solver(dev), MatchPoseConfig(), overlap_registration=not ser, overlap_gather=not ser, native=True, result_views=True)
eng.native_timing = not os.environ.get("ENG_NO_TIMING")
for k_ in ("n_slots", "gather_sets", "reg_streams", "reg_lag", "screen", "x3_prefetch"):
    if os.environ.get("ENG_" + k_.upper()):
        eng.native_geometry[k_] = int(os.environ["ENG_" + k_.upper()])
if os.environ.get("ENG_PYTHON"):
    eng = MatchPoseEngine(bench.build_solver(dev), MatchPoseConfig(), overlap_registration=True, overlap_gather=True, native=False)
    eng.py_timeline = []
key = torch.arange(B, device=dev)
if os.environ.get("ENG_HARD"):
    # bench.py's `hard_descriptors` inputs: smooth rank-8 fields + noise (every anchor ambiguous for the int8 stage)
    gen = torch.Generator(device=dev).manual_seed(77)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, H, device=dev), indexing="ij")
    coef = torch.stack([torch.ones_like(xx), xx, yy, xx * yy, torch.sin(3 * xx), torch.cos(3 * yy), torch.sin(7 * yy), torch.cos(5 * xx)])
    basis = torch.randn((B, C, coef.shape[0]), generator=gen, device=dev)
    inp["feat_q"].copy_(torch.einsum("bck,khw->bchw", basis, coef))
    inp["feat_q"].add_(0.02 * torch.randn(inp["feat_q"].shape, generator=gen, device=dev))
    inp["feat_a"].copy_(inp["feat_q"]).add_(0.01 * torch.randn(inp["feat_a"].shape, generator=gen, device=dev))
    torch.cuda.synchronize()


PACE = float(os.environ.get("ENG_PACE_MS", "0")) * 1e-3


def run(n):
    import time as _t
    prev = None
    for _ in range(n):
        t_ = _t.perf_counter()
        while PACE and _t.perf_counter() - t_ < PACE:
            pass
        cur = eng.run(inp["feat_a"], inp["feat_q"], inp["mask_a"], inp["mask_q"], inp["depth_a"], inp["depth_q"], inp["cam"], inp["cam"], key,
                      inputs_resident=True)
        if prev is not None:
            eng.finish(prev)
        prev = cur
    eng.finish(prev)


if os.environ.get("ENG_STATS"):
    eng.collect_i8_stats = True
if os.environ.get("ENG_SF_TOGGLE"):
    eng.cfg.sample_first = 1024
    run(8)
    torch.cuda.synchronize()
    eng.cfg.sample_first = 0
if os.environ.get("ENG_SIDE_STREAM"):
    _side = torch.cuda.Stream()
    torch.cuda.set_stream(_side)          # the caller's stream is not the legacy default stream: blocking streams do not synchronise with it
run(5)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
nat = eng._native
if nat is None and getattr(eng, "py_timeline", None):
    tl = eng.py_timeline[-min(steps, 14):]
    base = tl[0][0]
    print("python schedule, absolute ms since the gather start of the first listed step:  G[begin end]  M[begin end]  R[begin end]")
    for i, t in enumerate(tl):
        e = {k: base.elapsed_time(v) for k, v in t.items()}
        print(f"{i:4d}  G[{e[0]:7.2f} {e[1]:7.2f}]  M[{e[2]:7.2f} {e[3]:7.2f}]  R[{e[6]:7.2f} {e[7]:7.2f}]")
if nat is None or not eng.native_timing:
    raise SystemExit(0)
first = nat.steps - min(steps, 14)
print("absolute ms since the gather start of step", first, ": step  G[begin end]  M[begin end] (screen begin end)  R[begin end]")
for k in range(first, nat.steps):
    ts = [nat.elapsed(first, 0, k, ev) for ev in range(8)]
    print(f"{k:4d}  G[{ts[0]:7.2f} {ts[1]:7.2f}]  M[{ts[2]:7.2f} {ts[3]:7.2f}] ({ts[4]:7.2f} {ts[5]:7.2f})  R[{ts[6]:7.2f} {ts[7]:7.2f}]")
