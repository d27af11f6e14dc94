"""Timing ablations of the pipelined cfg2 step (development aid): what the step would cost with K0 / the registration / the screen removed."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs, build_solver
import oryon_amd
oryon_amd.configure()
from oryon_amd import ops
from oryon_amd.engine import MatchPoseEngine, MatchPoseConfig
dev = torch.device("cuda", 0)
B, H, C = 64, 224, 256
inp = make_inputs(B, H, C, first=0, dev=dev)
key = torch.arange(B, dtype=torch.int64, device=dev)
solver = build_solver(dev)
cfg = MatchPoseConfig(dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1, match_mode="screened")
def run(label, steps=30):
    eng = MatchPoseEngine(solver, cfg, overlap_registration=True, overlap_gather=True)
    sub = lambda: eng.run(inp["feat_a"], inp["feat_q"], inp["mask_a"], inp["mask_q"], inp["depth_a"], inp["depth_q"], inp["cam"], inp["cam"], key, inputs_resident=True)
    def steps_(n):
        prev = None
        for _ in range(n):
            cur = sub()
            if prev is not None: eng.finish(prev)
            prev = cur
        eng.finish(prev)
    steps_(4); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps_(steps); torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
run("full step")
orig_g = ops.gather_q8
cache = {}
def cached_gather(feat, roi, n, cap, c_pad, **kw):
    k = (feat.data_ptr(), cap, tuple(sorted(kw.items())))
    if k not in cache: cache[k] = orig_g(feat, roi, n, cap, c_pad, **kw)
    return cache[k]
ops.gather_q8 = cached_gather
run("without K0 (cached outputs)")
ops.gather_q8 = orig_g
orig_reg = solver.register
saved = {}
def cached_reg(*a, **k):
    if "o" not in saved: saved["o"] = orig_reg(*a, **k)
    return saved["o"]
solver.register = cached_reg
run("without the registration (cached poses)")
ops.gather_q8 = cached_gather
run("without K0 and registration")
