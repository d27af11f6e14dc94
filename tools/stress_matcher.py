"""Extended randomised sweep of the screened matchers (K1s8 -> K1s -> exact re-scoring; the lazy int8 and MX-fp6 routes with the sampler)
against the exact fp32 scan (K1).
Same generator families as tests/test_gpu_matcher.py::test_screened_paths_randomised_stress plus adversarial ones: anchors placed
at the threshold, near-duplicate query rows at graded distances (around the fp16 / int8 decision margins), tiny and huge norms.
usage (GPU box): python tools/stress_matcher.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_matcher import _lazy_vs_eager, _screen_vs_exact  # noqa: E402

dev = "cuda"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
rng = np.random.default_rng(seed)
t0 = time.time()
fails = 0
stats = {k: [0, 0] for k in range(7)}
for case in range(n_cases):
    C = int(rng.choice([129, 160, 192, 256, 257, 300, 384, 448, 512]))
    H, W = int(rng.integers(12, 72)), int(rng.integers(12, 72))
    B = int(rng.integers(1, 5))
    thr = float(rng.choice([0.05, 0.1, 0.25, 0.4, 0.5]))
    g = torch.Generator(device=dev).manual_seed(seed * 100003 + case)
    kind = case % 7
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    fq = rn(B, C, H, W)
    if kind == 0:
        fa = fq.flip(-1) + float(rng.uniform(0.01, 0.6)) * rn(B, C, H, W)
    elif kind == 1:
        fa = rn(B, C, H, W)
    elif kind == 2:
        basis = rn(B, C, 6)
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
        coef = torch.stack([torch.ones_like(xx), xx, yy, xx * yy, torch.sin(3 * xx), torch.cos(3 * yy)])
        fq = torch.einsum("bck,khw->bchw", basis, coef) + float(rng.choice([1e-4, 1e-3, 1e-2])) * rn(B, C, H, W)
        fa = fq + float(rng.choice([1e-4, 1e-3, 5e-3])) * rn(B, C, H, W)
    elif kind == 3:
        fq = fq * torch.exp(float(rng.uniform(0.5, 3.0)) * rn(1, C, 1, 1))
        fa = fq.roll(3, -1) + 0.05 * rn(B, C, H, W)
    elif kind == 4:
        # anchors at cosine distance ~thr from their best query: a = cos(t) q + sin(t) n with dist = (1-cos t)/2 ~ thr +- jitter
        q = fq.reshape(B, C, -1)
        q = q / q.norm(dim=1, keepdim=True)
        n = rn(B, C, H * W)
        n = n - (n * q).sum(1, keepdim=True) * q
        n = n / n.norm(dim=1, keepdim=True)
        d = thr + torch.randn(B, 1, H * W, generator=g, device=dev) * float(rng.choice([1e-7, 1e-5, 1e-3]))
        cos_t = (1 - 2 * d).clamp(-1, 1)
        fa = (cos_t * q + torch.sqrt((1 - cos_t * cos_t).clamp_min(0)) * n).reshape(B, C, H, W)
        fq = q.reshape(B, C, H, W).clone()
    elif kind == 5:
        # graded near-duplicates: query pixel j+1 = query pixel j + eps_j * noise with eps spanning 1e-6 .. 1e-1
        q = fq.reshape(B, C, -1).clone()
        eps = torch.logspace(-6, -1, H * W, device=dev)[None, None, :]
        q[:, :, 1::2] = q[:, :, 0:-1:2][:, :, : q[:, :, 1::2].shape[2]] + eps[:, :, 1::2] * rn(B, C, q[:, :, 1::2].shape[2])
        fq = q.reshape(B, C, H, W)
        fa = fq + float(rng.choice([0.0, 1e-4, 1e-2])) * rn(B, C, H, W)
    else:
        # wild norms: per-pixel scales from 1e-12 to 1e6 (normalisation must absorb them; zeros stay zero rows)
        sc = torch.exp(float(rng.uniform(1, 9)) * rn(B, 1, H, W))
        fa = (fq + 0.1 * rn(B, C, H, W)) * sc
        fq = fq * torch.exp(float(rng.uniform(1, 9)) * rn(B, 1, H, W))
        fq[:, :, 0, 0] = 0.0
    dens_a, dens_q = float(rng.uniform(0.05, 1.0)), float(rng.uniform(0.05, 1.0))
    ma = (torch.rand(B, H, W, generator=g, device=dev) < dens_a).int()
    mq = (torch.rand(B, H, W, generator=g, device=dev) < dens_q).int()
    if int(ma.sum(dim=(1, 2)).max()) == 0 or int(mq.sum(dim=(1, 2)).max()) == 0:
        continue
    c_pad = 256 if C <= 256 else 512
    try:
        va, na = _screen_vs_exact(fa.contiguous(), fq.contiguous(), ma, mq, c_pad, thr=thr)
        # the lazy routes (int8 screen, MX-fp6 screen) + sampler against select_corrs on the exact scan: same valid set, same counts,
        # same sampled correspondences
        _lazy_vs_eager(fa.contiguous(), fq.contiguous(), ma, mq, c_pad, thr, max_corrs=int(rng.choice([16, 100, 500])))
        for b in range(B):
            stats[kind][0] += int(va[b, : int(na[b])].sum())
            stats[kind][1] += int(na[b])
    except AssertionError as e:
        fails += 1
        print(f"FAIL case {case} kind {kind} C={C} H={H} W={W} B={B} thr={thr}: {e}", flush=True)
print("valid fraction per generator family:", {k: round(v[0] / max(1, v[1]), 3) for k, v in stats.items()})
print(f"{n_cases} cases, {fails} failures, {time.time() - t0:.1f} s")
sys.exit(1 if fails else 0)
