"""Randomised sweep of the registration path (K3-K10, `PointDSC.register`) against the CPU oracle (`oracle.pointdsc_forward`):
random rigid motions, inlier ratios, noise levels, correspondence counts and network sizes.  Prints the distribution of
|T_gpu - T_oracle|; gross disagreements are listed.  usage (GPU box): python tools/stress_pointdsc.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oryon_oracle as orc  # noqa: E402
from oryon_amd.pointdsc import PointDSC, get_pointdsc_pose  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
models = {}


def model(L, C, ps):
    key = (L, C, ps)
    if key not in models:
        m = PointDSC(in_dim=6, num_layers=L, num_channels=C, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
        P = orc.analytic_pointdsc_params(L, C, seed=ps)
        m.load_state_dict(P, strict=True)
        models[key] = (m.cuda().eval(), P)
    return models[key]


def rot(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


errs, gross = [], []
t0 = time.time()
for case in range(n_cases):
    L, C = [(12, 128), (6, 128), (2, 32)][case % 3]
    n = int(rng.integers(41, 501))
    inl = float(rng.uniform(0.25, 0.95))
    noise = float(rng.choice([0.0, 0.002, 0.01]))
    src = rng.uniform(-0.3, 0.3, (n, 3)) + np.array([0, 0, 0.8])
    R, t = rot(rng.normal(size=3), rng.uniform(0, 0.6)), rng.uniform(-0.1, 0.1, 3)
    tgt = src @ R.T + t + noise * rng.normal(size=(n, 3))
    out = rng.random(n) > inl
    tgt[out] = rng.uniform(-0.3, 0.3, (int(out.sum()), 3)) + np.array([0, 0, 0.8])
    m, P = model(L, C, case % 2)
    s, g = torch.from_numpy(src.astype(np.float32)), torch.from_numpy(tgt.astype(np.float32))
    cfg = dict(num_layers=L, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
    ref = orc.pointdsc_forward(s, g, P, cfg, return_all=True)
    pose = get_pointdsc_pose(m, s.cuda(), g.cuda(), "cuda").numpy()
    T_ref = ref["final_trans"].numpy().reshape(4, 4)
    e = float(np.abs(pose - T_ref).max())
    gt = np.eye(4); gt[:3, :3] = R; gt[:3, 3] = t
    errs.append(e)
    if e > 3e-3:
        gross.append((case, L, C, n, round(inl, 2), noise, e, float(np.abs(pose - gt).max()), float(np.abs(T_ref - gt).max())))
errs = np.array(errs)
print(f"{n_cases} cases in {time.time() - t0:.0f} s: max|T_gpu - T_oracle| quantiles 50/90/99/100 % = "
      f"{np.quantile(errs, 0.5):.2e} {np.quantile(errs, 0.9):.2e} {np.quantile(errs, 0.99):.2e} {errs.max():.2e}; "
      f"{(errs < 1e-4).mean() * 100:.0f} % below 1e-4, {len(gross)} above 3e-3")
for g_ in gross:
    print("  case %d L=%d C=%d n=%d inlier=%.2f noise=%g: |gpu-oracle| %.2e, |gpu-gt| %.2e, |oracle-gt| %.2e" % g_)
