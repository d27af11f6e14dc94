"""Stage-by-stage comparison of one tools/stress_pointdsc.py case against the oracle: python tools/debug_pdsc_case.py <seed> <case>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oryon_oracle as orc
from oryon_amd.pointdsc import PointDSC
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
def rot(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
for case in range(target + 1):
    L, C = [(12, 128), (6, 128), (2, 32)][case % 3]
    n = int(rng.integers(41, 501)); inl = float(rng.uniform(0.25, 0.95)); noise = float(rng.choice([0.0, 0.002, 0.01]))
    src = rng.uniform(-0.3, 0.3, (n, 3)) + np.array([0, 0, 0.8])
    R, t = rot(rng.normal(size=3), rng.uniform(0, 0.6)), rng.uniform(-0.1, 0.1, 3)
    tgt = src @ R.T + t + noise * rng.normal(size=(n, 3))
    out = rng.random(n) > inl
    tgt[out] = rng.uniform(-0.3, 0.3, (int(out.sum()), 3)) + np.array([0, 0, 0.8])
m = PointDSC(in_dim=6, num_layers=L, num_channels=C, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
P = orc.analytic_pointdsc_params(L, C, seed=case % 2); m.load_state_dict(P, strict=True); m = m.cuda().eval()
s, g = torch.from_numpy(src.astype(np.float32)), torch.from_numpy(tgt.astype(np.float32))
cfg = dict(num_layers=L, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
ref = orc.pointdsc_forward(s, g, P, cfg, return_all=True)
n_cap = (n + 127) // 128 * 128
S, T = torch.zeros((1, n_cap, 3), device="cuda"), torch.zeros((1, n_cap, 3), device="cuda")
S[0, :n], T[0, :n] = s.cuda(), g.cuda()
nn_ = torch.tensor([n], dtype=torch.int32, device="cuda")
feat, conf = m.encode(S, T, nn_)
print("n", n, "L", L, "C", C, "inliers", int((~out).sum()))
print("feat max diff", float((feat[0, :n].cpu() - ref["feat"]).abs().max()), "conf max diff", float((conf[0, :n].cpu() - ref["confidence"]).abs().max()))
seeds, ns = m.pick_seeds_batched(S, conf, nn_)
sg = seeds[0, : int(ns)].cpu().tolist(); sr = ref["seeds"].tolist()
print("seeds gpu", sg); print("seeds ref", sr)
seed_T, fit, best = m.hypotheses(S, T, feat, nn_, seeds, ns)
print("fitness gpu", [round(x, 4) for x in fit[0, : int(ns)].cpu().tolist()], "best", int(best))
print("fitness ref", [round(x, 4) for x in ref["fitness"].tolist()], "best", ref["best"])
if sg == sr:
    print("seed_T max diff per seed", [float(x) for x in (seed_T[0, : int(ns)].cpu() - ref["seed_trans"]).abs().amax(dim=(1, 2))])
