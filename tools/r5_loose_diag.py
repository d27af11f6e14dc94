"""Diagnostics behind the tightened assertions of round 5: what exactly differs where the old tests allowed slack."""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_pointdsc import load, names, build, padded
from oryon_amd import pcd

print("== nn_correspondences drop-in: differing sampled rows")
for name in names("g1_matcher_"):
    g = load(name)
    if "sampled_is_none" not in g or bool(g["sampled_is_none"]):
        continue
    torch.manual_seed(1)
    out = pcd.nn_correspondences(torch.from_numpy(g["feats1"]).cuda(), torch.from_numpy(g["feats2"]).cuda(), torch.from_numpy(g["mask1"]).cuda(),
                                 torch.from_numpy(g["mask2"]).cuda(), float(g["threshold"]), 500, 5000, "cpu")
    got, ref = out.cpu().numpy(), g["sampled_corrs"]
    diff = np.nonzero(np.any(got != ref, axis=1))[0]
    f1 = g["feats1"].astype(np.float64); f2 = g["feats2"].astype(np.float64)
    gaps = []
    for r in diff:
        a = f1[:, got[r, 0], got[r, 1]]; a = a / max(np.linalg.norm(a), 1e-8)
        def dist(y, x):
            q = f2[:, y, x]; q = q / max(np.linalg.norm(q), 1e-8)
            return 0.5 * (1 - a @ q)
        gaps.append(abs(dist(got[r, 2], got[r, 3]) - dist(ref[r, 2], ref[r, 3])))
    print(f"  {name}: {len(diff)} rows differ; anchors equal on all rows: {np.array_equal(got[:, :2], ref[:, :2])}; "
          f"max |d(got) - d(ref)| on differing rows: {max(gaps) if gaps else 0:.2e}")

print("== hypotheses from the reference's seeds: per-seed transform error and whether its kNN set is separated")
for name in names("g4_pointdsc_"):
    g = load(name); m = build(g); src, tgt, nn_, n = padded(g)
    n_cap = src.shape[1]
    feat = torch.zeros((1, n_cap, int(g["C"])), device="cuda"); feat[0, :n] = torch.from_numpy(g["feat"]).cuda()
    S = int(n * 0.1); S_cap = m.seed_cap(n_cap)
    seeds = torch.zeros((1, S_cap), dtype=torch.int32, device="cuda"); seeds[0, :S] = torch.from_numpy(g["seeds"].astype(np.int32)).cuda()
    seed_T, fitness, best = m.hypotheses(src, tgt, feat, nn_, seeds, torch.tensor([S], dtype=torch.int32, device="cuda"))
    seed_T = seed_T[0, :S].cpu().numpy()
    err = np.abs(seed_T - g["seed_trans"]).reshape(S, -1).max(1)
    fn = g["feat"].astype(np.float64); fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-12)
    k = min(40, n - 1)
    gap = []
    for s in g["seeds"][:S].astype(int):
        d = np.sort(2 - 2 * fn @ fn[s])
        gap.append(d[k + 1] - d[k] if k + 1 < n else 1.0)          # separation between the last neighbour and the first non-neighbour
    gap = np.array(gap)
    sep = gap > 1e-6
    print(f"  {name}: S={S}; err<1e-4: {(err < 1e-4).sum()}, <1e-3: {(err < 1e-3).sum()}; kNN-separated seeds {sep.sum()}: max err among them {err[sep].max() if sep.any() else 0:.2e}; "
          f"max err among the others {err[~sep].max() if (~sep).any() else 0:.2e}; worst 3 errs {np.sort(err)[-3:]}")
