"""One-off assurance run (GPU box): the whole cfg2 batch (64 pairs, 224x224, C=256) through the engine in all three matcher modes;
correspondences, lifted points, statuses and poses must be identical.   python tools/check_modes_full.py [B H C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver, make_inputs
from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine

B, H, C = (int(x) for x in (sys.argv[1:4] + ["64", "224", "256"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
inp = make_inputs(B, H, C, first=0, dev=dev)
solver = build_solver(dev)
key = torch.arange(B, dtype=torch.int64, device=dev)
outs = {}
for mode in ("exact", "screened16", "screened"):
    eng = MatchPoseEngine(solver, MatchPoseConfig(match_mode=mode))
    outs[mode] = eng.run(inp["feat_a"], inp["feat_q"], inp["mask_a"], inp["mask_q"], inp["depth_a"], inp["depth_q"], inp["cam"], inp["cam"], key, keep=True)
    torch.cuda.synchronize()
ref = outs["exact"]
for mode in ("screened16", "screened"):
    o = outs[mode]
    same = {k: bool(torch.equal(o[k], ref[k])) for k in ("status", "n_valid", "corrs", "pcd_a", "pcd_q", "pose")}
    live = torch.arange(ref["valid"].shape[1], device=dev)[None] < ref["n_a"][:, None]      # rows >= n_a are never written
    v = ref["valid"].bool() & live
    same["valid"] = bool(torch.equal(o["valid"].bool() & live, v))
    same["argmin_on_valid"] = bool(torch.equal(o["argmin"][v], ref["argmin"][v]))
    same["min_dist_on_valid"] = bool(torch.equal(o["min_dist"][v].view(torch.int32), ref["min_dist"][v].view(torch.int32)))
    print(mode, same)
    assert all(same.values()), mode
print("all modes identical on", B, "pairs; valid anchors:", int(ref["valid"].sum()), "of", int(ref["n_a"].sum()))
