#!/usr/bin/env python3
"""Test driver: the counterpart of the reference's `run_test.py` (hydra + Lightning `trainer.test(system, test_data)`, which ends
up calling `FPM_Pipeline.test_step` per batch, pipeline.py:306-355) for this build, on SYNTHETIC pairs (BASELINE.json configs[0]:
no dataset / checkpoint can be fetched here).  Per batch it runs  Pipeline.test_step_batched  (or, with --per-sample, the
reference-shaped per-sample loop  Pipeline.test_step), appends one prediction line per pair in the reference's CSV format
(pipeline.py:490-497) and finally prints ADD / ADD-S / rotation / translation errors against the generator's ground truth
(utils/metrics.py:194-259 as restated in oryon_amd/evaluation.py).

    python run_test.py --pairs 8 --batch 4                      # descriptor maps given (C=32 @ 192x192, the reference's shapes)
    python run_test.py --pairs 2 --batch 2 --backbone           # random-init Oryon.forward in front (CLIP ViT-L + Swin + fusion + decoder)
    python run_test.py --pairs 4 --batch 2 --per-sample         # reference-shaped loop, host RNG

Needs an MI355X (the match / lift / registration path has no CPU fallback by design).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oryon_amd import evaluation as ev  # noqa: E402
from oryon_amd.pipeline import Pipeline, default_args  # noqa: E402
from oryon_amd.synth import make_pair  # noqa: E402


def synthetic_batch(first, B, H, C, dev):
    pairs = [make_pair(first + i, H, H, C) for i in range(B)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    anchor_pose = torch.eye(4).repeat(B, 1, 1)
    anchor_pose[:, :3, 3] = torch.tensor([0.01, -0.02, 0.8])
    ids = list(range(first, first + B))
    batch = {
        "featmap_a": st("feat_a").to(dev), "featmap_q": st("feat_q").to(dev),
        "anchor": {"mask": st("mask_a").to(torch.uint8), "orig_depth": [p["depth_a"] for p in pairs], "camera": st("camera"),
                   "pose": anchor_pose, "instance_id": [f"synthetic {i} anchor" for i in ids], "sizes": torch.tensor([[H, H]] * B),
                   "rgb": torch.rand(B, 3, 224, 224, generator=torch.Generator().manual_seed(first))},
        "query": {"mask": st("mask_q").to(torch.uint8), "orig_depth": [p["depth_q"] for p in pairs], "camera": st("camera"),
                  "pose": torch.bmm(st("pose").float(), anchor_pose), "instance_id": [f"synthetic {i} query" for i in ids],
                  "sizes": torch.tensor([[H, H]] * B), "rgb": torch.rand(B, 3, 224, 224, generator=torch.Generator().manual_seed(first + 1))},
        "instance_id": [f"synthetic {i}" for i in ids], "cls_id": [1] * B,
    }
    return batch, pairs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=192, help="feature-map / depth size (reference: 192)")
    ap.add_argument("--channels", type=int, default=32, help="descriptor channels (reference: 32)")
    ap.add_argument("--per-sample", action="store_true", help="reference-shaped per-sample loop (Pipeline.test_step)")
    ap.add_argument("--backbone", action="store_true", help="run a random-init Oryon.forward in front (mask = oracle)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "run_test_predictions.csv"))
    a = ap.parse_args(argv)
    dev = "cuda"
    H, C = a.size, a.channels
    args = default_args(**{"test.mask": "oracle", "model.image_encoder.img_size": [H, H], "dataset.img_size": [H, H]})
    from bench import build_solver as bench_solver          # PointDSC 12x128 (reference configuration), closed-form weights
    model = None
    if a.backbone:
        from oryon_amd.net import Oryon, default_model_args
        assert (H, C) == (192, 32), "the network emits C=32 maps at 192x192"
        torch.manual_seed(0)
        model = Oryon(default_model_args(), dev).eval()
    pipe = Pipeline(args, model=model, pointdsc_solver=bench_solver(torch.device(dev)))
    toks = torch.randint(1, 49000, (1, 80, 77), generator=torch.Generator().manual_seed(7))
    toks[..., 12], toks[..., 13:] = 49407, 0
    sphere = np.random.default_rng(0).normal(size=(512, 3))
    sphere = (0.1 * sphere / np.linalg.norm(sphere, axis=1, keepdims=True)).astype(np.float32)       # stand-in object model, 0.2 m
    rows, t0 = [], time.perf_counter()
    for first in range(0, a.pairs, a.batch):
        B = min(a.batch, a.pairs - first)
        batch, pairs = synthetic_batch(first, B, H, C, dev)
        if a.backbone:
            batch["prompt_tokens"] = toks.expand(B, 80, 77).contiguous()
            batch.pop("featmap_a"), batch.pop("featmap_q")       # Pipeline.model.forward produces them inside the step
        if a.per_sample:
            recs = pipe.test_step(batch, first // a.batch)
            pose_rel = torch.stack([r["pred_pose_rel"].cpu() for r in recs])
            status = [r["status"] for r in recs]
        else:
            out = pipe.test_step_batched(batch, first_pair_index=first)
            pose_rel, status = out["pose"].cpu(), out["status"].cpu().tolist()
            for i in range(B):
                pipe.add_pred_pose(batch["anchor"]["instance_id"][i], batch["query"]["instance_id"][i], 1.0, 1.0, pose_rel[i].numpy())
        for i in range(B):
            gt = pairs[i]["pose"].double().numpy()
            pr = pose_rel[i].double().numpy()
            theta, shift = ev.compute_RT_distances(pr, gt)
            rows.append(dict(pair=first + i, status=int(status[i]), add=float(ev.compute_add(sphere, pr, gt)),
                             adds=float(ev.compute_adds(sphere, pr, gt)), rot_deg=float(theta[0]), trans_cm=float(shift[0])))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.writelines(pipe.pred_lines)
    ok = [r for r in rows if r["status"] == 0]
    summary = {
        "pairs": len(rows), "ok": len(ok), "failures": len(rows) - len(ok), "csv": a.out, "wall_s": round(wall, 3),
        "backbone": bool(a.backbone), "loop": "per-sample (reference-shaped)" if a.per_sample else "batched",
        "ADD_mean_m": float(np.mean([r["add"] for r in ok])) if ok else None,
        "ADDS_mean_m": float(np.mean([r["adds"] for r in ok])) if ok else None,
        "ADD_0.1d_accuracy": ev.add_accuracy(np.array([r["add"] for r in rows]), np.full(len(rows), 0.2)) if rows else None,
        "rot_err_deg_max": max((r["rot_deg"] for r in ok), default=None), "trans_err_cm_max": max((r["trans_cm"] for r in ok), default=None),
    }
    print(json.dumps(summary))
    return summary


if __name__ == "__main__":
    main()
