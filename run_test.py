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

Real assets (BASELINE.json configs[2] / configs[4]; the reference's `python run_test.py -cp exp_data/baseline ...`):

    python run_test.py --data-root data --dataset nocs --split cross_scene_test --obj all --mask predicted \
        --ckpt exp_data/baseline/models/epoch=0019.ckpt --catseg pretrained_models/catseg.pth \
        --pointdsc pretrained_models/pointdsc --bpe pretrained_models/bpe_simple_vocab_16e6.txt.gz [--half-descriptors]

reads the fixed split through oryon_amd.datasets.FixedSplit (PNG decode on the host, resize / collate on the device), loads the
Lightning checkpoint's `model.*` weights into Oryon (after the CATSeg remap, net.py:102-134) and the released PointDSC weights,
runs the batched pipeline and reports ADD(S)-0.1d, rotation / translation errors and mask IoU per the reference's evaluator
(utils/evaluator.py:206-256) plus MSSD / MSPD on float16-rounded poses (oryon_pose_bop_errors, pinned to the reference's my_mssd / my_mspd by
golden G9); the prediction CSV has the reference's format.  VSD / AR need the BOP toolkit's OpenGL renderer (out of scope).

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

import oryon_amd  # noqa: E402

oryon_amd.configure()        # hardware queues for the step engine's streams (before the first HIP call)

from oryon_amd import evaluation as ev  # noqa: E402
from oryon_amd import ops  # noqa: E402
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


def load_oryon_checkpoint(model, ckpt_path: str) -> dict:
    """Weights of a reference Lightning checkpoint (`trainer.test(..., ckpt_path=args.eval.ckpt)`, run_test.py:42): the LightningModule
    keeps the network under `self.model`, so its tensors are the `model.*` entries of `state_dict`.  Returns load statistics."""
    blob = torch.load(ckpt_path, map_location="cpu")
    sd = blob.get("state_dict", blob)
    mine = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")} or dict(sd)
    res = model.load_state_dict(mine, strict=False)
    return {"tensors": len(mine), "missing": len(res.missing_keys), "unexpected": len(res.unexpected_keys)}


def hashed_prompt_tokens(prompts) -> torch.Tensor:
    """[B, 80, 77] CLIP-shaped token ids from prompt strings WITHOUT the BPE vocabulary (smoke runs only: `--hash-prompts`): words are
    hashed into the vocabulary range, <start> / <end> markers placed as the real tokenizer would.  The first entry of every prompt
    list (the bare object name) is dropped like models/vlm.py:67 does."""
    import zlib
    out = torch.zeros((len(prompts), len(prompts[0]) - 1, 77), dtype=torch.int64)
    for b, plist in enumerate(prompts):
        for t, text in enumerate(plist[1:]):
            ids = [49406] + [1 + zlib.crc32(w.encode()) % 49000 for w in text.lower().split()][:75] + [49407]
            out[b, t, :len(ids)] = torch.tensor(ids)
    return out


def run_real(a) -> dict:
    """One pass over a fixed split of REAL275 ('nocs') or TOYL with real weights: the reference's test loop
    (pipeline.py:306-355 + utils/evaluator.py:206-256) on the batched engine."""
    from oryon_amd.data import DeviceCollate
    from oryon_amd.datasets import FixedSplit, extent_diameter
    from oryon_amd.net import Oryon, default_model_args
    from oryon_amd.pointdsc import get_pointdsc_solver
    dev = "cuda"
    name = a.dataset_name or {"nocs": "nocs", "toyl": "toyl"}[a.dataset]
    split = FixedSplit(a.dataset, a.data_root, name, a.split, a.obj, mask_type=a.mask)
    margs = default_model_args()
    margs.model.use_catseg_ckpt = False
    torch.manual_seed(0)
    model = Oryon(margs, dev, bpe_path=a.bpe).eval()
    loaded = {}
    if a.catseg:
        model.load_catseg_checkpoint(a.catseg)
        loaded["catseg"] = a.catseg
    if a.ckpt:
        loaded["ckpt"] = load_oryon_checkpoint(model, a.ckpt)
    if a.pointdsc:
        solver = get_pointdsc_solver(a.pointdsc, dev)
    else:
        from bench import build_solver as bench_solver
        solver = bench_solver(torch.device(dev))
    args = default_args(**{"test.mask": a.mask, "seed": a.seed})
    pipe = Pipeline(args, model=model, pointdsc_solver=solver)
    if a.half_descriptors:
        from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
        pipe._engine = MatchPoseEngine(solver, MatchPoseConfig(dist_th=args.test.dist_th, n_corrs=args.test.n_corrs,
                                                               src_sampling=args.test.src_sampling, seed=a.seed, half_descriptors=True))
    collate = DeviceCollate(args.dataset.max_corrs, args.dataset.img_size, dev)
    n = len(split) if a.pairs <= 0 else min(a.pairs, len(split))
    from oryon_amd.evaluation import Evaluator, evaluate_batch
    # mask IoUs exist on the batched path when the model's predicted masks are evaluated (test_step_batched: 'iou_a' in out): decided here,
    # once, from the run's mode - not from whatever the first batch happened to return (and an empty split still gets its summary)
    compute_iou = not a.per_sample
    evaluator = Evaluator(exp_tag=f"{a.dataset}_{a.split}_{a.mask}", compute_iou=compute_iou)
    n_rows, t0 = 0, time.perf_counter()
    for first in range(0, n, a.batch):
        idx = list(range(first, min(first + a.batch, n)))
        batch = collate([split[i] for i in idx])
        if a.hash_prompts:
            batch["prompt_tokens"] = hashed_prompt_tokens(batch["prompt"])
        if a.per_sample:
            recs = pipe.test_step(batch, first // a.batch)
            pose_rel = torch.stack([r["pred_pose_rel"].cpu() for r in recs])
            status = [r["status"] for r in recs]
            iou = None
        else:
            out = pipe.test_step_batched(batch, first_pair_index=first)
            pose_rel, status = out["pose"].cpu(), out["status"].cpu().tolist()
            iou = (out["iou_a"].cpu(), out["iou_q"].cpu()) if "iou_a" in out else None
            for i in range(len(idx)):
                ia = float(iou[0][i]) if iou is not None else 1.0
                iq = float(iou[1][i]) if iou is not None else 1.0
                pipe.add_pred_pose(batch["anchor"]["instance_id"][i], batch["query"]["instance_id"][i], ia, iq, pose_rel[i].numpy())
        # evaluation (f3), registered exactly as the reference's test loop does (pipeline.py:313-350, utils/evaluator.py:206-338): pairs
        # that failed (no mask / no correspondences) are automatic failures with every score 0, the others are scored on
        # pred_q = pose_rel @ anchor.pose with the errors computed on the device (ADD / ADD-S with the float16 model transform,
        # rotation / translation errors, MSSD / MSPD on float16-rounded poses)
        objects = {k: split.object_info(k) for k in dict.fromkeys(batch["cls_id"])}
        if compute_iou and iou is None:       # a model without mask logits on the batched path: the reference logs IoU 1 for external masks
            iou = (torch.ones(len(idx)), torch.ones(len(idx)))
        evaluate_batch(evaluator, pred_pose_rel=pose_rel.numpy(), anchor_pose=batch["anchor"]["pose"].cpu().numpy(),
                       gt_pose=batch["query"]["pose"].cpu().numpy(), K=batch["query"]["camera"].cpu().numpy().reshape(-1, 3, 3),
                       status=[int(s_) for s_ in status], cls_ids=list(batch["cls_id"]), instance_ids=list(batch["instance_id"]),
                       objects=objects, iou_a=iou[0].numpy() if compute_iou else None, iou_q=iou[1].numpy() if compute_iou else None,
                       device=dev)
        n_rows += len(idx)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.writelines(pipe.pred_lines)
    if n_rows == 0:
        print(json.dumps({"dataset": a.dataset, "split": a.split, "obj": a.obj, "mask": a.mask, "pairs": 0, "csv": a.out,
                          "note": "the split / filter selected no pair: nothing to evaluate"}))
        return {"pairs": 0}
    means = evaluator.get_means()
    metric_file = os.path.splitext(a.out)[0] + ".json"           # what scripts/evaluation/compute_metrics.py:52,116-118 writes next to the CSV
    with open(metric_file, "w") as f:
        evaluator.save(f)
    latex = evaluator.get_latex_str()
    for line in evaluator.test_summary():                        # per-class rows (utils/evaluator.py:340-358)
        print(line)
    print(latex, end="")
    summary = {
        "dataset": a.dataset, "split": a.split, "obj": a.obj, "mask": a.mask, "pairs": n_rows,
        "Missing segm": int(sum(evaluator.counts["Missing segm"])), "Failed pose": int(sum(evaluator.counts["Failed pose"])),
        "Zero pose": int(sum(evaluator.counts["Zero pose"])),
        "ADD(S)-0.1d": means.get("ADD(S)-0.1d"), "MSSD": means.get("MSSD"), "MSPD": means.get("MSPD"),
        "R_error_deg_mean": means.get("R error"), "T_error_cm_mean": means.get("T error"),
        "recalls": {k: v for k, v in means.items() if k.startswith("Recall")}, "Mean IoU": means.get("Mean IoU"),
        "latex_row": latex.strip(), "metrics_json": metric_file,
        "pairs_per_s": n_rows / wall if wall > 0 else None, "wall_s": round(wall, 3), "csv": a.out, "weights": loaded,
        "half_descriptors": bool(a.half_descriptors),
        "not_computed": "VSD / AR (OpenGL renderer of the BOP toolkit; SURVEY.md 2.1 out of scope): the LaTeX row carries '-' there, as the "
                        "reference's own compute_vsd=False format does",
    }
    print(json.dumps(summary))
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=192, help="feature-map / depth size (reference: 192)")
    ap.add_argument("--channels", type=int, default=32, help="descriptor channels (reference: 32)")
    ap.add_argument("--per-sample", action="store_true", help="reference-shaped per-sample loop (Pipeline.test_step)")
    ap.add_argument("--backbone", action="store_true", help="run a random-init Oryon.forward in front (mask = oracle)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "run_test_predictions.csv"))
    g = ap.add_argument_group("real assets (fixed test splits + checkpoints; every path as in the reference's configs/config.yaml)")
    g.add_argument("--data-root", default=None, help="dataset.root: the folder that holds <name>/fixed_split/... (switches real-asset mode on)")
    g.add_argument("--dataset", choices=["nocs", "toyl"], default="nocs", help="REAL275 ('nocs') or TOYL")
    g.add_argument("--dataset-name", default=None, help="dataset.test.name (sub-folder of --data-root; default = --dataset)")
    g.add_argument("--split", default="cross_scene_test", help="dataset.test.split")
    g.add_argument("--obj", default="all", help="dataset.test.obj (key of object_splits.json)")
    g.add_argument("--mask", default="predicted", help="test.mask: predicted | oracle | ovseg | san | oryon")
    g.add_argument("--ckpt", default=None, help="eval.ckpt: Lightning checkpoint of the trained Oryon")
    g.add_argument("--catseg", default=None, help="pretrained_models/catseg.pth (loaded first, as net.py:102-134)")
    g.add_argument("--pointdsc", default=None, help="pretrained.pointdsc folder (snapshot/PointDSC_3DMatch_release/...)")
    g.add_argument("--bpe", default=None, help="pretrained.vocabulary: CLIP BPE file")
    g.add_argument("--half-descriptors", action="store_true", help="BASELINE configs[4]: descriptors rounded to float16 (utils/pcd.py:195-197)")
    g.add_argument("--fp16x3", action="store_true", help="towers on the fp32-grade fp16x3 kernels (backbone.enable_fp16x3): ~2.4x faster features")
    g.add_argument("--hash-prompts", action="store_true", help="smoke runs without the BPE file: hash prompt words to token ids")
    g.add_argument("--seed", type=int, default=1)
    a = ap.parse_args(argv)
    if a.fp16x3:
        from oryon_amd.backbone import enable_fp16x3
        enable_fp16x3(True)
    if a.data_root:
        return run_real(a)
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
