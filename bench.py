#!/usr/bin/env python3
"""Benchmark of the Oryon hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU B=64 synthetic 224x224 RGB-D pairs with C=256 fp32 descriptor
maps resident in HBM; one step = masks -> ROI -> subsample 5000 -> gather+normalise -> cosine NN (fp32 MFMA)
-> sample 500 correspondences -> lift -> PointDSC (12 x 128) -> pose, for every pair, plus (N>1) the
all_gather collation of poses.  Weak scaling: every rank processes its own 64 pairs.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import oryon_amd  # noqa: E402

# five HIP streams (the caller's + the engine's four) need more than the runtime's default of four hardware queues: an explicit
# process-level setting, made before anything initialises HIP (oryon_amd.configure; importing the package changes nothing)
oryon_amd.configure()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oryon_amd import ops  # noqa: E402
from oryon_amd.dist import gather_pose_windows, gather_poses, init_from_env  # noqa: E402
from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine  # noqa: E402
from oryon_amd.pointdsc import PointDSC  # noqa: E402
from oryon_amd.synth import make_pair  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0      # same table, "Peak BF16/FP16 MFMA" dense
PEAK_I8_MFMA_TOPS = 5000.0         # dense INT8 = 2x the 16-bit rate (same table: FP8 ~5 PF dense; i8 32x32x32 measured 4.4 POP/s)
PEAK_FP6_MFMA_TFLOPS = 10000.0     # same table, "Peak FP6/FP4 MFMA ~10 PF dense" (MX block-scaled only; 32x32x64 measured 8.9 PF/s)
BARE_FP6_32x32x64_TFLOPS = 5320.0     # measured: bare loop of v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3, random bits), power-limited (profiles/r05_mfma_mx_rates.md)
PEAK_HBM_BYTES = 8.0e12            # same guide: HBM3E 8 TB/s
METRIC = "image-pairs/sec end-to-end (feat+match+reg) @224², C=256; ADD(-S) parity"


def build_solver(dev):
    """PointDSC at the released 3DMatch geometry (12 layers x 128 channels, k=40, ratio 0.1; utils/pointdsc/init.py:41-50),
    random init with a fixed seed (no checkpoint is shipped / no network)."""
    g = torch.random.get_rng_state()
    torch.manual_seed(1234)
    m = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
    torch.random.set_rng_state(g)
    return m.to(dev).eval()


def make_inputs(B, H, C, first, dev):
    feats_a = torch.empty((B, C, H, H), dtype=torch.float32, device=dev)
    feats_q = torch.empty((B, C, H, H), dtype=torch.float32, device=dev)
    mask_a = torch.empty((B, H, H), dtype=torch.int32, device=dev)
    mask_q = torch.empty((B, H, H), dtype=torch.int32, device=dev)
    depth_a = torch.empty((B, H, H), dtype=torch.float32, device=dev)
    depth_q = torch.empty((B, H, H), dtype=torch.float32, device=dev)
    cam = torch.empty((B, 3, 3), dtype=torch.float64)
    pose = torch.empty((B, 4, 4), dtype=torch.float64)
    for i in range(B):
        p = make_pair(first + i, H, H, C, device=dev)
        feats_a[i], feats_q[i], mask_a[i], mask_q[i] = p["feat_a"], p["feat_q"], p["mask_a"], p["mask_q"]
        depth_a[i], depth_q[i], cam[i], pose[i] = p["depth_a"], p["depth_q"], p["camera"], p["pose"]
    return dict(feat_a=feats_a, feat_q=feats_q, mask_a=mask_a, mask_q=mask_q, depth_a=depth_a, depth_q=depth_q,
                cam=cam.to(dev), pose_gt=pose)


class MatchTimer:
    """HIP events around the DOMINANT matcher kernel launch: the events are handed to the library (oryon_profile_events) which
    records them on the launch stream right before / after that one kernel."""

    def __init__(self, name):
        self.pairs = []
        self.name = name
        self._orig = getattr(ops, name)

    def __enter__(self):
        from oryon_amd._lib import lib

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()            # materialises the hipEvent_t handles (re-recorded by the library around the kernel)
            e1.record()
            lib().oryon_profile_events(e0.cuda_event, e1.cuda_event)
            out = self._orig(*a, **k)
            self.pairs.append((e0, e1))
            return out
        setattr(ops, self.name, timed)
        return self

    def __exit__(self, *exc):
        setattr(ops, self.name, self._orig)

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(1, len(self.pairs))


def cpu_baseline(H, C, budget_rows=512):
    """Reference-form CPU path (the oracle's restatement of utils/pcd.py:28-29,202-204: [N1,N2,C] broadcast cosine,
    amin/argmin; then lift + PointDSC with bs=1 and CPU SVD) on a BOUNDED sample: one pair, `budget_rows` of the 5000
    anchor rows for the matcher (cost extrapolated linearly in rows), full lift + PointDSC."""
    from oracle import oryon_oracle as orc
    p = make_pair(0, H, H, C, device="cpu")
    roi1 = orc.roi_from_mask(p["mask_a"])
    roi2 = orc.roi_from_mask(p["mask_q"])
    n1 = min(5000, roi1.shape[0])
    f1 = orc.gather_roi_feats(p["feat_a"], roi1[:n1])
    f2 = orc.gather_roi_feats(p["feat_q"], roi2)
    t0 = time.perf_counter()
    rows = 0
    for s in range(0, budget_rows, 16):
        orc.cosine_nn_broadcast(f1[s:s + 16], f2)
        rows += 16
    t_match_rows = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.cosine_nn_gemm(f1, f2)
    t_gemm = time.perf_counter() - t0
    # lift + PointDSC on 500 true correspondences of the same pair
    md, arg = orc.cosine_nn_gemm(f1[:2000], f2)
    keep = torch.nonzero(md < 0.25).squeeze(1)[:500]
    corrs = torch.cat((roi1[:2000][keep], roi2[arg][keep]), dim=1)
    P = orc.analytic_pointdsc_params(12, 128)
    cfg = dict(num_layers=12, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
    cam = p["camera"].reshape(9)
    t0 = time.perf_counter()
    pa, pq, _ = orc.lift_pair(p["depth_a"], p["depth_q"], cam, cam, corrs, (H, H), (H, H), (H, H))
    orc.pointdsc_forward(pa, pq, P, cfg)
    t_rest = time.perf_counter() - t0
    per_pair = t_match_rows * (n1 / rows) + t_rest
    return {
        "value": 1.0 / per_pair, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port", "extrapolated": True,
        "sample": f"1 pair of the workload; reference-form broadcast matcher timed on {rows} of {n1} anchor rows x {f2.shape[0]} "
                  f"query rows x C={C} ({t_match_rows:.1f} s, extrapolated x{n1 / rows:.0f}); lift + PointDSC(12x128, n={corrs.shape[0]}) "
                  f"in full ({t_rest * 1e3:.0f} ms); same matcher as a normalised GEMM on CPU: {t_gemm:.2f} s/pair",
        "host_cpu_count": os.cpu_count(),
    }


def run_stage_set(a, rank, world, dev, stage, steps, warmup, backbone_dtype="fp32"):
    """Stage sets beyond the headline, at the reference's own shapes (224x224 RGB -> C=32 @192x192), random-init weights:
       'full'   = Oryon.forward (CLIP ViT-L/14@336 + Swin-B stages 1-2 + fusion + decoder) + predicted masks + match + lift + PointDSC
       'decode' = fusion + decoder on cached CLIP / Swin encodings, then the same (SURVEY 8d 'decode+match+pose')
    Returns the record (rank 0) or None.  Timed exactly like the headline: `steps` complete steps between two barrier + synchronize
    brackets, max over ranks; the backbone alone is timed in a second, separate loop of the same length."""
    from oryon_amd.net import Oryon, default_model_args
    B, H = a.batch, 224
    torch.manual_seed(4321 + rank)
    model = Oryon(default_model_args(), dev).eval()
    clip_values = backbone_dtype == "fp16x3-clipload"
    if clip_values:
        # the reference's CLIPEncoder holds the OpenAI checkpoint as `clip.load` builds it - fp16 Linear / Conv / in_proj / projection
        # tensors - widened with `.to(torch.float32)` and frozen (models/vlm.py:19-29): every weight IS an fp16 value.  Random-init
        # weights of that kind here; the fp16x3 linear then needs two products instead of three (same results bit for bit).
        backbone_dtype = "fp16x3"
        with torch.no_grad():
            for p_ in model.vlm.clip_model.parameters():
                if p_.dim() >= 2:
                    p_.copy_(p_.half().float())
    gen = torch.Generator(device=dev).manual_seed(99 + rank)
    rgb_a = torch.rand((B, 3, H, H), generator=gen, device=dev)
    rgb_q = torch.rand((B, 3, H, H), generator=gen, device=dev)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=torch.Generator().manual_seed(7))
    toks[..., 12] = 49407
    toks[..., 13:] = 0
    toks = toks.expand(B, 80, 77).contiguous()
    geo = [make_pair(rank * B + i, H, H, 1, device=dev) for i in range(B)]
    depth_a = torch.stack([g["depth_a"] for g in geo])
    depth_q = torch.stack([g["depth_q"] for g in geo]).clamp_min(1.0)
    cam = torch.stack([g["camera"] for g in geo]).to(dev)
    engine = MatchPoseEngine(build_solver(dev), MatchPoseConfig(dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1,
                                                                match_mode=a.match_mode))
    key = torch.arange(rank * B, rank * B + B, dtype=torch.int64, device=dev)
    xs = {"anchor": {"rgb": rgb_a}, "query": {"rgb": rgb_q}, "prompt_tokens": toks}
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if backbone_dtype == "bf16" else torch.autocast("cuda", enabled=False)
    from oryon_amd.backbone import enable_fp16x3
    enable_fp16x3(backbone_dtype == "fp16x3")                 # fp32 tensors; linears / attention on the fp16 pipe with split operands (B2-B5)
    if backbone_dtype == "bf16w":
        model = model.to(torch.bfloat16)
        xs["anchor"]["rgb"], xs["query"]["rgb"] = rgb_a.to(torch.bfloat16), rgb_q.to(torch.bfloat16)
    total = B * world
    decode_only = stage == "decode"
    if decode_only:
        # the frozen towers' outputs are inputs of this stage set: evaluated once, outside the timed region
        with torch.no_grad(), amp:
            rgb = torch.cat([xs["anchor"]["rgb"], xs["query"]["rgb"]])
            enc = (model.vlm.encode_image(rgb), model.get_guidance_embeds(rgb))
            prompt = model.vlm.encode_tokens(toks).unsqueeze(1).to(rgb.dtype)
            prompt = torch.cat([prompt, prompt])

    def backbone():
        if not decode_only:
            return model(xs)
        mask, fm = model.decoder(model.fusion(enc[0], prompt, enc[1]), enc[1])
        return {"featmap_a": fm[:B], "featmap_q": fm[B:], "mask_a": mask[:B], "mask_q": mask[B:]}

    def step():
        with torch.no_grad(), amp:
            out = backbone()
        fa, fq = out["featmap_a"].float().contiguous(), out["featmap_q"].float().contiguous()
        ma = ops.mask_from_logits(out["mask_a"].float().squeeze(1), 0.5)
        mq = ops.mask_from_logits(out["mask_q"].float().squeeze(1), 0.5)
        res = engine.run(fa, fq, ma, mq, depth_a, depth_q, cam, cam, key)
        return res, gather_poses(res["pose"], res["status"], total)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        with torch.no_grad(), amp:
            backbone()
    e1.record()
    torch.cuda.synchronize()
    bb_ms = e0.elapsed_time(e1) / steps
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    pairs_ok = int((res["status"] == 0).sum())
    del model, engine
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    # ~0.39 TFLOP per ViT-L/14@336 image, 2 images per pair; fusion + decoder alone: 2.4 + 3.3 GFLOP per image
    flops_backbone = B * 2 * (5.7e9 if decode_only else 0.39e12)
    # fp16x3: every fp32 multiply-add is three fp16 MFMA multiply-adds, so the fp32-equivalent rate is priced against a third of the
    # dense fp16 peak (the towers' linears, attention and the Swin linears run there; the convolutions still run on the fp32 pipe)
    # (fp16x3-clipload: the CLIP linears - 0.9 of the FLOPs - take two products; priced against half the fp16 peak)
    peak = {"fp32": PEAK_FP32_MFMA_TFLOPS, "fp16x3": PEAK_F16_MFMA_TFLOPS / (2.0 if clip_values else 3.0)}.get(backbone_dtype, PEAK_F16_MFMA_TFLOPS)
    achieved = flops_backbone / (bb_ms * 1e-3) / 1e12
    return {
        "metric": ("image-pairs/sec (decode+match+reg): fusion + decoder on cached CLIP / Swin encodings (-> C=32 @192x192), then match + pose"
                   if decode_only else
                   "image-pairs/sec end-to-end (feat+match+reg) at the reference's shapes (224x224 RGB -> CLIP ViT-L/14@336 + Swin-B + fusion "
                   "+ decoder -> C=32 @192x192 -> match + lift + PointDSC)"),
        "value": total * steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "backbone_ms_per_step": bb_ms, "pairs_per_gpu": B, "pairs_ok": pairs_ok,
        "dtype": {"fp32": "f32 (PyTorch-ROCm fp32 GEMMs / convolutions, TF32-style shortcuts off)",
                  "fp16x3": "f32 tensors; CLIP / Swin / fusion linears and the CLIP attention as error-compensated fp16x3 MFMA kernels "
                            "(oryon_linear_f16x3, oryon_mha_f16x3: ~1e-6 relative, fp32-grade), fused fp32 LayerNorm / window attention, the "
                            "decoder as HIP implicit-GEMM convolutions on the same fp16x3 arithmetic (oryon_decoder_forward, csrc/decoder.hip); "
                            "the remaining convolutions (patch embeddings, fusion's 7x7 / 3x3) PyTorch-ROCm fp32",
                  "bf16": "bf16 backbone GEMMs (autocast), f32 match+pose",
                  "bf16w": "bf16 backbone (weights + activations), f32 match+pose"}[backbone_dtype],
        "data": "synthetic RGB-D, random-init weights of the reference architecture (no checkpoints / network)" +
                ("; the CLIP tower's weights rounded to fp16 values in fp32 storage, as `clip.load` + `.to(torch.float32)` leaves the "
                 "reference's frozen CLIPEncoder (models/vlm.py:19-29)" if clip_values else ""),
        "prompt_cache": "80-template text tower evaluated once (identical prompt set), as in eval of one object class",
        "roofline": {"bound": "mfma",
                     "kernel": ("backbone GEMMs: oryon_linear_f16x3 / oryon_mha_f16x3 / dec_conv3x3_kernel (fp32-equivalent FLOP/s against a "
                                "third of the dense fp16 peak" + (" - here half: fp16-valued CLIP weights have no low half, two products" if clip_values else "") +
                                ") + the remaining MIOpen convolutions" if backbone_dtype == "fp16x3" else
                                "backbone GEMMs / convolutions (hipBLASLt / MIOpen through PyTorch-ROCm)"),
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None},
    }


def flush_c_stdio():
    """RCCL prints its version banner through C stdio - fully buffered when stdout is a pipe, so it used to come out when the process exits,
    BEHIND the JSON line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:           # (no libc handle: nothing buffered that we could flush)
        pass
    sys.stdout.flush()


def emit_line(rec, rank, grouped):
    """The ONE JSON line, as the last thing the job writes to stdout: every rank flushes what C stdio still holds (the banner was already
    flushed once behind init_process_group), a barrier makes sure they all have, rank 0 prints, and only then the group is torn down."""
    flush_c_stdio()
    if grouped:
        dist.barrier()
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()


def bench_full(a, rank, world, dev):
    """`--stages full|decode`: that stage set alone, as its own JSON line (never mixed into the headline `value`)."""
    rec = run_stage_set(a, rank, world, dev, a.stages, a.steps, a.warmup, a.backbone_dtype)
    if rank == 0:
        rec.update({"higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "config": {"workload": f"Batch={a.batch} synthetic 224x224 RGB-D pairs per GPU, stage set {a.stages}", "stages": a.stages}})
    emit_line(rec if rank == 0 else None, rank, world > 1)


def collation_selftest(a):
    """The N>1 control flow of main() on CPU tensors over gloo: shard keys by rank, barrier + timed region, gather_poses, one JSON
    line on rank 0.  No kernel runs; the poses are a function of the global pair index so the collated order can be checked."""
    rank, world, _ = init_from_env("cpu")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    B = a.batch
    total = B * world
    idx = torch.arange(rank * B, rank * B + B, dtype=torch.float32)
    pose = torch.eye(4).repeat(B, 1, 1)
    pose[:, 0, 3] = idx
    status = (idx.to(torch.int32) % 3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if a.collate == "final":
        # one collective for the whole window: every step's rows staged, step j's pose carries j in its y translation
        stage = torch.zeros((a.steps, B, 17))
        for j in range(a.steps):
            stage[j, :, :16] = pose.reshape(B, 16)
            stage[j, :, 7] = float(j)
            stage[j, :, 16] = status.to(torch.float32)
        allr = gather_pose_windows(stage)
        if not all(bool((allr[:, j, :, 7] == float(j)).all()) for j in range(a.steps)):
            raise SystemExit("final collation: step order lost")
        last = allr[:, a.steps - 1].reshape(total, 17).clone()
        last[:, 7] = 0.0
        allp, alls = last[:, :16].reshape(total, 4, 4), last[:, 16].to(torch.int32)
    else:
        for _ in range(a.steps):
            allp, alls = gather_poses(pose, status, total)
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = bool(torch.equal(allp[:, 0, 3], torch.arange(total, dtype=torch.float32))) and \
        bool(torch.equal(alls, torch.arange(total, dtype=torch.int32) % 3))
    emit_line({"metric": "collation selftest (no kernels)", "n_gpus": world, "steps": a.steps, "global_pairs": total,
               "collated_in_order": ok, "collate": a.collate, "collectives": 1 if a.collate == "final" else a.steps,
               "elapsed_s": float(el.item())}, rank, world > 1)           # (the same exit sequence as the measured runs)
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step (cfg2: 64)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", choices=["nchw", "nhwc"], default="nchw",
                    help="memory layout of the given descriptor maps: nchw = contiguous [B,C,H,W] as net.py:162-167 returns them (default, "
                         "the headline); nhwc = the same logical tensor in torch.channels_last storage (what a channels_last decoder emits)")
    ap.add_argument("--stages", choices=["match+pose", "decode", "full"], default="match+pose",
                    help="match+pose: BASELINE configs[1], descriptor maps given (default, the headline line); decode: fusion + decoder "
                         "forward on cached CLIP / Swin encodings, then match + pose (SURVEY 8d 'decode+match+pose'); full: random-init "
                         "CLIP ViT-L/14@336 + Swin-B + fusion + decoder forward on 224x224 RGB (-> C=32 @192x192, the reference's own "
                         "shapes), then match + pose - a separate stage set, never mixed into the headline")
    ap.add_argument("--backbone-dtype", choices=["fp32", "fp16x3", "fp16x3-clipload", "bf16", "bf16w"], default="fp32",
                    help="fp32 | bf16 (autocast over fp32 weights) | bf16w (weights converted to bf16 once)")
    ap.add_argument("--no-overlap-gather", dest="overlap_gather", action="store_false",
                    help="keep the K0 gather of step k+1 on the main stream (default: on its own stream, under the screening / registration of "
                         "step k; the inputs are resident before the timed region)")
    ap.add_argument("--overlap-gather", dest="overlap_gather", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(overlap_gather=True)
    ap.add_argument("--no-overlap", action="store_true",
                    help="do not overlap the registration of step k with the matching of step k+1 (second HIP stream)")
    ap.add_argument("--match-mode", choices=["screened", "screened16", "exact"], default="screened",
                    help="screened: fp16-MFMA screening + exact fp32 re-scoring (K1s, identical results); exact: full fp32 scan (K1)")
    ap.add_argument("--sample-first", type=int, default=0,
                    help="MatchPoseConfig.sample_first: the matcher runs on a random subset of this many anchors per pair first (identically "
                         "distributed correspondences; pairs that come up short are redone on all anchors).  0 = off (default, the headline)")
    ap.add_argument("--reps", type=int, default=7,
                    help="repetitions of the timed window of --steps steps (each bracketed by barrier + synchronize); `value` / `ms_per_step` are "
                         "the MEDIAN window, every window is listed in `timing.windows_ms_per_step`")
    ap.add_argument("--engine", choices=["native", "python"], default="native",
                    help="native: one C-ABI call per step (oryon_engine_submit: engine-owned streams / events, persistent arena, no torch "
                         "allocation per step); python: the per-call schedule of oryon_amd/engine.py (torch streams, ~40 torch allocations per step)")
    ap.add_argument("--screen", choices=["mx6", "int8"], default="mx6",
                    help="screening operands of the native engine's lazy matcher: MX-fp6 (v_mfma_scale_f32_32x32x64_f8f6f4, default) or int8")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="distinct sets of B synthetic pairs (descriptor maps, masks, depths, poses AND pair keys) the timed steps rotate through: "
                         "step k runs on set k %% N, so no step re-reads the maps its predecessor left in the Infinity Cache / L2 (VERDICT r04)")
    ap.add_argument("--stream-roles", type=int, default=-1,
                    help="placement of the engine's four HIP streams on the process's hardware queues (oryon_engine_config_t.stream_roles: four "
                         "digits = pool positions of match / gather / registration 0 / registration 1).  -1 (default): measure the candidates "
                         "during warm-up (MatchPoseEngine.tune_stream_roles, outside every timed window) and keep the best; 0: the library "
                         "default (2345); anything else: that placement")
    ap.add_argument("--collate", choices=["step", "final"], default="step",
                    help="N > 1 only.  step (default): one all_gather of [B,17] per step inside the timed region; final: every step's poses are "
                         "staged on the device and ONE all_gather per timed window collates them (north_star: 'all-gather of per-pair poses ... "
                         "only for the final collation').  The other mode is measured in extra windows and reported in `multi_gpu` as well")
    ap.add_argument("--process-group", action="store_true",
                    help="--gpus 1 only: create the RCCL process group all the same (one rank) and collate every step through it - the part of "
                         "the multi-GPU path a single-GPU box can execute (communicator created after the engine's stream pool, "
                         "all_gather_into_tensor of the [B,17] rows inside the timed region); reported in `multi_gpu`")
    ap.add_argument("--no-stage-sets", action="store_true",
                    help="skip the 'decode+match+pose' and 'full' stage sets that the default run measures after the headline")
    ap.add_argument("--collation-selftest", action="store_true",
                    help="CPU-only check of the multi-rank launch path (gloo): every rank fabricates its poses, the collation runs, rank 0 "
                         "prints one JSON line.  Used by tests/test_cabi_and_host.py; measures nothing")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (how the driver calls it): become the launcher - one rank per GPU under torch.distributed.run,
        # rendezvous on 127.0.0.1 and a free port; the ranks' stdout (rank 0's single JSON line) passes through unchanged
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    if a.collation_selftest:
        return collation_selftest(a)
    rank, world, local = init_from_env("cuda", force_group=a.process_group and a.gpus == 1)
    grouped = dist.is_initialized()
    flush_c_stdio()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, H, C = a.batch, a.size, a.channels
    if a.stages in ("full", "decode"):
        return bench_full(a, rank, world, dev)
    # N distinct input sets: set s holds the global pairs [s * world * B, (s + 1) * world * B), this rank its block of B of them
    n_sets = max(1, a.input_sets)
    # the rotating input sets must leave room for the engine's arena (K0 outputs, six result slots): at most ~40 % of the device for the maps
    # (cfg2: 4 sets = 26 GB; the cfg4 shard - 128 pairs at 384^2, C = 512 - is 77 GB per set: one set)
    set_bytes = 2 * B * C * H * H * 4 + 4 * B * H * H * 4
    n_fit = max(1, int(0.40 * torch.cuda.get_device_properties(dev).total_memory // set_bytes))
    if n_sets > n_fit:
        if rank == 0:
            print(f"bench.py: {n_sets} input sets of {set_bytes / 2**30:.0f} GiB do not fit beside the engine's arena: using {n_fit}", file=sys.stderr)
        n_sets = n_fit
    total = B * world
    sets = []
    for s_ in range(n_sets):
        d_ = make_inputs(B, H, C, first=s_ * total + rank * B, dev=dev)
        if a.layout == "nhwc":
            d_["feat_a"] = d_["feat_a"].contiguous(memory_format=torch.channels_last)
            d_["feat_q"] = d_["feat_q"].contiguous(memory_format=torch.channels_last)
        d_["cam"] = d_["cam"].reshape(B, 9).to(torch.float32).contiguous()     # the C ABI's type: fp32 [B,9] (pipeline.py:434-435 + lift_pcd)
        d_["key"] = torch.arange(s_ * total + rank * B, s_ * total + rank * B + B, dtype=torch.int64, device=dev)
        sets.append(d_)
    inputs = sets[0]
    engine = MatchPoseEngine(build_solver(dev), MatchPoseConfig(dist_th=0.25, n_corrs=500, src_sampling=5000, seed=1,
                                                                match_mode=a.match_mode, sample_first=a.sample_first),
                             overlap_registration=not a.no_overlap, overlap_gather=a.overlap_gather and not a.no_overlap,
                             native=a.engine == "native", result_views=True)
    engine.native_timing = True           # HIP events around the three sections and the screening kernel of every native step
    engine.native_geometry["screen"] = 1 if a.screen == "mx6" else 0
    if a.stream_roles > 0:
        engine.native_geometry["stream_roles"] = a.stream_roles
    key = inputs["key"]
    host = {"submit_s": 0.0, "submits": 0, "rot": 0, "set_of": {}, "last_set": 0}

    def submit(keep=False, which=None):
        idx_ = host["rot"] % n_sets if which is None else which
        d_ = sets[idx_]
        if which is None:
            host["rot"] += 1
        t_ = time.perf_counter()
        out = engine.run(d_["feat_a"], d_["feat_q"], d_["mask_a"], d_["mask_q"], d_["depth_a"],
                         d_["depth_q"], d_["cam"], d_["cam"], d_["key"], keep=keep, inputs_resident=True)
        host["submit_s"] += time.perf_counter() - t_
        host["submits"] += 1
        host["set_of"][id(out)] = idx_                      # (not a key of `out`: the engine tells keep steps from ordinary ones by its size)
        return out

    # --collate final: a window's poses are staged in [steps, B, 17] (one small device copy per step: the slot views are re-used six
    # steps later) and collated by ONE all_gather when the window ends; --collate step: one all_gather per step
    collate = {"mode": a.collate if grouped else "step", "stage": None, "k": 0}

    def collect(out):
        host["last_set"] = host["set_of"].pop(id(out), host["last_set"])
        engine.finish(out)
        if collate["mode"] == "final":
            st_ = collate["stage"]
            if st_ is None or st_.shape[0] <= collate["k"]:
                st_ = collate["stage"] = torch.zeros((max(a.steps, 64), B, 17), dtype=torch.float32, device=dev)
                collate["k"] = 0
            row = st_[collate["k"]]
            row[:, :16].copy_(out["pose"].reshape(B, 16))
            row[:, 16].copy_(out["status"])
            collate["k"] += 1
            return out, out["pose"], out["status"]
        pose, status = gather_poses(out["pose"], out["status"], total)
        return out, pose, status

    def collate_window():
        """--collate final: the one collective of a timed window - every rank's [k, B, 17] block; returns the LAST step's global poses."""
        k_ = collate["k"]
        collate["k"] = 0
        if collate["mode"] != "final" or k_ == 0:
            return None
        allr = gather_pose_windows(collate["stage"][:k_])          # [world, k, B, 17]
        last = allr[:, k_ - 1].reshape(world * B, 17)              # global pair order: rank-major blocks of B
        return last[:, :16].reshape(world * B, 4, 4).contiguous(), last[:, 16].to(torch.int32)

    def step(keep=False):
        res_ = collect(submit(keep, which=0))          # the sanity / checksum step: always input set 0
        fin_ = collate_window()
        return res_ if fin_ is None else (res_[0], fin_[0], fin_[1])

    def run_steps(n):
        """n complete steps; with overlap the registration of step k runs on a second stream under the matching of step k+1
        (software pipelining across batches): every step is still submitted, completed and collated inside the call."""
        prev, res = None, None
        for _ in range(n):
            cur = submit()
            if prev is not None:
                res = collect(prev)
            prev = cur
        if prev is not None:
            res = collect(prev)
        fin = collate_window()
        if fin is not None:
            res = (res[0], fin[0], fin[1])
        return res

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup:
        run_steps(a.warmup)
    barrier()
    screened = a.match_mode in ("screened", "screened16") and 64 < C <= 512
    use_i8 = screened and a.match_mode == "screened" and C > 128
    want_native = use_i8 and a.engine == "native"
    if want_native and engine._native is None:
        run_steps(1)                      # --warmup 0: the first step builds the engine (arena, streams); not a timed step
        barrier()
    native = engine._native if want_native else None
    # stream placement: measured on THIS process during warm-up (never inside a timed window), unless the command line fixed it
    roles_rec = None
    if native is not None and not a.no_overlap:
        if a.stream_roles < 0:
            roles_rec = engine.tune_stream_roles(run_steps, steps=12, warm=3)
            roles_rec["how"] = "tuned during warm-up: 3 + 12 steps per candidate, outside the timed windows (MatchPoseEngine.tune_stream_roles)"
            if world > 1:
                # every rank tunes on its own GPU; what rank 0 reports is its own choice, the others' are in multi_gpu.stream_roles_per_rank
                pass
        else:
            roles_rec = {"roles": native.stream_roles(), "ms_per_step": {}, "default": 2345,
                         "how": "fixed by --stream-roles" if a.stream_roles > 0 else "library default (--stream-roles 0)"}
        barrier()

    def alloc_counters():
        ms = torch.cuda.memory_stats(dev)
        return ms.get("num_device_alloc", 0), ms.get("allocation.all.allocated", 0), ms.get("num_alloc_retries", 0)

    # R repetitions of the timed window: EXACTLY --steps steps between two barrier + synchronize brackets, max over ranks; the
    # headline is the median window.  Every window also records where the host was (time inside the submit calls), what the
    # caching allocator did (hipMalloc calls / torch allocations inside the window) and, from the native engine's HIP events, how
    # long each of the three streams was busy per step - so a reader can tell a host-bound or allocator-bound window from a GPU-bound one.
    windows, sections = [], []
    with MatchTimer("match_corrs_i8" if use_i8 else "match_screened" if screened else "match") as mt:
        for _ in range(max(1, a.reps)):
            a0 = alloc_counters()
            h0 = (host["submit_s"], host["submits"], native.host_stats()[1] if native else 0.0)
            first_step = native.steps if native else 0
            barrier()
            t0 = time.perf_counter()
            out, pose, status = run_steps(a.steps)
            barrier()
            elapsed = time.perf_counter() - t0
            el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            if grouped:
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
            a1 = alloc_counters()
            w = {"ms_per_step": float(el.item()) / a.steps * 1e3, "ms_per_step_rank": elapsed / a.steps * 1e3,
                 "host_submit_ms_per_step": (host["submit_s"] - h0[0]) / max(1, host["submits"] - h0[1]) * 1e3,
                 "device_allocs": a1[0] - a0[0], "torch_allocs_per_step": (a1[1] - a0[1]) / a.steps, "alloc_retries": a1[2] - a0[2]}
            if native:
                w["host_submit_ms_per_step_c_abi"] = (native.host_stats()[1] - h0[2]) / a.steps
                sections += [native.timing(k) for k in range(max(first_step, native.steps - 64), native.steps)]
            windows.append(w)
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else None
    ms_med = med([w["ms_per_step"] for w in windows])
    elapsed = ms_med * 1e-3 * a.steps

    # result sanity + algorithmic work of the dominant kernel (this rank's launch)
    out, pose, status = step(keep=True)
    torch.cuda.synchronize()
    n_a, n_q = out["n_a"].double(), out["n_q"].double()
    flops = float((2.0 * n_a * n_q * C).sum())
    match_ms = med([t["screen_kernel_ms"] for t in sections]) if sections else mt.mean_ms()
    match_ms_mean = sum(t["screen_kernel_ms"] for t in sections) / len(sections) if sections else match_ms
    ok = status[:total] == 0
    gt = inputs["pose_gt"].to(torch.float32)
    mine = out["pose"].cpu()
    st_local = out["status"].cpu() == 0
    rot_err = (mine[:, :3, :3] - gt[:, :3, :3]).abs().amax(dim=(1, 2))[st_local]
    trans_err = (mine[:, :3, 3] - gt[:, :3, 3]).abs().amax(dim=1)[st_local]

    # collated result of the last complete step (identical on every rank): a checksum of the pose bytes in global pair order, so that a
    # sharded run can be compared bit for bit with a single-GPU run over the same global pairs (tests/test_gpu_bench_contract.py)
    import hashlib
    pose_sha = hashlib.sha256(pose[:total].contiguous().cpu().numpy().tobytes() + status[:total].contiguous().cpu().numpy().tobytes()).hexdigest()
    multi = None
    if grouped:
        # the only collective of the path: one all_gather of [B_r,17] fp32 per step (pose + status), timed on its own with HIP events;
        # and every rank's own throughput over the median window
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gather_poses(out["pose"], out["status"], total)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(20):
            gather_poses(out["pose"], out["status"], total)
        ev1.record()
        torch.cuda.synchronize()
        mine_ms = torch.tensor([med([w["ms_per_step_rank"] for w in windows])], dtype=torch.float64, device=dev)
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        my_roles = torch.tensor([native.stream_roles() if native is not None else 0], dtype=torch.int64, device=dev)
        all_roles = [torch.zeros_like(my_roles) for _ in range(world)]
        dist.all_gather(all_roles, my_roles)
        # the OTHER collation mode in three extra windows of the same length (same engine, same inputs), so that one line carries both
        other = "final" if collate["mode"] == "step" else "step"
        keep_mode, collate["mode"] = collate["mode"], other
        run_steps(3)
        other_ms = []
        for _ in range(3):
            barrier()
            t0_ = time.perf_counter()
            run_steps(a.steps)
            barrier()
            el_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
            dist.all_reduce(el_, op=dist.ReduceOp.MAX)
            other_ms.append(float(el_.item()) / a.steps * 1e3)
        collate["mode"] = keep_mode
        other_med = sorted(other_ms)[1]
        multi = {"backend": "nccl (RCCL over xGMI)" if world > 1 else "nccl (RCCL, ONE rank: --process-group on a single-GPU box; no link is crossed)",
                 "rccl_ranks": world, "all_gather_us": ev0.elapsed_time(ev1) / 20 * 1e3,
                 "all_gather_bytes_per_rank": B * 17 * 4, "per_rank_pairs_per_s": [B * 1e3 / float(t.item()) for t in all_ms],
                 "collate": keep_mode,
                 "collectives_per_step": 1 if keep_mode == "step" else 1.0 / a.steps,
                 f"collate_{other}_ms_per_step": other_med, f"collate_{other}_pairs_per_s": total * 1e3 / other_med,
                 f"collate_{keep_mode}_ms_per_step": ms_med, f"collate_{keep_mode}_pairs_per_s": total * 1e3 / ms_med,
                 "stream_roles_per_rank": [int(t.item()) for t in all_roles],
                 "stream_pool_created_before_process_group": True}

    # the dominant kernel WITHOUT the other streams beside it: a few steps of a serial engine (everything on one stream) in this same
    # process, the same HIP events around the same launch.  The pipelined number above is what the kernel costs inside the step (it
    # shares CUs, power and HBM with K0 and the registration); this one is the kernel itself and is what a rocprofv3 kernel trace of
    # this command reports (the profiler serialises kernels of different queues)
    unshared_ms = k0_unshared_ms = None
    if native is not None:
        ser = MatchPoseEngine(engine.solver, engine.cfg, overlap_registration=False, overlap_gather=False, native=True, result_views=True)
        ser.native_timing = True
        ser.native_geometry["screen"] = engine.native_geometry["screen"]
        for _ in range(6):
            ser.run(inputs["feat_a"], inputs["feat_q"], inputs["mask_a"], inputs["mask_q"], inputs["depth_a"], inputs["depth_q"],
                    inputs["cam"], inputs["cam"], key, inputs_resident=True)
        torch.cuda.synchronize()
        ts_ = [ser._native.timing(k)["screen_kernel_ms"] for k in range(1, 6)]
        unshared_ms = sum(ts_) / len(ts_)
        tg_ = [ser._native.gather_ms(k) for k in range(1, 6)]             # the two gather launches (ROI kernels excluded)
        k0_unshared_ms = sum(tg_) / len(tg_)
        del ser
        torch.cuda.empty_cache()

    if rank == 0:
        cp = 32 if C <= 32 else 64 if C <= 64 else 128 if C <= 128 else 256 if C <= 256 else (C + 31) // 32 * 32
        if use_i8:
            kernel, peak = f"match_i8_screen_v2_kernel<{cp}, 0, {8 if cp == 256 else 4}> (int8-MFMA pre-screen of K1s8)", PEAK_I8_MFMA_TOPS
        elif screened:
            kernel, peak = f"match_f16_screen_kernel<{max(cp, 128)},2> (fp16-MFMA screening pass of K1s)", PEAK_F16_MFMA_TFLOPS
        elif cp <= 256:
            kernel, peak = f"match_f32_regb_kernel<{cp}>", PEAK_FP32_MFMA_TFLOPS
        else:
            kernel, peak = "match_f32_kernel (LDS-staged, wide descriptors)", PEAK_FP32_MFMA_TFLOPS
        from oryon_amd._lib import lib as _L
        dispatched = _L().oryon_dominant_kernel().decode()          # what the library actually launched between the events
        op = "int8 multiply-accumulate, 2 ops each, i32 accumulate" if use_i8 else "floating-point fma, 2 flops each"
        if "mx6" in dispatched:
            kernel, peak = dispatched + " (MX-fp6 MFMA screen of the lazy matcher: v_mfma_scale_f32_32x32x64_f8f6f4, e2m3 operands)", PEAK_FP6_MFMA_TFLOPS
            op = "fp6 (e2m3, block-scaled) multiply-accumulate, 2 flops each, f32 accumulate"
        elif dispatched:
            kernel = dispatched + kernel[kernel.index(" ("):] if " (" in kernel else dispatched
        launch_ms = match_ms_mean                                   # average launch duration over the timed windows (HIP events)
        achieved = flops / (launch_ms * 1e-3) / 1e12
        # the other roofline of SURVEY.md 8(d): operand bytes the kernel has to read once (rows actually used, in the kernel's
        # operand type) + its per-anchor outputs, against the 8 TB/s HBM peak - far from binding for ROIs of this size
        opb = 1 if use_i8 else 2 if screened else 4
        alg_bytes = float((opb * cp * (n_a + n_q) + 12.0 * n_a).sum())
        hbm_frac = alg_bytes / (launch_ms * 1e-3) / PEAK_HBM_BYTES
        # HBM bytes per launch cannot be counted from inside the process (PMC needs rocprofv3 around it): the value comes from the committed
        # rocprofv3 --pmc passes of this exact workload AND this exact kernel source - the record carries the sha256 of the kernel's
        # source file, and a mismatch (the kernel changed since the counters were collected) reports null instead of a stale number
        traffic, traffic_src = None, "no PMC record for this workload"
        if use_i8 and (B, H, C) == (64, 224, 256) and a.layout == "nchw":
            import glob
            import hashlib
            with open(os.path.join(ROOT, "oryon_amd", "csrc", "screen_mx6.hip" if a.screen == "mx6" else "match16.hip"), "rb") as fh:
                sha = hashlib.sha256(fh.read()).hexdigest()
            want = "mx6" if a.screen == "mx6" else "i8"
            for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):      # newest round first
                with open(tpath) as fh:
                    tj = json.load(fh)
                if want not in tj.get("kernel", ""):
                    continue
                if tj.get("kernel_source_sha256") == sha:
                    traffic, traffic_src = tj["traffic_bytes_per_launch"], tj.get("source", os.path.basename(tpath))
                else:
                    # loud, not silent: the committed counters describe an OLDER build of the kernel
                    traffic = "stale"
                    traffic_src = (f"profiles/{os.path.basename(tpath)} is STALE: the screen kernel source changed since the PMC passes were "
                                   "collected (sha256 mismatch) - re-run tools/collect_profiles.sh + tools/make_traffic_json.py")
                break
        # K0 (the one HBM-bound kernel pair of the matcher: gather + normalise + MX-fp6 conversion of the ROI rows, csrc/gather8.hip) against
        # the 8 TB/s HBM peak.  Algorithmic bytes per step (SURVEY 8d "descriptors actually used" + what the pass has to write):
        #   reads  4 C (n_a + n_q)                            the ROI rows' fp32 descriptors, once
        #   writes n_q (256 + 4) + n_a (256 + 4 + 4 c_pad)    mx6 operand row + norm per row; anchors also their fp32 unit row (the exact re-scoring's operand)
        # `moved` scales that by the ratio the rocprofv3 PMC passes of this workload counted (FETCH_SIZE x 2 + WRITE_SIZE vs algorithmic:
        # DRAM delivers whole sectors and the ROIs are not dense - DESIGN.md "K0"), source named below
        k0 = None
        if k0_unshared_ms is not None and "mx6" in dispatched:
            k0_alg = float((4.0 * C * (n_a + n_q) + 260.0 * n_q + (260.0 + 4.0 * cp) * n_a).sum())
            k0_pipe_ms = (sum(t["gather_ms"] for t in sections) / len(sections)) if sections else None     # (whole section, ROI kernels included)
            K0_MOVED_OVER_ALGORITHMIC = 5.61 / 3.97        # profiles/r05_pmc_counters.md (cfg2, K0v4): 4.51 GB fetched + 1.10 GB written for 3.97 GB
            k0 = {"kernels": "gather_mx6_v4_kernel x2 (queries, anchors) of one step, HIP events around the two launches on a serial engine",
                  "algorithmic_bytes_per_step": k0_alg, "unshared_ms": k0_unshared_ms, "gather_section_in_pipeline_ms": k0_pipe_ms,
                  "algorithmic_frac": k0_alg / (k0_unshared_ms * 1e-3) / PEAK_HBM_BYTES,
                  "moved_frac": k0_alg * K0_MOVED_OVER_ALGORITHMIC / (k0_unshared_ms * 1e-3) / PEAK_HBM_BYTES,
                  "moved_over_algorithmic": K0_MOVED_OVER_ALGORITHMIC if (B, H, C) == (64, 224, 256) else None,
                  "moved_source": "profiles/r05_pmc_counters.md / r06_pmc_counters.md (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of the two gather launches)",
                  "peak_bytes_per_s": PEAK_HBM_BYTES}
        rec = {
            "metric": METRIC, "value": total * a.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "multi_gpu": multi,
            "timing": {
                "statistic": f"median of {len(windows)} windows of {a.steps} steps, each bracketed by barrier + torch.cuda.synchronize, max over ranks",
                "windows_ms_per_step": [round(w["ms_per_step"], 4) for w in windows],
                "engine": ("native: one oryon_engine_submit per step (engine-owned HIP streams / events, persistent arena)" if native else
                           "python: per-call schedule (torch streams, torch allocations per step)"),
                "host_submit_ms_per_step": med([w["host_submit_ms_per_step"] for w in windows]),
                "host_submit_ms_per_step_c_abi": med([w["host_submit_ms_per_step_c_abi"] for w in windows]) if native else None,
                "torch_allocs_per_step": [round(w["torch_allocs_per_step"], 2) for w in windows],
                "device_allocs_in_window": [w["device_allocs"] for w in windows],
                "alloc_retries_in_window": [w["alloc_retries"] for w in windows],
                "stream_busy_ms_per_step": ({k: med([t[k] for t in sections]) for k in ("gather_ms", "match_ms", "screen_kernel_ms", "registration_ms")}
                                            if sections else None),
                "step_timeline_ms": ({k: med([t[k] for t in sections]) for k in ("match_start", "match_end", "registration_start", "registration_end")}
                                     if sections else None),
                "step_latency_ms": med([t["registration_end"] for t in sections]) if sections else None,
                "timed_steps_with_events": len(sections),
                "stream_roles": roles_rec["roles"] if roles_rec else None,
                "stream_roles_tuning": roles_rec,
            },
            "config": {
                "workload": f"{'cfg2' if (H, C) == (224, 256) else 'cfg4 geometry' if (H, C) == (384, 512) else 'custom'}: Batch={B} synthetic {H}x{H} pairs per GPU, C={C} fp32 descriptors given (HIP matcher + lift + "
                            f"PointDSC 12x128), N1<=5000, n_corrs=500",
                "stages": "match+lift+registration (descriptor maps resident in HBM; backbone not in the timed region)",
                "descriptor_layout": "NCHW contiguous fp32 (as Oryon.forward returns them)" if a.layout == "nchw" else "channels_last (NHWC storage) fp32",
                "match_mode": a.match_mode + ((" (MX-fp6 MFMA screen with a proven bound, exact fp32 re-scoring of the sampled anchors' candidates; anchors the bound cannot settle are resolved exactly: outputs identical to the fp32 scan)" if native is not None and engine.native_geometry.get("screen", 0) == 1 else " (int8-MFMA pre-screen, fp16-MFMA screening of the undecided anchors, exact fp32 re-scoring: outputs identical to the fp32 scan)") if use_i8 else " (fp16-MFMA screening, exact fp32 re-scoring: outputs identical to the fp32 scan)" if screened else ""),
                "sample_first": a.sample_first or None,
                "input_sets": f"{n_sets} distinct sets of {B} pairs per GPU (maps, masks, depths, poses, pair keys), step k runs on set k % {n_sets}; "
                              "sanity / pose_sha256 on set 0",
                "pairs_per_gpu": B, "global_pairs": total, "parallelism": f"pairs sharded over {world} GPU(s), all_gather of poses",
                "pipelining": ("none" if a.no_overlap else "registration of step k on a second HIP stream under the matching of step k+1"
                               + ("; K0 (ROI + gather) of step k+1 on a third stream under the screening / registration of step k"
                                  if a.overlap_gather else "")),
                "pose_sha256": pose_sha, "first_pair": 0,
                "pairs_ok": int(ok.sum()), "max_rot_err_vs_gt": float(rot_err.max()) if rot_err.numel() else None,
                "max_trans_err_m_vs_gt": float(trans_err.max()) if trans_err.numel() else None,
            },
            "roofline": {
                "bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "op": op,
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                "flops_per_launch": flops, "avg_launch_ms": launch_ms,
                "algorithmic_bytes_per_launch": alg_bytes, "hbm_frac": hbm_frac,
                "share_of_step": match_ms / (elapsed / a.steps * 1e3),
                "measured": "HIP events on the launch stream around every launch of the kernel inside the timed windows (pipelined: K0 of the "
                            "next step and the registration of the previous ones run beside it)",
                # the same figures as scalars (a parser that keeps only scalar entries of `roofline` still sees them)
                "unshared_avg_launch_ms": unshared_ms,
                "unshared_frac": None if unshared_ms is None else flops / (unshared_ms * 1e-3) / 1e12 / peak,
                "frac_of_bare_loop_rate": (flops / (unshared_ms * 1e-3) / 1e12 / BARE_FP6_32x32x64_TFLOPS) if (unshared_ms is not None and "mx6" in dispatched) else None,
                "k0_algorithmic_frac": k0["algorithmic_frac"] if k0 else None,
                "k0_moved_frac": k0["moved_frac"] if k0 else None,
                "k0_unshared_ms": k0_unshared_ms,
                "k0": k0,
                "unshared": (None if unshared_ms is None else
                             {"avg_launch_ms": unshared_ms, "achieved": flops / (unshared_ms * 1e-3) / 1e12,
                              "frac": flops / (unshared_ms * 1e-3) / 1e12 / peak,
                              # what a bare register-resident loop of the same instruction sustains on this part under its power limit
                              # (tools/probe_mfma_mx_rates.hip, profiles/r05_mfma_mx_rates.md: fp6 e2m3 32x32x64, random operand bits)
                              "frac_of_bare_loop_rate": (flops / (unshared_ms * 1e-3) / 1e12 / BARE_FP6_32x32x64_TFLOPS) if "mx6" in dispatched else None,
                              "bare_loop_tflops": BARE_FP6_32x32x64_TFLOPS if "mx6" in dispatched else None,
                              "measured": "same events, 5 steps of a serial engine (one stream, nothing beside the kernel) in this run: the "
                                          "figure a rocprofv3 kernel trace of this command shows, since the profiler serialises the queues"}),
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(H, C)
    # the optional "sample first" schedule on the same inputs (not the headline: it evaluates the matcher on a random 1024-anchor subset per
    # pair first - the 500 sampled correspondences are identically distributed, but the validity of the other anchors is never computed)
    sfirst = None
    if not a.no_stage_sets and not a.sample_first and use_i8:
        engine.cfg.sample_first = 1024
        run_steps(3)
        barrier()
        t0 = time.perf_counter()
        SF_STEPS = 30
        sout, spose, sstatus = run_steps(SF_STEPS)
        barrier()
        sel = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if grouped:
            dist.all_reduce(sel, op=dist.ReduceOp.MAX)
        engine.cfg.sample_first = 0
        if rank == 0:
            smine = sout["pose"].cpu()
            sok = sout["status"].cpu() == 0
            gt = sets[host["last_set"]]["pose_gt"].to(torch.float32)          # the input set the last step ran on
            sfirst = {"schedule": "matcher on a uniformly random 1024-anchor subset per pair first (>= 500 valid rows there give an identically "
                                  "distributed sample of the 500 correspondences); pairs that come up short are redone on all anchors, gated on "
                                  "the device.  NOT the headline: the default route settles the validity of all <= 5000 anchors like the reference",
                      "value": total * SF_STEPS / float(sel.item()), "unit": "pairs/s", "ms_per_step": float(sel.item()) / SF_STEPS * 1e3, "steps": SF_STEPS,
                      "pairs_ok": int((sstatus[:total] == 0).sum()),
                      "max_rot_err_vs_gt": float((smine[:, :3, :3] - gt[:, :3, :3]).abs().amax(dim=(1, 2))[sok].max()),
                      "max_trans_err_m_vs_gt": float((smine[:, :3, 3] - gt[:, :3, 3]).abs().amax(dim=1)[sok].max())}
    # the headline's int8 stage decides everything on the generator's Gaussian descriptors; report the same step on a HARD distribution too:
    # smooth low-rank descriptor fields (neighbouring pixels nearly parallel - every anchor has many near-ties), where the int8 bound
    # cannot separate the candidates
    hard = None
    if not a.no_stage_sets and (H, C) == (224, 256):
        yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, H, device=dev), indexing="ij")
        coef = torch.stack([torch.ones_like(xx), xx, yy, xx * yy, torch.sin(3 * xx), torch.cos(3 * yy), torch.sin(7 * yy), torch.cos(5 * xx)])
        for s_, d_ in enumerate(sets):                # every input set gets its own smooth fields (same rotation as the headline)
            gen = torch.Generator(device=dev).manual_seed(77 + rank + 1000 * s_)
            basis = torch.randn((B, C, coef.shape[0]), generator=gen, device=dev)
            d_["feat_q"].copy_(torch.einsum("bck,khw->bchw", basis, coef))
            d_["feat_q"].add_(0.02 * torch.randn(d_["feat_q"].shape, generator=gen, device=dev))
            d_["feat_a"].copy_(d_["feat_q"]).add_(0.01 * torch.randn(d_["feat_a"].shape, generator=gen, device=dev))
        torch.cuda.synchronize()                      # the gather stream reads the maps as soon as a step is submitted: finish rewriting them first
        engine.collect_i8_stats = True                # report the int8 stage's undecided fraction on these inputs (asynchronous, no sync)
        run_steps(3)                                  # lets the asynchronous statistics arrive
        barrier()
        t0 = time.perf_counter()
        HARD_STEPS = 40                               # enough steps for the pipeline's fill and drain not to dominate the figure
        hout, _, hstatus = run_steps(HARD_STEPS)
        barrier()
        hel = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if grouped:
            dist.all_reduce(hel, op=dist.ReduceOp.MAX)
        if rank == 0:
            hard = {"descriptors": "smooth rank-8 fields + 1-2 % noise (anchor map = query map + noise): the int8 bound settles every anchor's VALIDITY "
                                   "but cannot separate its near-ties, so the argmin of the <= 500 sampled anchors per pair comes from the second level: the "
                                   "fp16x3 scan of just those rows (K1x3, match_x3.hip: seeded hi-only sweep, compensated products on the flagged "
                                   "(64-anchor group, tile) jobs) + the canonical fp32 chain on its few candidates; the engine's K0 pass writes the "
                                   "hi / lo rows itself once its feedback says the previous steps needed them (x3_prefetch) "
                                   "(ORYON_AMB_X3=0: exact fp32 scan against fp32 query rows materialised for the pair)",
                    "value": total * HARD_STEPS / float(hel.item()), "unit": "pairs/s", "ms_per_step": float(hel.item()) / HARD_STEPS * 1e3, "steps": HARD_STEPS,
                    "int8_undecided_fraction": float(engine._i8_frac), "int8_stage_skipped": bool(engine._i8_frac > engine.i8_max_undecided),
                    "k0_passes_with_hi_lo_rows": (engine._native.x3_steps() if engine._native is not None else None),
                    "fraction_of_headline": total * HARD_STEPS / float(hel.item()) / rec["value"],
                    "pairs_ok": int((hstatus[:total] == 0).sum())}
    # the other two stage sets of SURVEY 8(d), measured in this same run (10 steps each) and carried in the same line; `value`
    # stays the configs[1] number (descriptors given)
    stage_recs = {}
    if not a.no_stage_sets:
        del inputs, engine, sets
        torch.cuda.empty_cache()
        for stage, bdt, label in (("decode", "fp16x3", "decode+match+pose"),
                                  ("decode", "fp32", "decode+match+pose, torch / MIOpen fp32 modules"),
                                  ("full", "fp32", "full (feat+match+pose), fp32 torch linears"),
                                  ("full", "fp16x3", "full (feat+match+pose), fp16x3 linears"),
                                  ("full", "fp16x3-clipload", "full (feat+match+pose), fp16x3 linears, CLIP weights fp16-valued as `clip.load` leaves them"),
                                  ("full", "bf16w", "full (feat+match+pose), bf16 backbone (weights + activations) - NOT fp32-grade: for the record only")):
            r = run_stage_set(a, rank, world, dev, stage, steps=10, warmup=2, backbone_dtype=bdt)
            if rank == 0:
                stage_recs[label] = r
    if rank == 0:
        rec["stages"] = stage_recs or None
        rec["hard_descriptors"] = hard
        rec["sample_first_schedule"] = sfirst
        # the other measured rates as scalars of `config` (kept by parsers that drop nested records)
        rec["config"]["hard_pairs_per_s"] = hard["value"] if hard else None
        rec["config"]["hard_fraction_of_headline"] = hard["fraction_of_headline"] if hard else None
        full_g = stage_recs.get("full (feat+match+pose), fp16x3 linears")
        full_c = stage_recs.get("full (feat+match+pose), fp16x3 linears, CLIP weights fp16-valued as `clip.load` leaves them")
        dec_ = stage_recs.get("decode+match+pose")
        rec["config"]["full_fp32grade_pairs_per_s"] = full_g["value"] if full_g else None
        rec["config"]["full_fp32grade_clipload_pairs_per_s"] = full_c["value"] if full_c else None
        rec["config"]["decode_match_pose_pairs_per_s"] = dec_["value"] if dec_ else None
        rec["config"]["stream_roles"] = rec["timing"]["stream_roles"]
    emit_line(rec if rank == 0 else None, rank, grouped)


if __name__ == "__main__":
    main()
