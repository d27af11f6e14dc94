"""GPU tests of the callers (oryon_amd.pipeline.Pipeline / engine): reference-shaped per-sample loop, batched
engine, failure statuses, partition invariance (sharded == unsharded bit for bit)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _solver(L=2, C=32):
    from oracle import oryon_oracle as orc
    from oryon_amd.pointdsc import PointDSC
    m = PointDSC(in_dim=6, num_layers=L, num_channels=C, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
    m.load_state_dict(orc.analytic_pointdsc_params(L, C), strict=True)
    return m.cuda().eval()


def _batch(idx, H=48, C=32, dev="cuda"):
    from oryon_amd.synth import make_pair
    pairs = [make_pair(i, H, H, C) for i in idx]
    B = len(pairs)
    st = lambda k: torch.stack([p[k] for p in pairs])
    anchor_pose = torch.eye(4).repeat(B, 1, 1)
    anchor_pose[:, :3, 3] = torch.tensor([0.01, -0.02, 0.8])
    batch = {
        "featmap_a": st("feat_a").to(dev), "featmap_q": st("feat_q").to(dev),
        "anchor": {"mask": st("mask_a").to(torch.uint8), "orig_depth": [p["depth_a"] for p in pairs], "camera": st("camera"),
                   "pose": anchor_pose, "instance_id": [f"s {i} a" for i in idx], "sizes": torch.tensor([[H, H]] * B)},
        "query": {"mask": st("mask_q").to(torch.uint8), "orig_depth": [p["depth_q"] for p in pairs], "camera": st("camera"),
                  "pose": st("pose").float(), "instance_id": [f"s {i} q" for i in idx], "sizes": torch.tensor([[H, H]] * B)},
        "instance_id": [f"s {i}" for i in idx], "cls_id": [1] * B,
    }
    return batch, pairs


def _pipeline(H=48):
    from oryon_amd.pipeline import Pipeline, default_args
    args = default_args(**{"test.mask": "oracle", "model.image_encoder.img_size": [H, H], "dataset.img_size": [H, H]})
    return Pipeline(args, pointdsc_solver=_solver())


def test_test_step_reference_shaped_loop():
    pl = _pipeline()
    batch, pairs = _batch([3, 4, 5])
    batch["query"]["mask"][1] = 0                      # pair 1: empty query mask -> invalid detection
    torch.manual_seed(1)
    recs = pl.test_step(batch, 0)
    assert [r["status"] for r in recs] == [0, 1, 0]
    assert torch.equal(recs[1]["pred_pose_rel"], torch.eye(4)) and recs[1]["pred_pose"] is None
    assert len(pl.pred_lines) == 3 and pl.pred_lines[1].split(",")[2].split(" ")[0] == "1.0"
    for b in (0, 2):
        T = recs[b]["pred_pose_rel"].numpy()
        gt = pairs[b]["pose"].numpy()
        assert np.abs(T[:3, :3] - gt[:3, :3]).max() < 1e-2 and np.abs(T[:3, 3] - gt[:3, 3]).max() < 5e-3
        np.testing.assert_allclose(recs[b]["pred_pose"].numpy(), T @ batch["anchor"]["pose"][b].numpy(), atol=1e-6)


def test_unknown_solver_raises_runtime_error():
    pl = _pipeline()
    pl.args.test.solver = "ransac"
    batch, _ = _batch([3])
    with pytest.raises(RuntimeError):
        pl.get_pose(batch, torch.zeros((500, 4), dtype=torch.int64), 0)


def test_batched_step_statuses_and_ground_truth():
    pl = _pipeline()
    batch, pairs = _batch([3, 4, 5, 6])
    batch["anchor"]["mask"][2] = 0                     # NO_MASK
    batch["featmap_q"][3] = 0.0                          # null descriptors: cos = 0 -> dist 0.5, nothing under 0.25 -> NO_CORR
    out = pl.test_step_batched(batch)
    st = out["status"].cpu().tolist()
    assert st == [0, 0, 1, 2]
    eye = torch.eye(4)
    assert torch.equal(out["pose"][2].cpu(), eye) and torch.equal(out["pose"][3].cpu(), eye)
    for b in (0, 1):
        T = out["pose"][b].cpu().numpy()
        gt = pairs[b]["pose"].numpy()
        assert np.abs(T[:3, :3] - gt[:3, :3]).max() < 1e-2 and np.abs(T[:3, 3] - gt[:3, 3]).max() < 5e-3
    np.testing.assert_allclose(out["pred_q"][0].cpu().numpy(), out["pose"][0].cpu().numpy() @ batch["anchor"]["pose"][0].numpy(), atol=1e-6)


def test_partition_invariance_bit_for_bit():
    """Processing pairs [0..5] in one batch == processing them as the shards a 2- or 3-rank job would own
    (global pair index keys the device RNG)."""
    from oryon_amd.dist import shard_range
    pl = _pipeline()
    idx = list(range(10, 16))
    batch, _ = _batch(idx)
    full = pl.test_step_batched(batch, first_pair_index=10)
    for world in (2, 3):
        poses = []
        for r in range(world):
            s, e = shard_range(len(idx), r, world)
            sub, _ = _batch(idx[s:e])
            poses.append(pl.test_step_batched(sub, first_pair_index=10 + s)["pose"])
        assert torch.equal(torch.cat(poses), full["pose"])


def test_predicted_mask_path():
    from oryon_amd.pipeline import Pipeline, default_args
    H = 48
    args = default_args(**{"test.mask": "predicted", "model.image_encoder.img_size": [H, H]})
    pl = Pipeline(args, pointdsc_solver=_solver())
    batch, pairs = _batch([3, 4])
    for key, mk in (("mask_a", "anchor"), ("mask_q", "query")):
        batch[key] = (batch[mk]["mask"].float().cuda() * 8.0 - 4.0)[:, None]      # logits: +4 inside, -4 outside
    res = pl.mask_results(batch, pl.model.forward(batch))
    assert torch.equal(res["mask_a"].cpu(), batch["anchor"]["mask"].int())
    assert float(res["iou_a"].min()) == 1.0
    out = pl.test_step_batched(batch)
    assert out["status"].cpu().tolist() == [0, 0]


@pytest.mark.parametrize("H,C", [(64, 160), (48, 512)])         # 160 pads to 256 channels; 512 = BASELINE cfg4 width
def test_engine_screened_equals_exact_bit_for_bit(H, C):
    """The fp16-screened matcher must leave every downstream result untouched: same correspondences, same poses."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(20, 24)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    solver = _solver()
    outs = []
    for mode in ("exact", "screened"):
        eng = MatchPoseEngine(solver, MatchPoseConfig(match_mode=mode))
        outs.append(eng.run(st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"),
                            st("camera").to(dev), st("camera").to(dev), keep=True))
    a, b = outs
    assert torch.equal(a["status"], b["status"]) and a["status"].tolist() == [0, 0, 0, 0]
    assert torch.equal(a["n_valid"], b["n_valid"])
    assert torch.equal(a["corrs"], b["corrs"])
    assert torch.equal(a["pcd_a"], b["pcd_a"]) and torch.equal(a["pcd_q"], b["pcd_q"])
    assert torch.equal(a["pose"], b["pose"])


def test_oryon_backbone_into_batched_pipeline_gpu():
    """Rows a1-a5 on the GPU feeding the HIP path: Oryon.forward (shallow CLIP for speed) -> predicted masks -> match -> pose."""
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    from oryon_amd.pipeline import Pipeline, default_args
    from oryon_amd.synth import make_pair
    dev = "cuda"
    torch.manual_seed(0)
    net = Oryon(default_model_args(), dev, clip_cfg=CLIPConfig(v_layers=2, t_layers=1)).eval()
    B, H = 2, 224
    geo = [make_pair(i, H, H, 1) for i in range(B)]
    toks = torch.randint(1, 49000, (1, 80, 77)); toks[..., 9] = 49407; toks[..., 10:] = 0
    batch = {"anchor": {"rgb": torch.rand(B, 3, H, H), "orig_depth": [g["depth_a"] for g in geo], "camera": torch.stack([g["camera"] for g in geo]),
                        "pose": torch.eye(4).repeat(B, 1, 1), "instance_id": ["a0", "a1"]},
             "query": {"rgb": torch.rand(B, 3, H, H), "orig_depth": [g["depth_q"].clamp_min(1.0) for g in geo],
                       "camera": torch.stack([g["camera"] for g in geo]), "instance_id": ["q0", "q1"]},
             "prompt_tokens": toks.expand(B, 80, 77).contiguous(), "instance_id": ["p0", "p1"]}
    with torch.no_grad():
        out = net(batch)
    assert tuple(out["featmap_a"].shape) == (B, 32, 192, 192) and out["featmap_a"].device.type == "cuda"
    pl = Pipeline(default_args(**{"test.mask": "predicted"}), model=net, pointdsc_solver=_solver())
    with torch.no_grad():
        res = pl.test_step_batched(batch)
    assert tuple(res["pose"].shape) == (B, 4, 4) and torch.isfinite(res["pose"]).all()
    assert set(res["status"].cpu().tolist()) <= {0, 1, 2}


def test_engine_overlap_stream_gives_identical_results():
    """Registration on a second HIP stream (pipelined submission) must not change any result."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    H, C = 48, 32
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(30, 36)]
    st = lambda k, sl: torch.stack([p[k] for p in pairs[sl]])
    solver = _solver()
    args = lambda sl: (st("feat_a", sl), st("feat_q", sl), st("mask_a", sl), st("mask_q", sl), st("depth_a", sl), st("depth_q", sl),
                       st("camera", sl).to(dev), st("camera", sl).to(dev))
    ref = MatchPoseEngine(solver, MatchPoseConfig())
    r0 = ref.run(*args(slice(0, 3)), torch.arange(0, 3, device=dev))
    r1 = ref.run(*args(slice(3, 6)), torch.arange(3, 6, device=dev))
    eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=True)
    o0 = eng.run(*args(slice(0, 3)), torch.arange(0, 3, device=dev))
    o1 = eng.run(*args(slice(3, 6)), torch.arange(3, 6, device=dev))      # queued before o0 is collected
    eng.finish(o0)
    eng.finish(o1)
    torch.cuda.synchronize()
    assert torch.equal(o0["pose"], r0["pose"]) and torch.equal(o1["pose"], r1["pose"])
    assert torch.equal(o0["status"], r0["status"]) and torch.equal(o1["status"], r1["status"])


def test_engine_gather_stream_gives_identical_results():
    """overlap_gather (K0 of the next batch on its own stream, under the screening / registration of the current one) + overlap_registration
    over four back-to-back batches at the int8 route's size (C = 256), inputs handed over both as resident and behind a producer event:
    every pose / status / count equals the plain engine's, bit for bit."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    H, C = 64, 256
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(50, 58)]
    st = lambda k, sl: torch.stack([p[k] for p in pairs[sl]])
    solver = _solver()
    args = lambda sl: (st("feat_a", sl), st("feat_q", sl), st("mask_a", sl), st("mask_q", sl), st("depth_a", sl), st("depth_q", sl),
                       st("camera", sl).to(dev), st("camera", sl).to(dev))
    slices = [slice(0, 2), slice(2, 4), slice(4, 6), slice(6, 8)]
    ref = MatchPoseEngine(solver, MatchPoseConfig())
    want = [ref.run(*args(sl), torch.arange(sl.start, sl.stop, device=dev)) for sl in slices]
    eng = MatchPoseEngine(solver, MatchPoseConfig(), overlap_registration=True, overlap_gather=True)
    ins = [args(sl) for sl in slices]
    torch.cuda.synchronize()
    outs = []
    for i, sl in enumerate(slices):
        if i % 2 == 0:
            outs.append(eng.run(*ins[i], torch.arange(sl.start, sl.stop, device=dev), inputs_resident=True))
        else:
            ev = torch.cuda.Event()
            ev.record()                                   # "the producer finished these maps here"
            outs.append(eng.run(*ins[i], torch.arange(sl.start, sl.stop, device=dev), inputs_event=ev))
    for o in outs:
        eng.finish(o)
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        for k in ("pose", "status", "n_valid", "n_lifted"):
            assert torch.equal(o[k], w[k]), k
    # the sample-first schedule allocates more under the gather stream (the first-stage ROI): same check, many batches in flight
    cfg = MatchPoseConfig(sample_first=256)
    ref = MatchPoseEngine(solver, cfg)
    want = [ref.run(*ins[i % 4], torch.arange(8 * i, 8 * i + 2, device=dev)) for i in range(12)]
    eng = MatchPoseEngine(solver, cfg, overlap_registration=True, overlap_gather=True)
    torch.cuda.synchronize()
    outs = [eng.run(*ins[i % 4], torch.arange(8 * i, 8 * i + 2, device=dev), inputs_resident=True) for i in range(12)]
    for o in outs:
        eng.finish(o)
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        for k in ("pose", "status", "n_lifted"):
            assert torch.equal(o[k], w[k]), k


def test_cfg4_sized_pair_recovers_ground_truth():
    """BASELINE cfg4 geometry (384x384 maps, C=512) for one rank's worth of two pairs: the pose must match the generator's."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    pairs = [make_pair(i, 384, 384, 512, device=dev) for i in (40, 41)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    eng = MatchPoseEngine(_solver(), MatchPoseConfig())
    out = eng.run(st("feat_a"), st("feat_q"), st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"),
                  st("camera").to(dev), st("camera").to(dev))
    assert out["status"].tolist() == [0, 0]
    for b in range(2):
        T, gt = out["pose"][b].cpu().numpy(), pairs[b]["pose"].cpu().numpy()
        assert np.abs(T[:3, :3] - gt[:3, :3]).max() < 1e-2 and np.abs(T[:3, 3] - gt[:3, 3]).max() < 5e-3


def test_quick_gelu_bf16_fused():
    from oryon_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (8 * 1000, 8 * 1000 + 5, 3):
        x = (4.0 * torch.randn(n, generator=g, device="cuda")).to(torch.bfloat16)
        y = ops.quick_gelu_bf16(x)
        xf = x.float()
        ref = (xf * torch.sigmoid(1.702 * xf)).to(torch.bfloat16)           # fp32 arithmetic, one rounding
        diff = (y.float() - ref.float()).abs()
        assert float((diff / ref.float().abs().clamp_min(1e-3)).max()) < 1e-2     # at most one bf16 ulp (fast exp)
        assert float((y.float() - ref.float()).abs().mean()) < 1e-4


def test_add_layernorm_bf16_fused():
    """B2 against torch's own bf16 chain (add rounded to bf16, nn.LayerNorm with fp32 statistics): at most one bf16 ulp apart."""
    import torch.nn.functional as F
    from oryon_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    for rows, D in ((577 * 3, 1024), (77 * 5, 768), (7, 8), (130, 4096), (1001, 128), (333, 256), (50, 200), (9, 512)):
        x = (3.0 * torch.randn(rows, D, generator=g, device="cuda")).to(torch.bfloat16)
        d = torch.randn(rows, D, generator=g, device="cuda").to(torch.bfloat16)
        w = (1.0 + 0.2 * torch.randn(D, generator=g, device="cuda")).to(torch.bfloat16)
        b = (0.1 * torch.randn(D, generator=g, device="cuda")).to(torch.bfloat16)
        s_ref = x + d
        h_ref = F.layer_norm(s_ref.float(), (D,), w.float(), b.float(), 1e-5)
        s, h = ops.add_layernorm_bf16(x, d.clone(), w, b, 1e-5)
        assert torch.equal(s, s_ref)
        err = (h.float() - h_ref).abs()
        assert float((err / h_ref.abs().clamp_min(0.5)).max()) < 2.0 ** -8        # bf16 rounding of an fp32-accurate value
        s2, h2 = ops.add_layernorm_bf16(x, None, w, b, 1e-5)
        assert s2.data_ptr() == x.data_ptr()
        h2_ref = F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
        assert float(((h2.float() - h2_ref).abs() / h2_ref.abs().clamp_min(0.5)).max()) < 2.0 ** -8


def test_clip_bf16_fused_blocks_match_unfused():
    """The fused bf16 residual path of the CLIP towers (B1 + B2) against the same weights run through the plain torch blocks."""
    from oryon_amd.backbone.clip import CLIP, CLIPConfig
    torch.manual_seed(0)
    cfg = CLIPConfig(image_size=56, patch=14, v_width=256, v_layers=3, v_heads=4, embed_dim=64, t_width=128, t_layers=2, t_heads=4)
    m = CLIP(cfg).cuda().eval().to(torch.bfloat16)
    x = torch.randn(4, 17, 256, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        fused = m.visual.transformer(x)
        plain = m.visual.transformer.resblocks(x)
        rgb = torch.rand(2, 3, 56, 56, device="cuda").to(torch.bfloat16)
        toks = m.patch_tokens(rgb)
    assert fused.shape == plain.shape and toks.shape == (2, 256, 4, 4)
    rel = (fused.float() - plain.float()).norm() / plain.float().norm()
    assert float(rel) < 2e-2                                   # bf16 round-off through three blocks
    # autocast over fp32 weights must keep to torch's own ops (the fused kernels take bf16 parameters)
    m32 = CLIP(cfg).cuda().eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        t32 = m32.patch_tokens(torch.rand(2, 3, 56, 56, device="cuda"))
    assert t32.shape == (2, 256, 4, 4) and bool(torch.isfinite(t32.float()).all())


def test_swin_window_attention_kernel_vs_torch_fp32():
    """B3 against the torch formulation of the same block (pad, roll, partition, bias, mask, softmax, merge) evaluated in fp32 on the
    same bf16 parameters and inputs: the kernel's only rounding is the final bf16 one."""
    from oryon_amd.backbone import swin
    torch.manual_seed(1)
    for dim, heads, H, W in ((128, 4, 20, 17), (256, 8, 14, 14), (64, 2, 7, 9)):
        for shift in (0, 3):
            att = swin._WindowAttention(dim, heads, 7, shift).cuda().eval()
            with torch.no_grad():
                att.relative_position_bias_table.normal_(std=0.5)
                att.qkv.bias.normal_(std=0.3)
            att16 = att.to(torch.bfloat16)
            x = torch.randn(2, H, W, dim, device="cuda").to(torch.bfloat16)
            with torch.no_grad():
                fused = att16(x)
                ref_mod = swin._WindowAttention(dim, heads, 7, shift).cuda().eval()
                ref_mod.load_state_dict({k: v.float() for k, v in att16.state_dict().items()})
                ref = ref_mod(x.float())
            assert fused.dtype == torch.bfloat16 and fused.shape == ref.shape
            err = (fused.float() - ref).abs().max() / ref.abs().max()
            assert float(err) < 2e-2, (dim, heads, H, W, shift, float(err))


def test_oryon_forward_bf16_weights_tracks_fp32():
    """Oryon.forward with bf16 weights (PatchEmbed GEMM + B1 / B2 / B3 kernels in the towers) against the same random-init network in
    fp32 (torch ops only): descriptor maps and mask logits stay strongly correlated - bf16 round-off, not a different function."""
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    torch.manual_seed(0)
    net = Oryon(default_model_args(), "cuda", clip_cfg=CLIPConfig(v_layers=2, t_layers=1)).eval()
    gen = torch.Generator().manual_seed(0)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 10] = 49407
    toks[..., 11:] = 0
    rgb_a, rgb_q = torch.rand(2, 3, 224, 224, generator=gen).cuda(), torch.rand(2, 3, 224, 224, generator=gen).cuda()
    xs = {"anchor": {"rgb": rgb_a}, "query": {"rgb": rgb_q}, "prompt_tokens": toks.expand(2, 80, 77).contiguous()}
    with torch.no_grad():
        ref = net(xs)
        net16 = net.to(torch.bfloat16)
        net16.vlm._prompt_cache.clear()
        out = net16({"anchor": {"rgb": rgb_a.to(torch.bfloat16)}, "query": {"rgb": rgb_q.to(torch.bfloat16)}, "prompt_tokens": xs["prompt_tokens"]})
    for k in ("featmap_a", "featmap_q", "mask_a", "mask_q"):
        a, b = ref[k].float().flatten(), out[k].float().flatten()
        assert out[k].dtype == torch.bfloat16 and bool(torch.isfinite(b).all())
        corr = torch.corrcoef(torch.stack((a, b)))[0, 1]
        assert float(corr) > 0.98, (k, float(corr))


def test_swin_bf16_fast_layernorm_matches_plain():
    import torch.nn as nn
    from oryon_amd.backbone import swin
    torch.manual_seed(0)
    m = swin.SwinGuidance().cuda().eval().to(torch.bfloat16)
    img = torch.randn(2, 3, 112, 112, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        fast = m(img)
        keep = swin._fast_ln
        swin._fast_ln = lambda x: False
        try:
            plain = m(img)
        finally:
            swin._fast_ln = keep
    for k in fast:
        rel = (fast[k].float() - plain[k].float()).norm() / plain[k].float().norm()
        assert fast[k].shape == plain[k].shape and float(rel) < 2e-2, (k, float(rel))


def test_engine_int8_stage_backs_off_on_ambiguous_descriptors():
    """Descriptor maps made of a few hundred distinct vectors (every anchor has many near-ties): the int8 pre-screen cannot decide
    them, the engine notices (async read-back) and skips it for the following batches - results stay those of the exact matcher."""
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    H, C = 48, 256
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(30, 32)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    g = torch.Generator(device=dev).manual_seed(9)
    base = torch.randn(C, 150, generator=g, device=dev)
    idx = torch.randint(0, 150, (2, H * H), generator=g, device=dev)
    fq = torch.stack([base[:, idx[b]].reshape(C, H, H) for b in range(2)])
    fa = fq + 0.01 * torch.randn(fq.shape, generator=g, device=dev)
    solver = _solver()
    ref = MatchPoseEngine(solver, MatchPoseConfig(match_mode="exact")).run(fa, fq, st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"),
                                                                          st("camera").to(dev), st("camera").to(dev), keep=True)
    eng = MatchPoseEngine(solver, MatchPoseConfig())
    eng.i8_max_undecided = 0.25            # the back-off is off by default since round 2 (lazy tail); keep=True below takes the eager route it serves
    for it in range(4):
        out = eng.run(fa, fq, st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"), st("camera").to(dev), st("camera").to(dev), keep=True)
        torch.cuda.synchronize()
        assert torch.equal(out["corrs"], ref["corrs"]) and torch.equal(out["status"], ref["status"])
    assert eng._i8_frac > eng.i8_max_undecided and eng._i8_skipped >= 1


def test_run_test_driver_batched_and_per_sample(tmp_path):
    """run_test.py (the build's counterpart of the reference's test entry point) on synthetic pairs: both loops recover the ground
    truth and write one CSV line per pair in the reference's format."""
    import run_test
    from oryon_amd.evaluation import read_pred_csv
    for extra in ([], ["--per-sample"]):
        out = str(tmp_path / ("pred" + "_".join(extra) + ".csv"))
        s = run_test.main(["--pairs", "4", "--batch", "2", "--size", "64", "--channels", "32", "--out", out] + extra)
        assert s["pairs"] == 4 and s["failures"] == 0 and s["rot_err_deg_max"] < 1.0 and s["trans_err_cm_max"] < 1.0
        assert s["ADD_0.1d_accuracy"] == 1.0
        assert len(read_pred_csv(out)) == 4


def test_engine_half_descriptor_mode_matches_reference_half_branch():
    """BASELINE configs[4] / utils/pcd.py:195-197: descriptors rounded to float16 before the matcher.  The batched engine in
    half_descriptors mode (int8 route: K0v3 rounds on the way in; and the exact route on pre-rounded maps) must reproduce, row for row,
    what the per-sample facade's corrs_device='cuda' branch computes (exact fp32 scan of the rounded descriptors) - that branch is
    pinned to the reference's own half-precision pdist by tests/golden/g1_matcher_half.npz."""
    from oryon_amd import pcd
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    C, H = 256, 48
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(3)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    fa, fq = st("feat_a") * 3.7, st("feat_q") * 0.31               # scales that make the half rounding visible in different binades
    solver = _solver()
    cam = st("camera").to(dev)
    outs = {}
    for mode in ("screened", "exact"):
        eng = MatchPoseEngine(solver, MatchPoseConfig(match_mode=mode, half_descriptors=True, src_sampling=None))
        outs[mode] = eng.run(fa, fq, st("mask_a"), st("mask_q"), st("depth_a"), st("depth_q"), cam, cam, keep=True)
    torch.cuda.synchronize()
    for b in range(3):
        pre = pcd.match_presample(fa[b], fq[b], pairs[b]["mask_a"], pairs[b]["mask_q"], 0.25, half_descriptors=True)
        n = int(outs["exact"]["n_a"][b])
        v = pre["valid"]
        assert int(v.sum()) > 100
        for mode in ("screened", "exact"):
            o = outs[mode]
            assert torch.equal(o["valid"][b, :n].bool(), v), mode
            assert torch.equal(o["argmin"][b, :n][v].long(), pre["argmin"][v]), mode
            assert torch.equal(o["min_dist"][b, :n][v].view(torch.int32), pre["min_dist"][v].view(torch.int32)), mode
        # and the rounding is not a no-op on this input: the fp32 branch gives different distances
        ref32 = pcd.match_presample(fa[b], fq[b], pairs[b]["mask_a"], pairs[b]["mask_q"], 0.25)
        assert not torch.equal(ref32["min_dist"][v], pre["min_dist"][v])
    assert torch.equal(outs["screened"]["pose"], outs["exact"]["pose"])


def test_engine_sample_first_schedule():
    """MatchPoseConfig(sample_first=N): the matcher runs on a random N-anchor subset first.  Pairs whose subset holds >= n_corrs valid rows
    must return n_corrs distinct correspondences, each an (anchor of the pair's ROI, exact argmin of that anchor) pair of a valid row;
    pairs whose subset comes up short are redone on all anchors and must equal the default schedule bit for bit (same RNG keys); the
    statuses of degenerate pairs are unchanged; poses recover the ground truth."""
    from oryon_amd import ops
    from oryon_amd.engine import MatchPoseConfig, MatchPoseEngine
    from oryon_amd.synth import make_pair
    dev = "cuda"
    C, H = 256, 64
    pairs = [make_pair(i, H, H, C, device=dev) for i in range(5)]
    st = lambda k: torch.stack([p[k] for p in pairs])
    fa, fq, ma, mq = st("feat_a").clone(), st("feat_q").clone(), st("mask_a").clone(), st("mask_q").clone()
    ma[:] = 1                                               # 4096 anchors per pair
    fq[1] = torch.randn_like(fq[1])                         # pair 1: nothing matches -> NO_CORR through the second stage
    ma[2] = 0                                               # pair 2: NO_MASK
    g = torch.Generator(device=dev).manual_seed(4)
    fa[3] = torch.randn(fa[3].shape, generator=g, device=dev)
    fa[3, :, :3, :] = fq[3, :, 5:8, :] + 0.05 * torch.randn((C, 3, H), generator=g, device=dev)     # pair 3: 192 valid anchors only
    solver = _solver()
    cam = st("camera").to(dev)
    run = lambda cfg: MatchPoseEngine(solver, cfg).run(fa, fq, ma, mq, st("depth_a"), st("depth_q"), cam, cam, keep=False)
    ref = MatchPoseEngine(solver, MatchPoseConfig()).run(fa, fq, ma, mq, st("depth_a"), st("depth_q"), cam, cam, keep=True)
    out = run(MatchPoseConfig(sample_first=1024))
    dflt = run(MatchPoseConfig())
    torch.cuda.synchronize()
    assert out["status"].tolist() == ref["status"].tolist() == [0, 2, 1, 0, 0]
    # recompute the sampled rows of the sample-first run through the op layer to look at them (the engine does not return them)
    eng = MatchPoseEngine(solver, MatchPoseConfig(sample_first=1024))
    roi_a, n_a = ops.roi_compact(ma)
    ops.roi_subsample_(roi_a, n_a, 5000, 1, torch.arange(5, dtype=torch.int64, device=dev))
    valid_ref, argmin_ref, roi_q = ref["valid"], ref["argmin"], ref["roi_q"]
    for b in (0, 4):                                        # pairs served by the first stage
        assert int(ref["n_valid"][b]) > 2000 and int(out["n_valid"][b]) >= 500 and int(out["n_valid"][b]) < int(ref["n_valid"][b])
    for b in (1, 3):                                        # pairs redone on all anchors: identical to the default schedule
        assert int(out["n_valid"][b]) == int(ref["n_valid"][b]) == int(dflt["n_valid"][b])
        assert torch.equal(out["pose"][b], dflt["pose"][b])
    for b in (0, 3, 4):
        gt = pairs[b]["pose"].float()
        T = out["pose"][b].cpu()
        if b != 3:
            assert float((T[:3, :3] - gt[:3, :3]).abs().max()) < 1e-2 and float((T[:3, 3] - gt[:3, 3]).abs().max()) < 5e-3


def test_registration_is_bit_stable_beside_another_registration():
    """Two solver instances on two streams, the same 64 x 500 correspondences: every pose of the first equals its serial result bit for
    bit while the second one's encoder (fp16x3 MFMA kernels) and whole registration run beside it.  With packed fp32 VALU ops in the build
    this fails within a few iterations (a wave's v_pk_*_f32 results go wrong in lanes 48-63 next to another kernel's double-rate MFMAs,
    DESIGN.md "Concurrency and the packed-fp32 finding"); the library is built without them (csrc/Makefile NOPK)."""
    from oryon_amd.pointdsc import PointDSC
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)                     # the released 3DMatch geometry: 12 layers x 128 channels (the fp16x3 MFMA kernels' case)
    m = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1).to(dev).eval()
    noise = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    B = 64
    src = torch.rand(B, 512, 3, generator=g, device=dev)
    tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
    n = torch.full((B,), 500, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    feat, conf = [x.clone() for x in m.encode(src, tgt, n)]
    seeds, ns = m.pick_seeds_batched(src, conf, n)
    sT, _, best = m.hypotheses(src, tgt, feat, n, seeds, ns)
    T0 = sT[torch.arange(B, device=dev), best.long()].contiguous()
    want_refine = m.refine(src, tgt, n, T0).clone()
    want_pose = m.register(src, tgt, n, status)[0].clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for it in range(25):
        with torch.cuda.stream(sb):
            for _ in range(2):
                noise.encode(src, tgt, n)
            noise.register(src, tgt, n, status)
        with torch.cuda.stream(sa):
            refined = [m.refine(src, tgt, n, T0) for _ in range(20)]
            poses = [m.register(src, tgt, n, status)[0] for _ in range(2)]
        torch.cuda.synchronize()
        assert all(torch.equal(o, want_refine) for o in refined), f"refine differs beside another registration (iteration {it})"
        assert all(torch.equal(o, want_pose) for o in poses), f"register differs beside another registration (iteration {it})"
