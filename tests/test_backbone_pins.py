"""Numeric pins of the two third-party towers.  Their source (openai `clip`,
torchvision `swin_b`) is not in the reference tree, so parity with the reference is formally unpinned; what CAN be pinned here is
that this build's restatement computes the published architectures: the same parameters are copied into the independent
`transformers` implementations (CLIPModel, SwinModel) and the outputs the reference consumes are compared.

  * Swin-B guidance tower (net.py:45-75): stage-1 output, first and second patch-merging outputs - the three feature-extractor
    nodes - on CPU at 112x112 (no window padding) and 100x100 (25x25 tokens: padding to 28, odd-size patch merging), and on the
    GPU at the reference's 384x384 (96x96 tokens -> padded to 98).
  * CLIP at the ViT-L/14@336 widths (1024 / 16 heads / patch 14 / 577 positions; text 768 / 12 heads / ctx 77 / vocab 49408)
    with 2 layers per tower, on the GPU.
Bar: <= 2e-4 relative (the two implementations order their fp32 sums differently)."""
import pytest
import torch


def _swin_pair(size, device="cpu", seed=0):
    tr = pytest.importorskip("transformers")
    from oryon_amd.backbone.swin import SwinGuidance
    torch.manual_seed(seed)
    m = SwinGuidance().eval()
    for p in m.parameters():
        p.data.normal_(0, 0.3 if p.dim() == 1 else 0.05)
    # three stages so that stage 2 (index 1) owns a patch-merging layer; the third stage's block is never looked at
    cfg = tr.SwinConfig(image_size=size, patch_size=4, embed_dim=128, depths=[2, 2, 1], num_heads=[4, 8, 16], window_size=7, num_channels=3,
                        drop_path_rate=0.0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = tr.SwinModel(cfg).eval()
    P, sd = dict(hf.named_parameters()), m.state_dict()

    def cp(dst, src):
        assert P[dst].shape == sd[src].shape, (dst, src)
        P[dst].data.copy_(sd[src])
    cp("embeddings.patch_embeddings.projection.weight", "features.0.0.weight")
    cp("embeddings.patch_embeddings.projection.bias", "features.0.0.bias")
    cp("embeddings.norm.weight", "features.0.2.weight")
    cp("embeddings.norm.bias", "features.0.2.bias")
    for st, (f, dim) in enumerate(((1, 128), (3, 256))):
        for b in range(2):
            pre, src = f"encoder.layers.{st}.blocks.{b}.", f"features.{f}.{b}."
            w, bias = sd[src + "attn.qkv.weight"], sd[src + "attn.qkv.bias"]
            names = ("q_proj", "k_proj", "v_proj") if pre + "attention.q_proj.weight" in P else ("self.query", "self.key", "self.value")
            for i, n in enumerate(names):
                P[pre + f"attention.{n}.weight"].data.copy_(w[i * dim:(i + 1) * dim])
                P[pre + f"attention.{n}.bias"].data.copy_(bias[i * dim:(i + 1) * dim])
            o = "attention.o_proj" if pre + "attention.o_proj.weight" in P else "attention.output.dense"
            cp(pre + o + ".weight", src + "attn.proj.weight")
            cp(pre + o + ".bias", src + "attn.proj.bias")
            tbl = [k for k in P if k.startswith(pre) and k.endswith("relative_position_bias_table")][0]
            cp(tbl, src + "attn.relative_position_bias_table")
            fc1 = "mlp.fc1" if pre + "mlp.fc1.weight" in P else "intermediate.dense"
            fc2 = "mlp.fc2" if pre + "mlp.fc2.weight" in P else "output.dense"
            for a, bn in (("layernorm_before", "norm1"), ("layernorm_after", "norm2"), (fc1, "mlp.0"), (fc2, "mlp.3")):
                cp(pre + a + ".weight", src + bn + ".weight")
                cp(pre + a + ".bias", src + bn + ".bias")
        pre, src = f"encoder.layers.{st}.downsample.", f"features.{f + 1}."
        cp(pre + "reduction.weight", src + "reduction.weight")
        cp(pre + "norm.weight", src + "norm.weight")
        cp(pre + "norm.bias", src + "norm.bias")
    return m.to(device), hf.to(device)


def _swin_check(size, device, B=2):
    m, hf = _swin_pair(size, device)
    x = torch.randn(B, 3, size, size, device=device)
    with torch.no_grad():
        mine = m(x)
        emb, dims = hf.embeddings(x)
        before = hf.encoder(emb, dims, output_hidden_states=True, output_hidden_states_before_downsampling=True).reshaped_hidden_states
        after = hf.encoder(emb, dims, output_hidden_states=True, output_hidden_states_before_downsampling=False).reshaped_hidden_states
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    # reshaped_hidden_states are NCHW; index 0 is the stem, 1 / 2 the first two stages
    ref3 = before[1].permute(0, 2, 3, 1)                # stage 1 before merging     = features.1.1.add_1
    ref2 = after[1].permute(0, 2, 3, 1)                 # after the first merging    = features.2.reduction
    ref1 = after[2].permute(0, 2, 3, 1)                 # after the second merging   = features.4.reduction
    assert mine["guidance3"].shape == ref3.shape and mine["guidance2"].shape == ref2.shape and mine["guidance1"].shape == ref1.shape
    errs = (rel(mine["guidance3"], ref3), rel(mine["guidance2"], ref2), rel(mine["guidance1"], ref1))
    assert max(errs) < 2e-4, errs
    return errs


@pytest.mark.parametrize("size", [112, 100])
def test_swin_guidance_matches_transformers_cpu(size):
    _swin_check(size, "cpu")


@pytest.mark.gpu
def test_swin_guidance_matches_transformers_gpu_384():
    torch.backends.cuda.matmul.allow_tf32 = False
    _swin_check(384, "cuda", B=2)


@pytest.mark.gpu
def test_clip_vit_l14_336_widths_match_transformers_gpu():
    """Two-layer towers at the real ViT-L/14@336 widths: patch tokens after ln_post (what vlm.py:46-59 hands to the fusion) and the
    projected EOT text embedding (vlm.py:74-83)."""
    tr = pytest.importorskip("transformers")
    from oryon_amd.backbone.clip import CLIP, CLIPConfig
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = "cuda"
    cfg = CLIPConfig.vit_l14_336()
    cfg.v_layers, cfg.t_layers = 2, 2
    torch.manual_seed(0)
    m = CLIP(cfg).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.2)
    hf_cfg = tr.CLIPConfig(
        text_config=dict(vocab_size=cfg.vocab, hidden_size=cfg.t_width, intermediate_size=4 * cfg.t_width, num_hidden_layers=2,
                         num_attention_heads=cfg.t_heads, max_position_embeddings=cfg.ctx, hidden_act="quick_gelu", eos_token_id=cfg.vocab - 1,
                         bos_token_id=cfg.vocab - 2, pad_token_id=0),
        vision_config=dict(hidden_size=cfg.v_width, intermediate_size=4 * cfg.v_width, num_hidden_layers=2, num_attention_heads=cfg.v_heads,
                           image_size=cfg.image_size, patch_size=cfg.patch, hidden_act="quick_gelu"),
        projection_dim=cfg.embed_dim)
    hf = tr.CLIPModel(hf_cfg).eval()
    sd, P = m.state_dict(), dict(hf.named_parameters())

    def copy_block(prefix_hf, prefix, width):
        w, b = sd[prefix + ".attn.in_proj_weight"], sd[prefix + ".attn.in_proj_bias"]
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            P[f"{prefix_hf}.self_attn.{n}.weight"].data.copy_(w[i * width:(i + 1) * width])
            P[f"{prefix_hf}.self_attn.{n}.bias"].data.copy_(b[i * width:(i + 1) * width])
        for a, c in ((".self_attn.out_proj", ".attn.out_proj"), (".layer_norm1", ".ln_1"), (".layer_norm2", ".ln_2"),
                     (".mlp.fc1", ".mlp.c_fc"), (".mlp.fc2", ".mlp.c_proj")):
            P[prefix_hf + a + ".weight"].data.copy_(sd[prefix + c + ".weight"])
            P[prefix_hf + a + ".bias"].data.copy_(sd[prefix + c + ".bias"])
    with torch.no_grad():
        P["vision_model.embeddings.patch_embedding.weight"].copy_(sd["visual.conv1.weight"])
        P["vision_model.embeddings.class_embedding"].copy_(sd["visual.class_embedding"])
        P["vision_model.embeddings.position_embedding.weight"].copy_(sd["visual.positional_embedding"])
        name_pre = "vision_model.pre_layrnorm" if "vision_model.pre_layrnorm.weight" in P else "vision_model.pre_layernorm"
        P[name_pre + ".weight"].copy_(sd["visual.ln_pre.weight"]); P[name_pre + ".bias"].copy_(sd["visual.ln_pre.bias"])
        P["vision_model.post_layernorm.weight"].copy_(sd["visual.ln_post.weight"]); P["vision_model.post_layernorm.bias"].copy_(sd["visual.ln_post.bias"])
        P["text_model.embeddings.token_embedding.weight"].copy_(sd["token_embedding.weight"])
        P["text_model.embeddings.position_embedding.weight"].copy_(sd["positional_embedding"])
        P["text_model.final_layer_norm.weight"].copy_(sd["ln_final.weight"]); P["text_model.final_layer_norm.bias"].copy_(sd["ln_final.bias"])
        P["text_projection.weight"].copy_(sd["text_projection"].T)
        for i in range(2):
            copy_block(f"vision_model.encoder.layers.{i}", f"visual.transformer.resblocks.{i}", cfg.v_width)
            copy_block(f"text_model.encoder.layers.{i}", f"transformer.resblocks.{i}", cfg.t_width)
    m, hf = m.to(dev), hf.to(dev)
    img = torch.randn(2, 3, cfg.image_size, cfg.image_size, device=dev)
    toks = torch.randint(1, cfg.vocab - 2, (6, cfg.ctx), device=dev)
    toks[:, 0] = cfg.vocab - 2
    for r, e in enumerate((5, 9, 76, 3, 40, 12)):
        toks[r, e] = cfg.vocab - 1                           # EOT = highest id -> argmax position (vlm.py:81)
        toks[r, e + 1:] = 0
    g = cfg.image_size // cfg.patch
    with torch.no_grad():
        ours_v = m.patch_tokens(img)
        hv = hf.vision_model(pixel_values=img).last_hidden_state
        ref_v = hf.vision_model.post_layernorm(hv[:, 1:, :]).transpose(1, 2).reshape(2, cfg.v_width, g, g)
        ours_t = m.text_features(toks)
        ref_t = hf.get_text_features(input_ids=toks, attention_mask=torch.ones_like(toks))
        if not torch.is_tensor(ref_t):
            ref_t = ref_t.pooler_output if hasattr(ref_t, "pooler_output") else ref_t[0]
    assert tuple(ours_v.shape) == (2, 1024, 24, 24) and tuple(ours_t.shape) == (6, 768)
    assert float((ours_v - ref_v).abs().max()) < 2e-4 * float(ref_v.abs().max())
    assert float((ours_t - ref_t).abs().max()) < 2e-4 * float(ref_t.abs().max())


@pytest.mark.gpu
def test_oryon_forward_gpu_equals_cpu_and_prompt_cache():
    """Rows a1 / f2 on the MI355X: the whole Oryon.forward (CLIP at ViT-L/14@336 widths with 2 layers per tower for speed, Swin-B
    stages 1-2, fusion, decoder; random init) evaluated by PyTorch-ROCm must reproduce the CPU fp32 evaluation of the same module -
    descriptor maps and mask logits <= 1e-4 relative (the north-star descriptor bar) - and the prompt-embedding cache must serve the
    second batch without touching the text tower."""
    import copy
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    cfg = CLIPConfig.vit_l14_336()
    cfg.v_layers, cfg.t_layers = 2, 2
    torch.manual_seed(0)
    cpu = Oryon(default_model_args(), "cpu", clip_cfg=cfg).eval()
    gpu = copy.deepcopy(cpu)
    gpu.device = "cuda"
    gpu.vlm.device = "cuda"
    gpu = gpu.to("cuda").eval()
    B = 2
    gen = torch.Generator().manual_seed(3)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 11] = 49407
    toks[..., 12:] = 0
    xs = {"anchor": {"rgb": torch.rand(B, 3, 224, 224, generator=gen)}, "query": {"rgb": torch.rand(B, 3, 224, 224, generator=gen)},
          "prompt_tokens": toks.expand(B, 80, 77).contiguous()}
    xg = {"anchor": {"rgb": xs["anchor"]["rgb"].cuda()}, "query": {"rgb": xs["query"]["rgb"].cuda()}, "prompt_tokens": xs["prompt_tokens"]}
    with torch.no_grad():
        ref = cpu(xs)
        out = gpu(xg)
    for k in ("featmap_a", "featmap_q", "mask_a", "mask_q"):
        err = float((out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max())
        assert err < 1e-4, (k, err)
    # f2: a second batch with the same prompt set is served from the cache (the text tower must not run)
    calls = []
    orig = gpu.vlm.clip_model.text_features
    gpu.vlm.clip_model.text_features = lambda t: (calls.append(1), orig(t))[1]
    with torch.no_grad():
        out2 = gpu(xg)
    # (two GPU evaluations are not bit-identical: MIOpen / hipBLASLt pick their kernels per call)
    assert calls == [] and float((out2["featmap_q"] - out["featmap_q"]).abs().max()) <= 1e-4 * float(out["featmap_q"].abs().max())
    other = xs["prompt_tokens"].clone()
    other[:, 0, 1] += 1
    with torch.no_grad():
        gpu({"anchor": xg["anchor"], "query": xg["query"], "prompt_tokens": other})
    assert len(calls) == 1                                        # one new prompt set, shared by the B samples of the batch
    # ... and an in-place weight update invalidates it
    with torch.no_grad():
        gpu.vlm.clip_model.text_projection.mul_(1.5)
        gpu(xg)
    assert len(calls) == 2


@pytest.mark.gpu
def test_oryon_forward_fast_path_matches_the_fp32_modules():
    """The whole fast inference path of Oryon.forward (backbone.enable_fp16x3: fp16x3 linears / attentions in both towers and the fusion
    module, whole-map HIP convolutions and fused window attention in fusion, the HIP decoder) against the torch / MIOpen fp32 evaluation
    of the same module on the same device: descriptor maps and mask logits <= 1e-4 relative, the north-star descriptor bar."""
    from oryon_amd.backbone import enable_fp16x3
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    cfg = CLIPConfig.vit_l14_336()
    cfg.v_layers, cfg.t_layers = 2, 2
    torch.manual_seed(1)
    net = Oryon(default_model_args(), "cuda", clip_cfg=cfg).eval()
    B = 2
    gen = torch.Generator().manual_seed(4)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 11] = 49407
    toks[..., 12:] = 0
    xs = {"anchor": {"rgb": torch.rand(B, 3, 224, 224, generator=gen).cuda()}, "query": {"rgb": torch.rand(B, 3, 224, 224, generator=gen).cuda()},
          "prompt_tokens": toks.expand(B, 80, 77).contiguous()}
    with torch.no_grad():
        ref = net(xs)
        enable_fp16x3(True)
        try:
            out = net(xs)
            from oryon_amd.backbone import fusion as _F; assert _F._fast_cache(net.decoder).get("hip")                 # the HIP decoder ran
        finally:
            enable_fp16x3(False)
    for k in ("featmap_a", "featmap_q", "mask_a", "mask_q"):
        err = float((out[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert err < 1e-4, (k, err)


@pytest.mark.gpu
def test_fp16x3_linear_acc_equals_the_separate_residual_add():
    """oryon_linear_f16x3_acc (C += A W^T + bias, one fire-and-forget fp32 atomic add per element) gives exactly `x + linear(h)`: general and
    fp16-valued weights, ragged M, the half-wide last column tile; and the CLIP image tower whose blocks update the residual stream in
    place (backbone.clip.ACC_RESIDUAL) returns the same bits as the tower with separate adds, leaving its input untouched."""
    from oryon_amd import ops
    from oryon_amd.backbone import clip as clip_mod
    from oryon_amd.backbone.clip import CLIP, CLIPConfig
    dev = "cuda"
    torch.set_grad_enabled(False)
    g = torch.Generator(device=dev).manual_seed(11)
    for M, K, N, exact in ((577 * 3, 1024, 1024, False), (1000, 4096, 1024, True), (130, 64, 384, False), (1, 128, 128, True)):
        h = torch.randn(M, K, generator=g, device=dev) * 2.0
        w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
        if exact:
            w = w.half().float()
        b = torch.randn(N, generator=g, device=dev)
        x = torch.randn(M, N, generator=g, device=dev) * 5.0
        assert ops.linear_f16x3_acc_supported(w, x)
        ref = x + ops.linear_f16x3(h, w, b)
        buf = x.clone()
        assert ops.linear_f16x3_acc(h, w, b, buf) is buf and torch.equal(buf, ref), (M, K, N, float((buf - ref).abs().max()))
    assert not ops.linear_f16x3_acc_supported(torch.zeros(256, 32, device=dev), torch.zeros(4, 256, device=dev))  # K < 64
    cfg = CLIPConfig.vit_l14_336()
    cfg.v_layers, cfg.t_layers = 3, 1
    torch.manual_seed(0)
    m = CLIP(cfg).to(dev).eval()
    img = torch.randn(2, 3, 336, 336, device=dev)
    keep = img.clone()
    calls = []
    real = ops.linear_f16x3_acc
    ops.linear_f16x3_acc = lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1]
    clip_mod.FP16X3_LINEAR = True
    try:
        clip_mod.ACC_RESIDUAL = False
        sep = m.patch_tokens(img)
        assert not calls
        clip_mod.ACC_RESIDUAL = True
        acc = m.patch_tokens(img)
    finally:
        clip_mod.FP16X3_LINEAR, clip_mod.ACC_RESIDUAL = False, True
        ops.linear_f16x3_acc = real
    assert len(calls) == 2 * cfg.v_layers          # the in-place path was the one that ran
    assert torch.equal(sep, acc) and torch.equal(img, keep)


@pytest.mark.gpu
def test_fp16x3_linear_with_fp16_exact_weights_is_bit_identical():
    """Weights that are fp16 values in fp32 storage (what `clip.load(...)` + `.to(torch.float32)` leaves in the reference's CLIPEncoder,
    models/vlm.py:19-22) have an all-zero low half: ops.linear_f16x3 detects that once per weight and passes W_lo = NULL - the kernel
    variant without the a_hi * w_lo products.  Its results equal the three-product kernel's bit for bit, for every activation, ragged
    M and the half-wide last column tile; a weight with ONE element off the fp16 grid takes the three-product kernel."""
    from oryon_amd import ops
    dev = "cuda"
    torch.set_grad_enabled(False)
    g = torch.Generator(device=dev).manual_seed(5)
    for M, K, N, act in ((577 * 3, 1024, 3072, None), (1000, 1024, 4096, "quick"), (130, 4096, 1024, None), (1, 64, 256, None),
                         (1500, 128, 384, "erf"), (700, 512, 128, None)):
        x = torch.randn(M, K, generator=g, device=dev) * 3.0
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).half().float()
        b = torch.randn(N, generator=g, device=dev)
        kw = dict(quick_gelu=act == "quick", gelu=act == "erf")
        hi, lo = ops._split_weight_f16x3(w)
        assert lo is None and torch.equal(hi.float(), w)
        got = ops.linear_f16x3(x, w, b, **kw)
        ops.X3_EXACT_WEIGHTS = False
        try:
            w3 = w.clone()
            assert ops._split_weight_f16x3(w3)[1] is not None
            ref = ops.linear_f16x3(x, w3, b, **kw)
        finally:
            ops.X3_EXACT_WEIGHTS = True
        assert torch.equal(got, ref), (M, K, N, act, float((got - ref).abs().max()))
        r64 = torch.nn.functional.linear(x.double(), w.double(), b.double())
        if act == "erf":
            r64 = torch.nn.functional.gelu(r64)
        elif act:
            r64 = r64 * torch.sigmoid(1.702 * r64)
        assert float((got.double() - r64).abs().max()) / float(r64.abs().max()) < 5e-6
    w = (torch.randn(256, 128, generator=g, device=dev)).half().float()
    w[17, 5] += 2.0 ** -14                        # off the fp16 grid
    assert ops._split_weight_f16x3(w)[1] is not None
    w32 = torch.randn(256, 32, generator=g, device=dev).half().float()          # K = 32: the small-tile kernel keeps its (zero) low half
    assert ops._split_weight_f16x3(w32)[1] is not None


@pytest.mark.gpu
def test_fp16x3_linear_and_clip_tower_match_fp32():
    """B4 (oryon_linear_f16x3): the error-compensated fp16x3 linear against an fp64 reference (must be at least as accurate as torch's
    fp32 linear), ragged M, fused QuickGELU; and the CLIP image tower evaluated with it against the fp32 torch evaluation: patch tokens
    within 1e-5 relative - an order of magnitude inside the 1e-4 descriptor bar that bf16 cannot meet."""
    from oryon_amd import ops
    from oryon_amd.backbone import clip as clip_mod
    from oryon_amd.backbone.clip import CLIP, CLIPConfig
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = "cuda"
    torch.set_grad_enabled(False)                # inference-only kernel: the dispatch refuses to run under autograd
    g = torch.Generator(device=dev).manual_seed(1)
    for M, K, N, gelu in ((577 * 3, 1024, 3072, False), (1000, 1024, 4096, True), (130, 4096, 1024, False), (1, 64, 256, False),
                          (1500, 128, 384, "erf"), (700, 512, 128, False), (300, 32, 256, "erf")):   # half-wide last tile; K = 32 kernel
        x = torch.randn(M, K, generator=g, device=dev) * 3.0
        w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
        b = torch.randn(N, generator=g, device=dev)
        assert ops.linear_f16x3_supported(x, w)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        f32 = torch.nn.functional.linear(x, w, b)
        if gelu == "erf":
            ref, f32 = torch.nn.functional.gelu(ref), torch.nn.functional.gelu(f32)
        elif gelu:
            ref, f32 = ref * torch.sigmoid(1.702 * ref), f32 * torch.sigmoid(1.702 * f32)
        got = ops.linear_f16x3(x, w, b, quick_gelu=gelu is True, gelu=gelu == "erf")
        scale = float(ref.abs().max())
        e_x3, e_32 = float((got.double() - ref).abs().max()) / scale, float((f32.double() - ref).abs().max()) / scale
        assert e_x3 < 5e-6 and e_x3 <= 2.0 * e_32 + 1e-7, (M, K, N, e_x3, e_32)
    assert not ops.linear_f16x3_supported(torch.zeros(4, 100, device=dev), torch.zeros(256, 100, device=dev))    # K % 32
    assert not ops.linear_f16x3_supported(torch.zeros(4, 32, device=dev), torch.zeros(128, 32, device=dev))      # N % 256 when K < 64
    cfg = CLIPConfig.vit_l14_336()
    cfg.v_layers, cfg.t_layers = 3, 1
    torch.manual_seed(0)
    m = CLIP(cfg).to(dev).eval()
    img = torch.randn(2, 3, 336, 336, device=dev)
    with torch.no_grad():
        ref = m.patch_tokens(img)
        clip_mod.FP16X3_LINEAR = True
        try:
            got = m.patch_tokens(img)
        finally:
            clip_mod.FP16X3_LINEAR = False
    torch.set_grad_enabled(True)
    assert float((got - ref).abs().max()) < 1e-5 * float(ref.abs().max())


@pytest.mark.gpu
def test_swin_fused_f32_attention_matches_plain_path():
    """oryon_swin_window_attention_f32 (the B3 kernel on fp32 tensors) inside SwinGuidance against the plain torch evaluation of the
    same module: the three guidance maps within 2e-5 relative at the reference's 384x384 input (96x96 tokens, padded windows, shifts)."""
    from oryon_amd.backbone import swin as swin_mod
    from oryon_amd.backbone.swin import SwinGuidance
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.manual_seed(0)
    m = SwinGuidance().eval()
    for p in m.parameters():
        p.data.normal_(0, 0.3 if p.dim() == 1 else 0.05)
    m = m.cuda()
    x = torch.randn(2, 3, 384, 384, device="cuda")
    with torch.no_grad():
        ref = m(x)
        swin_mod.FUSED_F32_ATTENTION = True
        try:
            got = m(x)
        finally:
            swin_mod.FUSED_F32_ATTENTION = False
    for k in ("guidance1", "guidance2", "guidance3"):
        assert float((got[k] - ref[k]).abs().max()) < 2e-5 * float(ref[k].abs().max()), k


@pytest.mark.gpu
def test_mha_f16x3_matches_fp64_attention():
    """B5 (oryon_mha_f16x3) on the CLIP image tower's shapes (L = 577: ragged last key tile and query block, 16 heads of 64) against an
    fp64 evaluation of softmax(Q K^T / 8) V; must be at least as accurate as torch's fp32 scaled_dot_product_attention."""
    from oryon_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(2)
    for N, L, H in ((3, 577, 16), (2, 64, 2), (1, 130, 1)):
        D = 64 * H
        qkv = torch.randn(N, L, 3 * D, generator=g, device=dev) * 1.5
        got = ops.mha_f16x3(qkv, H)
        q, k, v = qkv.view(N, L, 3, H, 64).permute(2, 0, 3, 1, 4)
        ref = torch.softmax(q.double() @ k.double().transpose(-2, -1) / 8.0, dim=-1) @ v.double()
        ref = ref.transpose(1, 2).reshape(N, L, D)
        f32 = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(N, L, D)
        scale = float(ref.abs().max())
        e_x3, e_32 = float((got.double() - ref).abs().max()) / scale, float((f32.double() - ref).abs().max()) / scale
        assert e_x3 < 5e-6 and e_x3 <= 3.0 * e_32 + 2e-7, (N, L, H, e_x3, e_32)


@pytest.mark.gpu
def test_add_layernorm_f32_matches_fp64():
    """fp32 twin of B2 (oryon_add_layernorm_f32): (x + delta, LayerNorm(x + delta)) against an fp64 evaluation on every row width the
    towers use (Swin 128..2048, CLIP 768 / 1024) plus a ragged row count; must be at least as accurate as torch's fp32 LayerNorm."""
    from oryon_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    for rows, D in ((1000, 128), (999, 256), (577 * 2, 1024), (77 * 3, 768), (50, 2048), (7, 64), (3, 40)):
        x = torch.randn(rows, D, generator=g, device=dev) * 2.0 + 0.5
        d = torch.randn(rows, D, generator=g, device=dev)
        w = torch.randn(D, generator=g, device=dev)
        b = torch.randn(D, generator=g, device=dev)
        for delta in (None, d):
            s_ref = x if delta is None else x + delta
            h_ref = torch.nn.functional.layer_norm(s_ref.double(), (D,), w.double(), b.double(), 1e-5)
            h_32 = torch.nn.functional.layer_norm(s_ref, (D,), w, b, 1e-5)
            s, h = ops.add_layernorm_f32(x, None if delta is None else delta.clone(), w, b, 1e-5)
            assert torch.equal(s, s_ref)
            e, e32 = float((h.double() - h_ref).abs().max()), float((h_32.double() - h_ref).abs().max())
            assert e < 5e-6 and e <= 2.0 * e32 + 1e-7, (rows, D, e, e32)


def test_fp16x3_switches_and_the_fallback_context():
    """enable_fp16x3 switches every fast-path flag together - since round 5 the fused kernels stay ON in guard mode too (they carry the
    device-side range flag) - and fp16x3_disabled() restores exactly what was set."""
    from oryon_amd import ops
    from oryon_amd.backbone import enable_fp16x3, fp16x3_disabled, fp16x3_enabled, fusion
    try:
        enable_fp16x3(True, guard=True)
        assert fusion.FP16X3_LINEAR and fusion.FUSED_KERNELS and fusion.HIP_DECODER and ops.X3_GUARD and fp16x3_enabled()
        with fp16x3_disabled():
            assert not fp16x3_enabled() and not ops.X3_GUARD
        assert fusion.FP16X3_LINEAR and fusion.FUSED_KERNELS and fusion.HIP_DECODER and ops.X3_GUARD
        enable_fp16x3(True)
        assert fusion.FP16X3_LINEAR and fusion.FUSED_KERNELS and fusion.HIP_DECODER and not ops.X3_GUARD
    finally:
        enable_fp16x3(False)
    assert not (fusion.FP16X3_LINEAR or fusion.FUSED_KERNELS or fusion.HIP_DECODER or fp16x3_enabled())


@pytest.mark.gpu
def test_fp16x3_device_range_flag():
    """oryon_x3_range_flag (round 5): the fp16x3 linear / convolution kernels raise a per-device flag when a pre-activation output is not a
    finite value below 60000 - in-range calls leave it clear; an activation beyond float16's range (hi = inf), a NaN, and an output that
    merely grows past the limit each set it; reading with reset clears it."""
    from oryon_amd import ops
    torch.manual_seed(0)
    w = torch.randn(256, 128, device="cuda") * 0.05
    b = torch.randn(256, device="cuda")
    x = torch.randn(300, 128, device="cuda")
    with torch.no_grad():
        ops.x3_range_flag(x.device, reset=True)
        ops.linear_f16x3(x, w, b)
        ops.linear_f16x3(x, w, b, gelu=True)
        assert ops.x3_range_flag(x.device) is False
        xb = x.clone(); xb[299, 5] = 7.0e4                          # splits into hi = inf: every output of that row is inf / NaN
        y = ops.linear_f16x3(xb, w, b)
        assert not torch.isfinite(y[299]).all() and torch.isfinite(y[:299]).all()
        assert ops.x3_range_flag(x.device) is True and ops.x3_range_flag(x.device) is False        # read + reset
        xn = x.clone(); xn[0, 0] = float("nan")
        ops.linear_f16x3(xn, w, b, quick_gelu=True)
        assert ops.x3_range_flag(x.device) is True
        ops.linear_f16x3(x * 3.0e3, w * 40.0, b)                    # operands in range, outputs ~1e5: the NEXT split could not hold them
        assert ops.x3_range_flag(x.device) is True
        # the 24 x 24 convolution and the decoder raise it the same way
        xc = torch.randn(1, 24, 24, 64, device="cuda")
        wc = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
        ops.conv24_f16x3(xc, wc, None)
        assert ops.x3_range_flag(x.device) is False
        xc[0, 3, 3, 1] = 1.0e5
        ops.conv24_f16x3(xc, wc, None)
        assert ops.x3_range_flag(x.device) is True


@pytest.mark.gpu
def test_fp16x3_range_flag_is_per_stream():
    """the flag word belongs to (device, stream).  An overflow on stream A is seen by a reader of A only; a reader /
    reset on stream B neither sees nor clears it; `Oryon.forward`-style use (clear, launch, read) on B is undisturbed by A."""
    from oryon_amd import ops
    torch.manual_seed(0)
    w = torch.randn(256, 128, device="cuda") * 0.05
    b = torch.randn(256, device="cuda")
    x = torch.randn(300, 128, device="cuda")
    xb = x.clone(); xb[7, 5] = 7.0e4
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.no_grad():
        with torch.cuda.stream(sa):
            ops.x3_range_reset()
            ops.linear_f16x3(xb, w, b)                          # raises A's word
        with torch.cuda.stream(sb):
            ops.x3_range_reset()                                # B's forward begins: must not clear A's word
            ops.linear_f16x3(x, w, b)
            assert ops.x3_range_flag(x.device) is False         # B never overflowed, whatever A did
        assert ops.x3_range_flag(x.device) is False             # nor did the default stream
        with torch.cuda.stream(sa):
            assert ops.x3_range_flag(x.device) is True          # A's flag survived B's reset and B's read
            assert ops.x3_range_flag(x.device) is False         # ... and was cleared by its own read
        # a flag left behind on a stream (a direct call nobody read) is cleared by the reset that opens the next forward on that stream
        ops.linear_f16x3(xb, w, b)
        ops.x3_range_reset()
        ops.linear_f16x3(x, w, b)
        assert ops.x3_range_flag(x.device) is False
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_fp16x3_overflow_behind_an_unchecked_kernel_is_flagged_by_the_next():
    """The attention kernels carry no range check of their own (include/oryon_hip.h, oryon_x3_range_flag): the argument is that whatever an
    out-of-range operand makes of their output reaches a CHECKED kernel next.  Executed: an out-of-range v inside the CLIP attention
    (mha_f16x3) and inside fusion's window attention leaves the flag clear at that kernel and raises it in the projection that consumes
    the output; a residual value that the in-place linear's atomics push past the range is caught by the LayerNorm pass that reads it."""
    from oryon_amd import ops
    torch.manual_seed(1)
    dev = torch.device("cuda")
    with torch.no_grad():
        # (1) CLIP attention -> out projection
        qkv = torch.randn(2, 77, 3 * 128, device=dev)
        qkv[1, 5, 2 * 128 + 3] = 7.0e4                          # a v entry beyond float16: hi = inf
        ops.x3_range_reset()
        att = ops.mha_f16x3(qkv, heads=2)
        assert ops.x3_range_flag(dev) is False                  # the attention itself is not instrumented ...
        assert not torch.isfinite(att[1]).all() and torch.isfinite(att[0]).all()
        wo = torch.randn(128, 128, device=dev) * 0.05
        ops.linear_f16x3(att, wo, None)
        assert ops.x3_range_flag(dev) is True                   # ... the checked linear that consumes its output is
        # (2) fusion's window attention -> projection
        qk = torch.randn(1, 24, 24, 256, device=dev)
        v = torch.randn(1, 24, 24, 128, device=dev)
        v[0, 4, 4, 9] = 1.0e5
        ops.x3_range_reset()
        o = ops.fusion_window_attention(qk, v, heads=4, window=12, shift=0)
        assert ops.x3_range_flag(dev) is False and not torch.isfinite(o).all()
        ops.linear_f16x3(o, wo, None)
        assert ops.x3_range_flag(dev) is True
        # (3) the residual stream: C and the update are each in range, their sum is not - the in-place linear checks its own finished sum
        #     only (fire-and-forget atomics), the residual-add + LayerNorm pass that reads C next checks the stream itself
        x = torch.randn(256, 128, device=dev)
        w = torch.randn(128, 128, device=dev) * 0.05
        bias = torch.full((128,), 8.0e3, device=dev)
        C = torch.full((256, 128), 5.9e4, device=dev)
        ops.x3_range_reset()
        if ops.linear_f16x3_acc_supported(w, C):
            ops.linear_f16x3_acc(x, w, bias, C)
            assert ops.x3_range_flag(dev) is False and float(C.min()) > 6.55e4
        else:
            C += 8.0e3
        g, be = torch.ones(128, device=dev), torch.zeros(128, device=dev)
        ops.add_layernorm_f32(C, None, g, be, 1e-5)
        assert ops.x3_range_flag(dev) is True
        ops.add_layernorm_f32(torch.randn(256, 128, device=dev), torch.randn(256, 128, device=dev), g, be, 1e-5)
        assert ops.x3_range_flag(dev) is False


@pytest.mark.gpu
def test_oryon_forward_falls_back_to_fp32_when_the_range_flag_is_raised():
    """Oryon.forward on the fast path reads the flag once per forward; with a decoder weight blown up so that an up-convolution output
    leaves float16's range the forward is evaluated again with the torch fp32 modules: same result as enable_fp16x3(False), counted."""
    from oryon_amd.backbone import enable_fp16x3
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    torch.manual_seed(3)
    net = Oryon(default_model_args(), "cuda", clip_cfg=CLIPConfig(v_layers=1, t_layers=1)).eval()
    gen = torch.Generator().manual_seed(0)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 10] = 49407
    toks[..., 11:] = 0
    xs = {"anchor": {"rgb": torch.rand(1, 3, 224, 224, generator=gen).cuda()}, "query": {"rgb": torch.rand(1, 3, 224, 224, generator=gen).cuda()},
          "prompt_tokens": toks.cuda()}
    with torch.no_grad():
        enable_fp16x3(True)
        try:
            n0 = Oryon.x3_range_fallbacks
            fast = net(xs)
            assert Oryon.x3_range_fallbacks == n0                  # random-init network: everything in range, no fallback
            net.decoder.decoder1.up.bias += 9.0e4                   # the first up-convolution now outputs ~9e4: beyond the next split
            out = net(xs)
            assert Oryon.x3_range_fallbacks == n0 + 1
        finally:
            enable_fp16x3(False)
        ref = net(xs)
    assert all(torch.isfinite(v).all() for v in out.values())
    for k in ref:                                                  # two fp32 library evaluations (MIOpen picks its algorithms per call): equal to round-off
        torch.testing.assert_close(out[k], ref[k], rtol=1e-5, atol=1e-5 * float(ref[k].abs().max()))
    assert torch.isfinite(fast["featmap_a"]).all()


@pytest.mark.gpu
def test_fp16x3_guard_catches_operands_outside_the_float16_range():
    """enable_fp16x3(True, guard=True): an activation beyond float16's range would split into inf silently; the guard evaluates that
    layer with torch's fp32 linear instead (counted), in-range layers still take the kernel; an out-of-range weight is refused."""
    from oryon_amd import ops
    from oryon_amd.backbone import enable_fp16x3
    torch.manual_seed(0)
    w = torch.randn(256, 128, device="cuda") * 0.05
    b = torch.randn(256, device="cuda")
    x = torch.randn(4, 64, 128, device="cuda")
    with torch.no_grad():
        ref = torch.nn.functional.linear(x, w, b)
        enable_fp16x3(True, guard=True)
        try:
            before = ops.x3_guard_fallbacks
            y = ops.linear_f16x3(x, w, b)
            assert ops.x3_guard_fallbacks == before and float((y - ref).abs().max()) < 1e-5
            x_big = x.clone()
            x_big[0, 0, 0] = 7.0e4
            unguarded_ref = torch.nn.functional.linear(x_big, w, b)
            y2 = ops.linear_f16x3(x_big, w, b)
            assert ops.x3_guard_fallbacks == before + 1 and torch.isfinite(y2).all() and torch.equal(y2, unguarded_ref)
            with pytest.raises(Exception, match="does not fit the float16 split"):
                ops.linear_f16x3(x, w * 1.0e7, b)
        finally:
            enable_fp16x3(False)
    assert ops.X3_GUARD is False
