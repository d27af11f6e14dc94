"""CPU tests of the PyTorch backbone restatement (rows a1-a5): fusion + decoder against outputs of the REAL reference
(tests/golden/g5_backbone.npz), state-dict key compatibility, CLIP towers cross-checked against the independent
`transformers` implementation (structure check; parity with the third-party `clip` package is unpinned), Swin shapes,
the full Oryon forward signature on a tiny CLIP configuration, and the prompt-embedding cache."""
import os

import numpy as np
import pytest
import torch

from oracle import oryon_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def _inputs(B=2):
    img = orc.hashed_tensor((B, 1024, 24, 24), 100, 0, 1.0)
    text = orc.hashed_tensor((B, 1, 80, 768), 101, 0, 1.0)
    guid = [orc.hashed_tensor((B, 512, 24, 24), 102, 0, 1.0), orc.hashed_tensor((B, 256, 48, 48), 103, 0, 1.0),
            orc.hashed_tensor((B, 128, 96, 96), 104, 0, 1.0)]
    return img, text, guid


def test_fusion_and_decoder_match_reference_golden():
    from oryon_amd.backbone.fusion import ImageTextFusion, StandardDecoder
    g = np.load(os.path.join(GOLD, "g5_backbone.npz"))
    fusion = ImageTextFusion("cpu").eval()
    decoder = StandardDecoder("cpu", True, True, input_dim=128, decoder_dims=[64, 32]).eval()
    # identical key sets and order as the reference modules (so its checkpoints load unchanged)
    assert list(fusion.state_dict().keys()) == list(g["fusion_keys"])
    assert list(decoder.state_dict().keys()) == list(g["decoder_keys"])
    fusion.load_state_dict(orc.analytic_state_dict(fusion.state_dict(), seed=3), strict=True)
    decoder.load_state_dict(orc.analytic_state_dict(decoder.state_dict(), seed=4), strict=True)
    img, text, guid = _inputs()
    with torch.no_grad():
        feats = fusion(img, text, guid)
        mask, featmap = decoder(feats, guid)
    assert tuple(feats.shape) == (2, 128, 1, 24, 24) and tuple(mask.shape) == (2, 1, 192, 192) and tuple(featmap.shape) == (2, 32, 192, 192)
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
    assert rel(feats.numpy(), g["fusion_out"]) < 1e-4            # north_star: <= 1e-4 rel on float descriptors
    assert rel(mask[:, :, ::2, ::2].numpy(), g["mask"]) < 1e-4
    assert rel(featmap[:, :, ::4, ::4].numpy(), g["featmap_sub"]) < 1e-4
    assert rel(featmap.double().sum(dim=(2, 3)).numpy(), g["featmap_sum"]) < 1e-4
    assert rel(featmap.double().abs().sum(dim=(2, 3)).numpy(), g["featmap_abs_sum"]) < 1e-5


def test_clip_towers_against_transformers():
    """Tiny CLIP: copy our parameters into transformers' CLIPModel (independent implementation of the published
    architecture) and compare patch tokens / text features."""
    tr = pytest.importorskip("transformers")
    from oryon_amd.backbone.clip import CLIP, CLIPConfig
    cfg = CLIPConfig(image_size=28, patch=14, v_width=64, v_layers=2, v_heads=4, embed_dim=32, ctx=16, vocab=100, t_width=48,
                     t_layers=2, t_heads=4)
    torch.manual_seed(0)
    m = CLIP(cfg).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.2)
    hf_cfg = tr.CLIPConfig(
        text_config=dict(vocab_size=100, hidden_size=48, intermediate_size=192, num_hidden_layers=2, num_attention_heads=4,
                         max_position_embeddings=16, hidden_act="quick_gelu", eos_token_id=99, bos_token_id=98, pad_token_id=0),
        vision_config=dict(hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, image_size=28,
                           patch_size=14, hidden_act="quick_gelu"),
        projection_dim=32)
    hf = tr.CLIPModel(hf_cfg).eval()
    sd = m.state_dict()

    def copy_block(prefix_hf, prefix, width):
        layer = dict(hf.named_parameters())
        w, b = sd[prefix + ".attn.in_proj_weight"], sd[prefix + ".attn.in_proj_bias"]
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            layer[f"{prefix_hf}.self_attn.{n}.weight"].data.copy_(w[i * width:(i + 1) * width])
            layer[f"{prefix_hf}.self_attn.{n}.bias"].data.copy_(b[i * width:(i + 1) * width])
        for a, c in ((".self_attn.out_proj", ".attn.out_proj"), (".layer_norm1", ".ln_1"), (".layer_norm2", ".ln_2"),
                     (".mlp.fc1", ".mlp.c_fc"), (".mlp.fc2", ".mlp.c_proj")):
            layer[prefix_hf + a + ".weight"].data.copy_(sd[prefix + c + ".weight"])
            layer[prefix_hf + a + ".bias"].data.copy_(sd[prefix + c + ".bias"])

    P = dict(hf.named_parameters())
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
            copy_block(f"vision_model.encoder.layers.{i}", f"visual.transformer.resblocks.{i}", 64)
            copy_block(f"text_model.encoder.layers.{i}", f"transformer.resblocks.{i}", 48)
    img = torch.randn(3, 3, 28, 28)
    toks = torch.randint(1, 98, (5, 16))
    toks[:, 0] = 98
    for r, e in enumerate((5, 9, 15, 3, 12)):
        toks[r, e] = 99                                     # EOT = highest id -> argmax position (vlm.py:81)
        toks[r, e + 1:] = 0
    with torch.no_grad():
        ours_v = m.patch_tokens(img)                                            # ln_post on patch tokens, no projection
        hv = hf.vision_model(pixel_values=img).last_hidden_state               # before post_layernorm
        ref_v = hf.vision_model.post_layernorm(hv[:, 1:, :]).transpose(1, 2).reshape(3, 64, 2, 2)
        ours_t = m.text_features(toks)
        ref_t = hf.get_text_features(input_ids=toks, attention_mask=torch.ones_like(toks))
        if not torch.is_tensor(ref_t):
            ref_t = ref_t.pooler_output if hasattr(ref_t, "pooler_output") else ref_t[0]
    assert float((ours_v - ref_v).abs().max()) < 2e-4 * float(ref_v.abs().max())
    assert float((ours_t - ref_t).abs().max()) < 2e-4 * float(ref_t.abs().max())


def test_clip_state_dict_names_and_preprocess():
    from oryon_amd.backbone.clip import CLIP, CLIPConfig, clip_preprocess
    m = CLIP(CLIPConfig(image_size=28, patch=14, v_width=32, v_layers=1, v_heads=2, embed_dim=16, ctx=8, vocab=50, t_width=24,
                        t_layers=1, t_heads=2))
    keys = set(m.state_dict().keys())
    for k in ("visual.conv1.weight", "visual.class_embedding", "visual.positional_embedding", "visual.ln_pre.weight",
              "visual.ln_post.bias", "visual.proj", "visual.transformer.resblocks.0.attn.in_proj_weight",
              "visual.transformer.resblocks.0.attn.out_proj.bias", "visual.transformer.resblocks.0.mlp.c_fc.weight",
              "visual.transformer.resblocks.0.mlp.c_proj.bias", "visual.transformer.resblocks.0.ln_1.weight",
              "token_embedding.weight", "positional_embedding", "transformer.resblocks.0.ln_2.bias", "ln_final.weight",
              "text_projection", "logit_scale"):
        assert k in keys, k
    x = clip_preprocess(torch.rand(2, 3, 224, 224), 336)
    assert tuple(x.shape) == (2, 3, 336, 336)
    from oryon_amd.backbone.clip import CLIPConfig as C
    big = C.vit_l14_336()
    assert (big.v_width, big.v_layers, big.v_heads, big.t_width, big.t_layers, big.ctx, big.vocab) == (1024, 24, 16, 768, 12, 77, 49408)


def test_swin_guidance_shapes_and_names():
    from oryon_amd.backbone.swin import SwinGuidance, guidance_embeds
    sw = SwinGuidance().eval()
    keys = set(sw.state_dict().keys())
    for k in ("features.0.0.weight", "features.0.2.bias", "features.1.0.norm1.weight", "features.1.1.attn.qkv.weight",
              "features.1.1.attn.proj.bias", "features.1.0.attn.relative_position_bias_table",
              "features.1.0.attn.relative_position_index", "features.1.1.mlp.0.weight", "features.1.1.mlp.3.bias",
              "features.2.reduction.weight", "features.2.norm.bias", "features.3.1.attn.qkv.bias", "features.4.reduction.weight"):
        assert k in keys, k
    assert "features.2.reduction.bias" not in keys
    assert sw.state_dict()["features.1.0.attn.relative_position_bias_table"].shape == (169, 4)
    assert sw.state_dict()["features.3.0.attn.relative_position_bias_table"].shape == (169, 8)
    with torch.no_grad():
        g1, g2, g3 = guidance_embeds(sw, torch.rand(1, 3, 224, 224))
    assert tuple(g1.shape) == (1, 512, 24, 24) and tuple(g2.shape) == (1, 256, 48, 48) and tuple(g3.shape) == (1, 128, 96, 96)
    # shifted-window attention is a permutation-consistent operator: zero input + zero biases -> finite output
    assert torch.isfinite(g1).all()


def test_oryon_forward_contract_tiny_clip():
    """configs[0]-style plumbing: 2 synthetic 224x224 pairs through Oryon.forward on CPU with a shallow CLIP (full widths,
    1 layer per tower, so the 24x24x1024 / 768 interfaces are the real ones)."""
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    cfg = CLIPConfig(v_layers=1, t_layers=1)
    torch.manual_seed(0)
    net = Oryon(default_model_args(), "cpu", clip_cfg=cfg).eval()
    sd = net.state_dict()
    assert any(k.startswith("vlm.clip_model.visual.") for k in sd) and any(k.startswith("guidance_backbone.features.") for k in sd)
    assert any(k.startswith("fusion.layers.0.swin_block.block_1.attn.q.") for k in sd) and "decoder.head.weight" in sd
    assert len(net.get_trainable_parameters()) == len(list(net.fusion.parameters())) + len(list(net.decoder.parameters()))
    B = 2
    gen = torch.Generator().manual_seed(0)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 10] = 49407
    toks[..., 11:] = 0
    xs = {"anchor": {"rgb": torch.rand(B, 3, 224, 224, generator=gen)}, "query": {"rgb": torch.rand(B, 3, 224, 224, generator=gen)},
          "prompt_tokens": toks.expand(B, 80, 77).contiguous()}
    with torch.no_grad():
        out = net(xs)
        out2 = net(xs)
    assert set(out) == {"featmap_a", "featmap_q", "mask_a", "mask_q"}
    assert tuple(out["featmap_a"].shape) == (B, 32, 192, 192) and tuple(out["mask_q"].shape) == (B, 1, 192, 192)
    assert all(torch.isfinite(v).all() for v in out.values())
    assert len(net.vlm._prompt_cache) == 1                       # identical prompt sets are encoded once
    assert torch.equal(out["featmap_a"], out2["featmap_a"])
    # the joint anchor+query batch of the inference path must equal the two separate passes the reference makes (net.py:145-160)
    with torch.no_grad():
        solo = net({"anchor": {"rgb": xs["query"]["rgb"][:1]}, "query": {"rgb": xs["anchor"]["rgb"]}, "prompt_tokens": xs["prompt_tokens"][:1]})
    torch.testing.assert_close(solo["featmap_a"][0], out["featmap_q"][0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(solo["mask_q"][1], out["mask_a"][1], rtol=1e-4, atol=1e-5)
    net.train()
    assert net.fusion.training and not net.vlm.training          # CLIP never leaves eval mode (net.py:78-89, vlm.py:30-34)
    with pytest.raises(RuntimeError):
        net.vlm.encode_prompt([["mug"] + ["a photo of a mug"] * 80])   # string prompts need the BPE vocabulary file


def test_tokenizer_matches_reference_on_fabricated_merge_table():
    """Token ids of the reference's SimpleTokenizer (models/tokenizer.py:64-151, ftfy stubbed with the identity) for the prompts of
    tests/golden/g10_tokenizer.npz, on the fabricated merge table shipped beside it."""
    from oryon_amd.backbone.tokenizer import SimpleTokenizer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_tokenizer.npz"))
    tok = SimpleTokenizer(os.path.join(os.path.dirname(__file__), "golden", "bpe_fabricated.txt.gz"))
    ids = tok(g["texts"].tolist())
    assert tuple(ids.shape) == tuple(g["ids"].shape) == (len(g["texts"]), 77)
    assert np.array_equal(ids.numpy(), g["ids"])
    assert int((g["ids"][:, -1] != 0).sum()) == 1                      # the over-long prompt fills the context (no end token restored)
    assert tok("a photo of a mug.").shape == (1, 77)
