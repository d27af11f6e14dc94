"""a5 on the device: StandardDecoder.forward through oryon_decoder_forward (csrc/decoder.hip) against (1) the torch fp32 module layer
by layer on random inputs and (2) the outputs of the imported reference (tests/golden/g5_backbone.npz, models/decoder.py:82-108)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _decoder(seed):
    from oracle import oryon_oracle as orc
    from oryon_amd.backbone.fusion import StandardDecoder
    dec = StandardDecoder("cpu", True, True, input_dim=128, decoder_dims=[64, 32]).eval()
    dec.load_state_dict(orc.analytic_state_dict(dec.state_dict(), seed=seed), strict=True)
    return dec.to("cuda")


def _nhwc(buf, off, n, H, W, C):
    return buf[off:off + n * H * W * C * 4].view(torch.float32).view(n, H, W, C).permute(0, 3, 1, 2)


@pytest.mark.parametrize("n,h,w", [(2, 24, 24), (3, 8, 16)])
def test_hip_decoder_layers_match_torch_fp32(n, h, w):
    """Every intermediate the workspace exposes (cat buffers, raw convolution outputs) and both outputs against the torch modules run
    in fp32 on the same device: <= 2e-5 of the tensor's largest magnitude (the fp16x3 products are ~2^-22 relative)."""
    from oryon_amd.backbone import fusion
    from oryon_amd.backbone.decoder_hip import HipDecoder
    torch.backends.cudnn.allow_tf32 = False
    torch.manual_seed(5)
    dec = _decoder(7)
    x = torch.randn(n, 128, h, w, device="cuda")
    g2 = torch.randn(n, 256, 2 * h, 2 * w, device="cuda") * 2.0
    g3 = torch.randn(n, 128, 4 * h, 4 * w, device="cuda") * 0.5 + 0.3
    hip = HipDecoder(dec, x.device)
    off = hip.layout(n, h, w)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    with torch.no_grad():
        pg = [proj(g) for proj, g in zip(dec.decoder_guidance_projection, (g2, g3))] + [None]
        y = x
        H, W = h, w
        for i, blk in enumerate((dec.decoder1, dec.decoder2, dec.decoder3)):
            H, W = 2 * H, 2 * W
            cat = blk.up(y)
            if pg[i] is not None:
                cat = torch.cat([cat, pg[i]], dim=1)
            c1 = blk.conv.double_conv[0](cat)
            c2 = blk.conv.double_conv[3](blk.conv.double_conv[2](blk.conv.double_conv[1](c1)))
            y = blk.conv.double_conv[5](blk.conv.double_conv[4](c2))
            for j, ref in ((1, cat), (2, c1), (3, c2)):
                hip.forward(x, g2, g3, stop_after=3 * i + j)
                torch.cuda.synchronize()
                got = _nhwc(hip.workspace(n, h, w), off[j - 1], n, H, W, ref.shape[1])
                assert rel(got, ref) < 2e-5, (i, j, rel(got, ref))
        logits_ref = dec.head(y)
        lg, fm = hip.forward(x, g2, g3)
        assert rel(fm, y) < 2e-5 and rel(lg, logits_ref[:, 0]) < 2e-5, (rel(fm, y), rel(lg, logits_ref[:, 0]))
        # ... and element by element (the max-norm above says nothing about small entries): 1e-4 relative + 1e-5 of the tensor's scale
        torch.testing.assert_close(fm, y, rtol=1e-4, atol=1e-5 * float(y.abs().max()))
        torch.testing.assert_close(lg, logits_ref[:, 0], rtol=1e-4, atol=1e-5 * float(logits_ref.abs().max()))
        # the module switch routes StandardDecoder.forward through the same handle (at the module's own output size; for other input sizes
        # the reference resizes to (192, 192), models/decoder.py:101-104, and the torch path keeps doing that)
        fusion.enable_hip_decoder(True)
        try:
            lg2, fm2 = dec(x.view(n, 128, 1, h, w), [None, g2, g3])
        finally:
            fusion.enable_hip_decoder(False)
        if (h, w) == (24, 24):
            assert lg2.shape == (n, 1, 8 * h, 8 * w) and torch.equal(fm2, fm) and torch.equal(lg2[:, 0], lg)
        else:
            assert lg2.shape == (n, 1, 192, 192) and not fusion._fast_cache(dec).get("hip")
        lg3, fm3 = hip.forward(x, g2, g3)
        assert torch.equal(fm3, fm) and torch.equal(lg3, lg)                    # fixed reduction order: bit-reproducible
        # guidance maps as the Swin tower hands them out (permuted views of NHWC storage) are read in place: same values
        g2v, g3v = (g.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for g in (g2, g3))
        lg4, fm4 = hip.forward(x, g2v, g3v)
        assert torch.equal(fm4, fm) and torch.equal(lg4, lg)


def test_hip_decoder_matches_reference_golden():
    """The G5 fixture (fusion + decoder of the imported reference on hashed inputs / analytic weights): the decoder alone on the HIP path,
    fed with the torch fusion output of the same test - <= 1e-4 relative, the north-star descriptor bar."""
    from oracle import oryon_oracle as orc
    from oryon_amd.backbone import fusion as F_
    from oryon_amd.backbone.fusion import ImageTextFusion
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    g = np.load(os.path.join(GOLD, "g5_backbone.npz"))
    dev = "cuda"
    fusion = ImageTextFusion("cpu").eval()
    fusion.load_state_dict(orc.analytic_state_dict(fusion.state_dict(), seed=3), strict=True)
    fusion = fusion.to(dev)
    decoder = _decoder(4)
    B = 2
    img = orc.hashed_tensor((B, 1024, 24, 24), 100, 0, 1.0).to(dev)
    text = orc.hashed_tensor((B, 1, 80, 768), 101, 0, 1.0).to(dev)
    guid = [orc.hashed_tensor((B, 512, 24, 24), 102, 0, 1.0).to(dev), orc.hashed_tensor((B, 256, 48, 48), 103, 0, 1.0).to(dev),
            orc.hashed_tensor((B, 128, 96, 96), 104, 0, 1.0).to(dev)]
    from oryon_amd.backbone import enable_fp16x3
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
    F_.enable_hip_decoder(True)
    try:
        with torch.no_grad():
            feats = fusion(img, text, guid)
            mask, featmap = decoder(feats, guid)
        assert F_._fast_cache(decoder).get("hip")                         # the HIP path ran, not the torch modules
        # the whole fast inference path (what bench.py's fp16x3 stage sets run): fusion linears on the fp16x3 kernel as well
        enable_fp16x3(True)
        with torch.no_grad():
            feats_x3 = fusion(img, text, guid)
            mask_x3, featmap_x3 = decoder(feats_x3, guid)
        assert rel(feats_x3.cpu().numpy(), g["fusion_out"]) < 1e-4
        assert rel(mask_x3[:, :, ::2, ::2].cpu().numpy(), g["mask"]) < 1e-4
        assert rel(featmap_x3[:, :, ::4, ::4].cpu().numpy(), g["featmap_sub"]) < 1e-4
    finally:
        enable_fp16x3(False)
        F_.enable_hip_decoder(False)
    assert rel(mask[:, :, ::2, ::2].cpu().numpy(), g["mask"]) < 1e-4
    assert rel(featmap[:, :, ::4, ::4].cpu().numpy(), g["featmap_sub"]) < 1e-4
    assert rel(featmap.double().sum(dim=(2, 3)).cpu().numpy(), g["featmap_sum"]) < 1e-4
    assert rel(featmap.double().abs().sum(dim=(2, 3)).cpu().numpy(), g["featmap_abs_sum"]) < 1e-5


def test_hip_decoder_argument_checks():
    from oryon_amd._lib import lib
    assert lib().oryon_decoder_workspace_bytes(2, 24, 24) > 0
    assert lib().oryon_decoder_workspace_bytes(2, 20, 24) == 0 and lib().oryon_decoder_workspace_bytes(0, 24, 24) == 0


@pytest.mark.parametrize("shift", [0, 6])
def test_fusion_window_attention_matches_torch(shift):
    """ops.fusion_window_attention (roll + 12 x 12 windows + masked softmax attention + un-window in one kernel) against the module's
    torch path (models/fusion.py:75-103 inside :173-213) on random projections."""
    from oryon_amd import ops
    from oryon_amd.backbone.fusion import _GuidedSwinBlock, _to_windows, _from_windows
    torch.manual_seed(3 + shift)
    B, H, W, C, heads = 3, 24, 24, 128, 4
    blk = _GuidedSwinBlock(C, C, (H, W), heads, 12, shift).to("cuda").eval()
    q = torch.randn(B, H, W, C, device="cuda") * 2.0
    k = torch.randn(B, H, W, C, device="cuda") * 2.0
    v = torch.randn(B, H, W, C, device="cuda")
    got = ops.fusion_window_attention(torch.cat([q, k], dim=-1), v, heads, 12, shift)

    def win(t):
        if shift > 0:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        return _to_windows(t, 12)                                             # [nWB, N, C]
    d = C // heads
    qw, kw, vw = (win(t).view(-1, 144, heads, d).transpose(1, 2) for t in (q, k, v))
    attn = (qw * d ** -0.5) @ kw.transpose(-2, -1)
    if blk.attn_mask is not None:
        nW = blk.attn_mask.shape[0]
        attn = (attn.view(-1, nW, heads, 144, 144) + blk.attn_mask[None, :, None]).view(-1, heads, 144, 144)
    o = (torch.softmax(attn, dim=-1) @ vw).transpose(1, 2).reshape(-1, 144, C)
    o = _from_windows(o, 12, B, H, W)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    assert float((got - o).abs().max()) < 2e-5 * float(o.abs().max())


@pytest.mark.parametrize("k,cin,cout,relu", [(3, 512, 128, True), (7, 80, 128, False), (3, 36, 64, False)])
def test_conv24_matches_torch(k, cin, cout, relu):
    """oryon_conv24_f16x3 (ImageTextFusion's conv1 / guidance_projection shapes, models/fusion.py:562-570, and a ragged cin) against
    F.conv2d in fp32: <= 2e-5 of the output's maximum."""
    from oryon_amd import ops
    torch.backends.cudnn.allow_tf32 = False
    torch.manual_seed(k + cin)
    n = 3
    x = torch.randn(n, 24, 24, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * (3.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, device="cuda") * 0.1
    got = ops.conv24_f16x3(x, w, b, relu=relu)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, b, padding=k // 2)
    ref = torch.relu(ref) if relu else ref
    err = float((got.permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    assert torch.equal(got, ops.conv24_f16x3(x, w, b, relu=relu))


def test_new_entries_handle_empty_and_single_batches():
    """n = 0 / B = 0 are no-ops, n = 1 works (grid and workspace arithmetic at the small end)."""
    from oryon_amd import ops
    from oryon_amd.backbone.decoder_hip import HipDecoder
    torch.manual_seed(0)
    w = torch.randn(64, 8, 3, 3, device="cuda")
    assert ops.conv24_f16x3(torch.zeros(0, 24, 24, 8, device="cuda"), w, None).shape == (0, 24, 24, 64)
    assert ops.fusion_window_attention(torch.zeros(0, 24, 24, 256, device="cuda"), torch.zeros(0, 24, 24, 128, device="cuda"), 4, 12, 6).shape == (0, 24, 24, 128)
    dec = _decoder(9)
    hip = HipDecoder(dec, torch.device("cuda"))
    x, g2, g3 = torch.randn(1, 128, 24, 24, device="cuda"), torch.randn(1, 256, 48, 48, device="cuda"), torch.randn(1, 128, 96, 96, device="cuda")
    lg, fm = hip.forward(x, g2, g3)
    with torch.no_grad():
        lg0, fm0 = dec(x.view(1, 128, 1, 24, 24), [None, g2, g3])
    assert float((fm - fm0).abs().max() / fm0.abs().max()) < 2e-5 and float((lg - lg0[:, 0]).abs().max() / lg0.abs().max()) < 2e-5


def test_fusion_class_layer_matches_torch():
    """oryon_fusion_class_layer_f32 (AvgPool -> LayerNorm -> guided linear attention over T = 1 -> MLP -> bilinear upsampling -> residuals,
    models/fusion.py:300-332) against the torch module: <= 1e-5 of the output's maximum."""
    from oracle import oryon_oracle as orc
    from oryon_amd.backbone import fusion as F_
    torch.manual_seed(11)
    layer = F_._ClassLayer(128, 128, 4, (6, 6)).eval()
    layer.load_state_dict(orc.analytic_state_dict(layer.state_dict(), seed=5), strict=True)
    layer = layer.to("cuda")
    B = 5
    x = torch.randn(B, 24, 24, 128, device="cuda").permute(0, 3, 1, 2).unsqueeze(2)          # [B,128,1,24,24] on NHWC storage
    g = torch.randn(B, 1, 128, device="cuda")
    with torch.no_grad():
        ref = layer(x, g)
        F_.FP16X3_LINEAR = F_.FUSED_KERNELS = True
        try:
            got = layer(x, g)
        finally:
            F_.FP16X3_LINEAR = F_.FUSED_KERNELS = False
    assert got.shape == ref.shape
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    # the update itself (output minus the residual map) to the same bar relative to ITS size
    du, dr = got - x, ref - x
    assert float((du - dr).abs().max() / dr.abs().max()) < 1e-4


def test_hip_decoder_groupnorm_with_a_large_dc_offset():
    """GroupNorm statistics from E[x^2] - mean^2 in fp32 lose the variance once |mean| >> std.  The tile partials are now
    (sum, M2 about the tile mean) merged with the parallel-variance formula: a decoder whose up-convolution biases put a DC offset of
    ~100 standard deviations on every convolution output must still agree with the torch fp32 modules (Welford-style GroupNorm)."""
    from oryon_amd.backbone.decoder_hip import HipDecoder
    torch.backends.cudnn.allow_tf32 = False
    torch.manual_seed(9)
    dec = _decoder(11)
    with torch.no_grad():
        for blk in (dec.decoder1, dec.decoder2, dec.decoder3):
            blk.up.bias += 40.0                                                # every conv1 input channel rides on +40
            for j in (0, 3):
                blk.conv.double_conv[j].weight[:, :, 1, 1] += 2.0              # centre taps with a non-zero sum: the offset survives the
                                                                               # convolution, the zero padding does not modulate it
    n, h, w_ = 2, 24, 24
    x = torch.randn(n, 128, h, w_, device="cuda") * 0.1
    g2 = torch.randn(n, 256, 2 * h, 2 * w_, device="cuda")
    g3 = torch.randn(n, 128, 4 * h, 4 * w_, device="cuda")
    with torch.no_grad():
        pg = [proj(g) for proj, g in zip(dec.decoder_guidance_projection, (g2, g3))]
        c1 = dec.decoder1.conv.double_conv[0](torch.cat([dec.decoder1.up(x), pg[0]], dim=1))
        grp = c1.view(n, 4, 16, -1)
        ratio = float((grp.mean(dim=(2, 3)).abs() / grp.std(dim=(2, 3))).min())
        assert ratio > 30, ratio                                               # the regime the fix is for
        logits_ref, fm_ref = dec(x.view(n, 128, 1, h, w_), [None, g2, g3])
        lg, fm = HipDecoder(dec, x.device).forward(x, g2, g3)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(fm, fm_ref) < 1e-4 and rel(lg, logits_ref[:, 0]) < 1e-4, (rel(fm, fm_ref), rel(lg, logits_ref[:, 0]))


def test_modules_copy_and_pickle_after_the_fast_path_ran():
    """the fast path's caches (HipDecoder handle, q | k weights, clip_conv view) no longer live in module.__dict__ -
    a decoder / fusion module that ran it deep-copies, pickles and torch.save()s, and the copy builds its own handle."""
    import copy, io, pickle
    from oryon_amd.backbone import fusion as F_
    dec = _decoder(4)
    x = torch.randn(1, 128, 1, 24, 24, device="cuda")
    g2, g3 = torch.randn(1, 256, 48, 48, device="cuda"), torch.randn(1, 128, 96, 96, device="cuda")
    F_.enable_hip_decoder(True)
    try:
        with torch.no_grad():
            lg, fm = dec(x, [None, g2, g3])
            assert F_._fast_cache(dec).get("hip")
            dec2 = copy.deepcopy(dec)
            blob = pickle.dumps(dec)
            buf = io.BytesIO()
            torch.save(dec, buf)
            assert not F_._fast_cache(dec2).get("hip")                         # nothing travelled with the copy
            lg2, fm2 = dec2(x, [None, g2, g3])
            assert F_._fast_cache(dec2)["hip"] is not F_._fast_cache(dec)["hip"]
            assert torch.equal(fm2, fm) and torch.equal(lg2, lg)
            dec3 = pickle.loads(blob)
            lg3, fm3 = dec3(x, [None, g2, g3])
            assert torch.equal(fm3, fm)
    finally:
        F_.enable_hip_decoder(False)
