"""GPU parity tests added in round 2 (VERDICT r01 "what's weak" 1-4): the DEFAULT (screened) matcher kernels directly against the
reference's goldens and the C oracle - at golden sizes and at BASELINE cfg2 / cfg4 pair sizes -, direct tests of the two device
sampling kernels (K0 subsample, K1b select), the near-threshold sigmoid mask, the fusion + decoder forward on ROCm against the
reference's G5 golden, and a slice of the PointDSC stress sweep against the CPU oracle."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODES = ("exact", "screened16", "screened8", "screened6")


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name), allow_pickle=False).items()}


def names(prefix):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, prefix + "*.npz")) if "matcher_half" not in p)


def _presample(g, mode):
    from oryon_amd import pcd
    dev = "cuda"
    pre = pcd.match_presample(torch.from_numpy(g["feats1"]).to(dev), torch.from_numpy(g["feats2"]).to(dev),
                              torch.from_numpy(g["mask1"]).to(dev), torch.from_numpy(g["mask2"]).to(dev), float(g["threshold"]), mode=mode)
    return {k: v.cpu().numpy() for k, v in pre.items()}


def _check_vs_c_oracle(pre, ref, mode, thr):
    """exact: every row bit for bit.  screened: identical valid set; argmin / min_dist bit-identical on valid rows; a ruled-out row
    reports an estimate that is itself not below the threshold (minus the screen's error bound)."""
    assert np.array_equal(pre["valid"], ref["valid"]), f"{mode}: valid set differs from the C oracle"
    v = ref["valid"].astype(bool)
    if mode == "screened6":
        # the engine's default route is lazy: exact validity of every row (above), exact argmin for the rows it sampled - each sampled
        # correspondence must be (a valid anchor pixel, the pixel of the ORACLE's argmin for that anchor)
        corrs = pre["corrs"]
        if v.sum() <= 1:
            assert len(corrs) == 0 and int(pre["status"]) in (1, 2)          # NO_MASK (an empty ROI) / NO_CORR
            return
        assert len(corrs) == 500 and int(pre["status"]) == 0
        W = int(pre["roi1"][:, 1].max()) + 1 + int(pre["roi2"][:, 1].max()) + 1          # any common multiplier > every x
        key1 = pre["roi1"][:, 0] * W + pre["roi1"][:, 1]
        order = np.argsort(key1, kind="stable")
        rows = order[np.searchsorted(key1[order], corrs[:, 0] * W + corrs[:, 1])]
        assert np.array_equal(pre["roi1"][rows], corrs[:, :2]) and v[rows].all(), "a sampled anchor is not a valid anchor row"
        # duplicate anchor pixels cannot occur (nonzero of a mask): rows are unique per pixel
        assert np.array_equal(pre["roi2"][ref["argmin"][rows]], corrs[:, 2:]), f"{mode}: a sampled query pixel is not the oracle's argmin"
        assert np.array_equal(pre["argmin"][rows], ref["argmin"][rows])
        if v.sum() >= 500:
            assert len(np.unique(rows)) == 500
        return
    assert np.array_equal(pre["argmin"][v], ref["argmin"][v]), f"{mode}: argmin differs on valid rows"
    assert np.array_equal(pre["min_dist"][v].view(np.uint32), ref["min_dist"][v].view(np.uint32)), f"{mode}: min_dist bits differ"
    if mode == "exact":
        assert np.array_equal(pre["argmin"], ref["argmin"])
        assert np.array_equal(pre["min_dist"].view(np.uint32), ref["min_dist"].view(np.uint32))
    else:
        same = pre["min_dist"] == ref["min_dist"]
        assert np.all(same | (pre["min_dist"] >= thr - 0.2)), f"{mode}: a ruled-out row reports a distance far below the threshold"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", names("g1_matcher_"))
def test_matcher_modes_vs_golden_and_c_oracle(name, mode):
    """Every matcher mode - the engine's default int8 -> fp16 -> fp32 cascade included - against the reference's own outputs (G1)
    and the C oracle (utils/pcd.py:202-205)."""
    from oracle import c_oracle
    g = load(name)
    pre = _presample(g, mode)
    if mode == "screened6" and "corrs" not in pre:
        mode = "exact"                   # thresholds outside (0, 0.5] take the exact scan in every screened mode (match_presample)
    assert np.array_equal(pre["roi1"], g["roi1"]) and np.array_equal(pre["roi2"], g["roi2"])
    if "min_dist" not in g:
        return
    thr = float(g["threshold"])
    # (1) the reference's own outputs: valid set away from the cut, argmin where its top-2 gap is resolvable, first index on ties
    far = np.abs(g["min_dist"] - thr) > 1e-6
    assert np.array_equal(pre["valid"][far], g["valid"][far])
    v = pre["valid"].astype(bool)
    if mode == "screened6":
        # lazy route: argmin / min_dist exist for the sampled rows only - compared through the correspondences below (C oracle) and,
        # against the reference's own argmin, on the sampled rows whose top-2 gap the reference resolves
        if len(pre["corrs"]) and "gap" in g:
            W = int(max(g["roi1"][:, 1].max(), g["roi2"][:, 1].max())) + 1
            key1 = g["roi1"][:, 0] * W + g["roi1"][:, 1]
            order = np.argsort(key1, kind="stable")
            rows = order[np.searchsorted(key1[order], pre["corrs"][:, 0] * W + pre["corrs"][:, 1])]
            ok = (g["gap"][rows] > 1e-6) | (g["n_at_min"][rows] > 1)
            assert np.array_equal(g["roi2"][g["argmin"][rows]][ok], pre["corrs"][:, 2:][ok])
    else:
        np.testing.assert_allclose(pre["min_dist"][v], g["min_dist"][v], rtol=0, atol=1e-6)
        clear = (g["gap"] > 1e-6) & v
        assert np.array_equal(pre["argmin"][clear], g["argmin"][clear])
        tied = (g["n_at_min"] > 1) & v
        assert np.array_equal(pre["argmin"][tied], g["argmin"][tied])
    if mode == "exact":
        np.testing.assert_allclose(pre["min_dist"], g["min_dist"], rtol=0, atol=1e-6)
    # (2) the C oracle
    ref = c_oracle.match_presample(g["feats1"], g["feats2"], g["mask1"], g["mask2"], thr)
    _check_vs_c_oracle(pre, ref, mode, thr)


@pytest.mark.parametrize("H,C", [(224, 256), (384, 512)])            # BASELINE cfg2 / cfg4 pair sizes
def test_full_size_pair_all_modes_vs_c_oracle(H, C):
    """One pair at the full BASELINE size (N1 = 5000 subsampled anchors x every query pixel with depth x C) through all three matcher
    modes, each compared with the C oracle's scan of the same rows (51 / 330 G fmaf on the host cores)."""
    from oracle import c_oracle
    from oryon_amd import ops
    from oryon_amd.synth import make_pair
    dev = "cuda"
    p = make_pair(2, H, H, C, device=dev)
    roi_a, na = ops.roi_compact(p["mask_a"])
    roi_q, nq = ops.roi_compact(p["mask_q"])
    ops.roi_subsample_(roi_a, na, 5000, seed=1)
    n1, n2 = int(na), int(nq)
    assert n1 == 5000 and n2 > 0.4 * H * H
    fa, fq = p["feat_a"][None].contiguous(), p["feat_q"][None].contiguous()
    ref_md, ref_am, ref_va = c_oracle.match_lin(p["feat_a"].cpu().numpy(), p["feat_q"].cpu().numpy(), roi_a[0, :n1].cpu().numpy(),
                                                roi_q[0, :n2].cpu().numpy(), 0.25)
    ref = dict(min_dist=ref_md, argmin=ref_am.astype(np.int64), valid=ref_va)
    assert ref_va.mean() > 0.6
    cap_a, cap_q = ops.round_up(n1, 256), ops.round_up(n2, 256)
    for mode in MODES[:3]:          # the lazy default route ("screened6") at these sizes: tests/test_gpu_default_route_vs_oracle.py
        if mode == "exact":
            a_hat = ops.gather_normalise(fa, roi_a, na, cap_a)
            q_hat = ops.gather_normalise(fq, roi_q, nq, cap_q)
            md, am, va = ops.match(a_hat, q_hat, na, nq, 0.25)
        elif mode == "screened16":
            a_hat, a16 = ops.gather_normalise(fa, roi_a, na, cap_a, c_pad=C, want_f16=True)
            q_hat, q16 = ops.gather_normalise(fq, roi_q, nq, cap_q, c_pad=C, want_f16=True)
            md, am, va = ops.match_screened(a_hat, q_hat, a16, q16, na, nq, 0.25)
        else:
            a_hat, _, a8, a_sc, _ = ops.gather_normalise_q8(fa, roi_a, na, cap_a, C)
            q_hat, _, q8, q_sc, q_eps = ops.gather_normalise_q8(fq, roi_q, nq, cap_q, C)
            md, am, va = ops.match_screened8(a_hat, q_hat, a8, q8, a_sc, q_sc, q_eps, na, nq, 0.25, C)
        pre = dict(min_dist=md[0, :n1].cpu().numpy(), argmin=am[0, :n1].cpu().numpy().astype(np.int64),
                   valid=va[0, :n1].cpu().numpy().astype(bool))
        _check_vs_c_oracle(pre, ref, mode, 0.25)
        del a_hat, q_hat


# ------------------------------------------------------------------------------------------------ K0 subsample / K1b select
def test_roi_subsample_direct():
    """oryon_roi_subsample (replaces torch_sample_select on the anchor ROI, utils/pcd.py:188-190): lists longer than max_keep are cut
    to exactly max_keep entries that are a duplicate-free subset of the list in the original (row-major) order; shorter lists are
    untouched; the draw depends on (seed, map key) only; every entry is kept with the same probability."""
    from oryon_amd import ops
    dev = "cuda"
    H = W = 64
    n_maps = 6
    g = torch.Generator(device="cpu").manual_seed(3)
    mask = (torch.rand((n_maps, H, W), generator=g) < 0.8).int()
    mask[1] = 0
    mask[1].view(-1)[:100] = 1                                    # 100 < max_keep: untouched
    mask[2] = 0                                                   # empty
    roi0, cnt0 = ops.roi_compact(mask.to(dev))
    keys = torch.tensor([10, 11, 12, 13, 14, 10], dtype=torch.int64, device=dev)
    roi, cnt = roi0.clone(), cnt0.clone()
    ops.roi_subsample_(roi, cnt, 1000, seed=1, map_key=keys)
    c0, c1 = cnt0.tolist(), cnt.tolist()
    for m in range(n_maps):
        before = roi0[m, :c0[m]].cpu().numpy()
        after = roi[m, :c1[m]].cpu().numpy()
        if c0[m] <= 1000:
            assert c1[m] == c0[m] and np.array_equal(before, after)
            continue
        assert c1[m] == 1000
        assert np.all(np.diff(after) > 0)                          # order kept, no duplicates
        assert np.isin(after, before).all()
    # same (seed, key) -> same draw; a different key or seed -> a different one
    r2, c2 = roi0.clone(), cnt0.clone()
    ops.roi_subsample_(r2, c2, 1000, seed=1, map_key=keys)
    assert torch.equal(r2[0, :1000], roi[0, :1000])
    pos0 = np.searchsorted(roi0[0, :c0[0]].cpu().numpy(), roi[0, :1000].cpu().numpy())
    pos5 = np.searchsorted(roi0[5, :c0[5]].cpu().numpy(), roi[5, :1000].cpu().numpy())
    if c0[0] == c0[5]:
        assert np.array_equal(pos0, pos5)                          # maps 0 and 5 share key 10: same positions kept
    r3, c3 = roi0.clone(), cnt0.clone()
    ops.roi_subsample_(r3, c3, 1000, seed=2, map_key=keys)
    assert not torch.equal(r3[0, :1000], roi[0, :1000])
    # uniformity: 10^4 keys x keep 64 of 256 -> every position kept ~2500 times (chi-square, 255 dof: mean 255, sd 22.6)
    n_keys, n, keep = 10000, 256, 64
    lst = torch.arange(n, dtype=torch.int32, device=dev).repeat(n_keys, 1).contiguous()
    cn = torch.full((n_keys,), n, dtype=torch.int32, device=dev)
    ops.roi_subsample_(lst, cn, keep, seed=7, map_key=torch.arange(n_keys, dtype=torch.int64, device=dev))
    assert int(cn.min()) == keep and int(cn.max()) == keep
    hist = torch.bincount(lst[:, :keep].reshape(-1).long(), minlength=n).double().cpu().numpy()
    expect = n_keys * keep / n
    chi2 = float(((hist - expect) ** 2 / (expect * (1 - keep / n))).sum())
    assert 150 < chi2 < 370, chi2


def test_select_corrs_direct():
    """oryon_select_corrs (utils/pcd.py:205-214 + utils/misc.py:242-254): rows come from valid anchors only and carry
    (roi_a[row], roi_q[argmin[row]]) as (y1,x1,y2,x2); exactly max_corrs rows; without replacement when n_valid >= max_corrs, with
    replacement otherwise; <= 1 valid row -> NO_CORR; an empty ROI -> NO_MASK; sharding-invariant keys; uniform draws."""
    from oryon_amd import ops
    dev = "cuda"
    W, B, cap_a, max_corrs = 50, 6, 1024, 500
    g = torch.Generator(device="cpu").manual_seed(9)
    n_a = torch.tensor([900, 900, 900, 900, 0, 700], dtype=torch.int32)
    n_q = torch.tensor([2000, 2000, 2000, 2000, 2000, 0], dtype=torch.int32)
    roi_a = torch.stack([torch.sort(torch.randperm(W * W, generator=g)[:cap_a]).values for _ in range(B)]).int()
    roi_q = torch.stack([torch.sort(torch.randperm(W * W, generator=g)[:2048]).values for _ in range(B)]).int()
    argmin = torch.randint(0, 2000, (B, cap_a), generator=g).int()
    valid = torch.zeros((B, cap_a), dtype=torch.uint8)
    valid[0, :900] = (torch.rand(900, generator=g) < 0.8).to(torch.uint8)           # ~720 valid >= 500: without replacement
    valid[1, torch.randperm(900, generator=g)[:120]] = 1                            # 120 valid < 500: with replacement
    valid[2, 77] = 1                                                                # exactly one valid row: NO_CORR
    valid[3, [5, 600]] = 1                                                          # two valid rows: OK, with replacement
    valid[4, :10] = 1                                                               # n_a = 0: NO_MASK whatever valid says
    valid[5, :10] = 1                                                               # n_q = 0: NO_MASK
    valid[:, 900:] = 1                                                              # rows >= n_a must be ignored
    key = torch.arange(100, 100 + B, dtype=torch.int64)
    to = lambda t: t.to(dev).contiguous()
    corrs, n_valid, n_sel, status = ops.select_corrs(to(roi_a), to(roi_q), to(n_a), to(n_q), to(argmin), to(valid), W, max_corrs,
                                                     seed=1, pair_key=to(key), corr_rows=512)
    corrs, n_valid, n_sel, status = corrs.cpu(), n_valid.tolist(), n_sel.tolist(), status.tolist()
    nv = [int(valid[b, :int(n_a[b])].sum()) if int(n_a[b]) > 0 and int(n_q[b]) > 0 else 0 for b in range(B)]
    assert status == [0, 0, 2, 0, 1, 1] and n_valid == nv and n_sel == [500, 500, 0, 500, 0, 0]
    for b in (0, 1, 3):
        rows_valid = torch.nonzero(valid[b, :int(n_a[b])]).squeeze(1)
        pa, pq = roi_a[b, rows_valid].long(), roi_q[b, argmin[b, rows_valid].long()].long()
        allowed = {(int(a) // W, int(a) % W, int(q) // W, int(q) % W) for a, q in zip(pa, pq)}
        got = [tuple(r) for r in corrs[b, :500].tolist()]
        assert set(got) <= allowed
        if nv[b] >= max_corrs:
            lin = corrs[b, :500, 0] * W + corrs[b, :500, 1]
            assert len(set(lin.tolist())) == 500                                   # anchors are distinct pixels: no replacement
            assert bool((lin[1:] > lin[:-1]).all())                                # emitted in anchor-row order
        else:
            assert len(set(got)) <= nv[b]
            if nv[b] == 2:
                assert len(set(got)) == 2                                          # 500 draws from 2 rows hit both
    # the same pair under the same key elsewhere in a batch gives the same rows (sharding invariance)
    sel = [1, 0]
    c2, _, _, _ = ops.select_corrs(to(roi_a[sel]), to(roi_q[sel]), to(n_a[sel]), to(n_q[sel]), to(argmin[sel]), to(valid[sel]), W,
                                   max_corrs, seed=1, pair_key=to(key[sel]), corr_rows=512)
    assert torch.equal(c2.cpu()[0, :500], corrs[1, :500]) and torch.equal(c2.cpu()[1, :500], corrs[0, :500])
    # uniformity of the without-replacement draw: 4096 keys x 64 of 256 valid rows
    nk, n, keep = 4096, 256, 64
    ra = torch.arange(n, dtype=torch.int32).repeat(nk, 1)
    rq = torch.zeros((nk, 256), dtype=torch.int32)
    am = torch.zeros((nk, 256), dtype=torch.int32)
    va = torch.ones((nk, 256), dtype=torch.uint8)
    cn = torch.full((nk,), n, dtype=torch.int32)
    c3, _, _, st3 = ops.select_corrs(to(ra), to(rq), to(cn), to(cn), to(am), to(va), 1 << 20, keep, seed=5,
                                     pair_key=to(torch.arange(nk, dtype=torch.int64)), corr_rows=64)
    assert int(st3.abs().max()) == 0
    hist = torch.bincount(c3[:, :, 1].reshape(-1).long().cpu(), minlength=n).double().numpy()
    expect = nk * keep / n
    chi2 = float(((hist - expect) ** 2 / (expect * (1 - keep / n))).sum())
    assert 150 < chi2 < 370, chi2
    # ... and of the with-replacement draw: 4096 keys x 64 draws from 16 valid rows
    va16 = torch.zeros((nk, 256), dtype=torch.uint8)
    va16[:, 100:116] = 1
    c4, nv4, _, _ = ops.select_corrs(to(ra), to(rq), to(cn), to(cn), to(am), to(va16), 1 << 20, keep, seed=5,
                                     pair_key=to(torch.arange(nk, dtype=torch.int64)), corr_rows=64)
    assert set(nv4.tolist()) == {16}
    h4 = torch.bincount(c4[:, :, 1].reshape(-1).long().cpu(), minlength=n).double().numpy()
    assert h4[:100].sum() == 0 and h4[116:].sum() == 0
    e4 = nk * keep / 16
    chi2 = float(((h4[100:116] - e4) ** 2 / e4).sum())            # 15 dof
    assert chi2 < 45, chi2


# ------------------------------------------------------------------------------------------------ a6: sigmoid threshold
def test_mask_from_logits_near_threshold_vs_torch():
    """losses.py:58-59 `sigmoid(logit) > th` on 10^6 realistic logits plus 10^5 logits packed around the decision point: the device
    mask must equal torch's on every pixel whose sigmoid is not within 4 ulp of the threshold, and the number of such pixels is
    reported (DESIGN parity caveat 5)."""
    from oryon_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for th in (0.5, 0.3, 0.7):
        x0 = float(np.log(th / (1 - th)))
        logits = torch.cat((6.0 * torch.randn(1_000_000, generator=g, device=dev),
                            x0 + 1e-5 * torch.randn(50_000, generator=g, device=dev),
                            x0 + 1e-7 * torch.randn(50_000, generator=g, device=dev),
                            torch.tensor([x0, 0.0, -0.0, 88.0, -88.0, 104.0, -104.0, float("inf"), float("-inf")], device=dev)))
        got = ops.mask_from_logits(logits, th).bool()
        s_gpu = torch.sigmoid(logits)
        s_cpu = torch.sigmoid(logits.cpu())
        want_gpu, want_cpu = s_gpu > th, (s_cpu > th).to(dev)
        ulp = float(np.spacing(np.float32(th)))
        near = (s_cpu.to(dev) - th).abs() <= 4 * ulp
        assert torch.equal(got[~near], want_cpu[~near]) and torch.equal(got[~near], want_gpu[~near])
        # inside the 4-ulp band the three evaluations (libm expf on CPU, torch's device exp, this kernel's expf) may each flip
        flips = int((got[near] != want_cpu[near]).sum())
        assert flips <= int(near.sum())
        realistic_near = int(near[:1_000_000].sum())
        assert realistic_near <= 5, realistic_near                  # N(0,6) logits: essentially no pixel sits in the band


# ------------------------------------------------------------------------------------------------ a4 / a5 on ROCm
def test_fusion_and_decoder_match_reference_golden_on_gpu():
    """models/fusion.py:602-625 + models/decoder.py:82-108 evaluated on the MI355X (PyTorch-ROCm convolutions / GEMMs, fp32) against
    the outputs of the imported reference (tests/golden/g5_backbone.npz): <= 1e-4 relative, the north-star descriptor bar."""
    from oracle import oryon_oracle as orc
    from oryon_amd.backbone.fusion import ImageTextFusion, StandardDecoder
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    g = np.load(os.path.join(GOLD, "g5_backbone.npz"))
    dev = "cuda"
    fusion = ImageTextFusion("cpu").eval()
    decoder = StandardDecoder("cpu", True, True, input_dim=128, decoder_dims=[64, 32]).eval()
    fusion.load_state_dict(orc.analytic_state_dict(fusion.state_dict(), seed=3), strict=True)
    decoder.load_state_dict(orc.analytic_state_dict(decoder.state_dict(), seed=4), strict=True)
    fusion, decoder = fusion.to(dev), decoder.to(dev)
    B = 2
    img = orc.hashed_tensor((B, 1024, 24, 24), 100, 0, 1.0).to(dev)
    text = orc.hashed_tensor((B, 1, 80, 768), 101, 0, 1.0).to(dev)
    guid = [orc.hashed_tensor((B, 512, 24, 24), 102, 0, 1.0).to(dev), orc.hashed_tensor((B, 256, 48, 48), 103, 0, 1.0).to(dev),
            orc.hashed_tensor((B, 128, 96, 96), 104, 0, 1.0).to(dev)]
    with torch.no_grad():
        feats = fusion(img, text, guid)
        mask, featmap = decoder(feats, guid)
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
    assert rel(feats.cpu().numpy(), g["fusion_out"]) < 1e-4
    assert rel(mask[:, :, ::2, ::2].cpu().numpy(), g["mask"]) < 1e-4
    assert rel(featmap[:, :, ::4, ::4].cpu().numpy(), g["featmap_sub"]) < 1e-4
    assert rel(featmap.double().sum(dim=(2, 3)).cpu().numpy(), g["featmap_sum"]) < 1e-4
    assert rel(featmap.double().abs().sum(dim=(2, 3)).cpu().numpy(), g["featmap_abs_sum"]) < 1e-5


# ------------------------------------------------------------------------------------------------ PointDSC stress slice
def _rot(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_pointdsc_stress_slice_vs_oracle():
    """100 seeded cases of tools/stress_pointdsc.py (random motions, inlier ratios 0.25-0.95, noise, n = 41-500, three network shapes,
    two weight sets): the device registration (K3-K10 through get_pointdsc_pose, utils/pointdsc/init.py:10-29) against the CPU
    oracle's restatement of PointDSC.forward.  Bar: >= 97 % of the poses within 1e-4 of the oracle's, every other case must be a
    failed registration in BOTH (winning hypothesis explains < 40 % of the rows, DESIGN parity caveat 4)."""
    from oracle import oryon_oracle as orc
    from oryon_amd.pointdsc import PointDSC, get_pointdsc_pose
    rng = np.random.default_rng(5)
    models = {}

    def model(L, C, ps):
        if (L, C, ps) not in models:
            m = PointDSC(in_dim=6, num_layers=L, num_channels=C, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
            P = orc.analytic_pointdsc_params(L, C, seed=ps)
            m.load_state_dict(P, strict=True)
            models[(L, C, ps)] = (m.cuda().eval(), P)
        return models[(L, C, ps)]

    errs, bad = [], []
    for case in range(100):
        L, C = [(12, 128), (6, 128), (2, 32)][case % 3]
        n = int(rng.integers(41, 501))
        inl = float(rng.uniform(0.25, 0.95))
        noise = float(rng.choice([0.0, 0.002, 0.01]))
        src = rng.uniform(-0.3, 0.3, (n, 3)) + np.array([0, 0, 0.8])
        R, t = _rot(rng.normal(size=3), rng.uniform(0, 0.6)), rng.uniform(-0.1, 0.1, 3)
        tgt = src @ R.T + t + noise * rng.normal(size=(n, 3))
        out = rng.random(n) > inl
        tgt[out] = rng.uniform(-0.3, 0.3, (int(out.sum()), 3)) + np.array([0, 0, 0.8])
        m, P = model(L, C, case % 2)
        s, g = torch.from_numpy(src.astype(np.float32)), torch.from_numpy(tgt.astype(np.float32))
        cfg = dict(num_layers=L, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
        T_ref = orc.pointdsc_forward(s, g, P, cfg).numpy().reshape(4, 4)
        pose = get_pointdsc_pose(m, s.cuda(), g.cuda(), "cuda").numpy()
        e = float(np.abs(pose - T_ref).max())
        errs.append(e)
        if e > 1e-4:
            inl_gpu = float((np.linalg.norm(src @ pose[:3, :3].T + pose[:3, 3] - tgt, axis=1) < 0.1).mean())
            inl_ref = float((np.linalg.norm(src @ T_ref[:3, :3].T + T_ref[:3, 3] - tgt, axis=1) < 0.1).mean())
            bad.append((case, n, round(inl, 2), e, inl_gpu, inl_ref))
    errs = np.array(errs)
    assert (errs <= 1e-4).mean() >= 0.97, (np.sort(errs)[-5:], bad)
    for case, n, inl, e, ig, ir in bad:
        assert ig < 0.4 and ir < 0.4, f"case {case}: poses differ by {e:.2e} on a registration that succeeded (n={n}, inliers {inl})"
