"""The three checkpoint loaders of the drop-in (VERDICT r04 "missing 3"), on fabricated files with analytic / seeded weights:
  * get_pointdsc_solver   - utils/pointdsc/init.py:32-57 (PointDSC_3DMatch_release/{config.json, models/model_best.pkl})
  * Oryon.load_catseg_checkpoint - net.py:99-133 (catseg.pth: sem_seg_head.predictor.{transformer,clip_model}.* key remap)
  * run_test.load_oryon_checkpoint - run_test.py:42 (Lightning .ckpt: the network's tensors are the `model.*` entries of state_dict)
Every tensor the file holds for the module must arrive (no silent `strict=False` miss), nothing else may move, and the forward of the
loaded module equals the forward of a module that received the same tensors directly."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RELEASE_CONFIG = {  # the fields of the released PointDSC config.json that init.py:41-50 reads, plus ones it ignores
    "in_dim": 6, "num_layers": 12, "num_channels": 128, "num_iterations": 10, "ratio": 0.1, "sigma_d": 0.1, "k": 40,
    "inlier_threshold": 0.1, "dataset": "3DMatch", "descriptor": "fcgf", "batch_size": 16, "seed_ratio": 0.1,
}


def _pointdsc_tree(tmp_path, cfg=RELEASE_CONFIG):
    from oracle import oryon_oracle as orc
    rel = tmp_path / "snapshot" / "PointDSC_3DMatch_release"
    (rel / "models").mkdir(parents=True)
    json.dump(cfg, open(rel / "config.json", "w"))
    state = orc.analytic_pointdsc_params(cfg["num_layers"], cfg["num_channels"], sigma_d=cfg["sigma_d"])
    torch.save(state, rel / "models" / "model_best.pkl")
    return state


def test_get_pointdsc_solver_restores_every_tensor(tmp_path):
    from oryon_amd.pointdsc import get_pointdsc_solver
    state = _pointdsc_tree(tmp_path)
    m = get_pointdsc_solver(str(tmp_path), "cpu")
    sd = m.state_dict()
    assert set(sd) == set(state), (sorted(set(sd) ^ set(state))[:8])          # the released file and the module agree key for key
    for k, v in state.items():
        assert sd[k].shape == v.shape and torch.equal(sd[k].cpu(), v.to(sd[k].dtype)), k
    assert not m.training and all(not p.requires_grad for p in m.parameters())
    # constructor arguments come from config.json, nms_radius from inlier_threshold (init.py:49)
    assert (m.num_iterations, m.ratio, m.k) == (10, 0.1, 40) and abs(float(m.nms_radius) - 0.1) < 1e-12 and abs(float(m.inlier_threshold) - 0.1) < 1e-12
    assert len([k for k in sd if k.startswith("encoder.blocks.")]) > 0 and "sigma" in sd and "sigma_spat" in sd


@pytest.mark.gpu
def test_get_pointdsc_solver_forward_equals_directly_loaded_module(tmp_path):
    from oryon_amd.pointdsc import PointDSC, get_pointdsc_pose, get_pointdsc_solver
    from oryon_amd.synth import make_pair
    state = _pointdsc_tree(tmp_path)
    loaded = get_pointdsc_solver(str(tmp_path), "cuda")
    direct = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
    direct.load_state_dict(state, strict=True)
    direct = direct.cuda().eval()
    g = torch.Generator().manual_seed(3)
    src = torch.rand(300, 3, generator=g) * 0.4
    ang = torch.tensor(0.3)
    R = torch.tensor([[torch.cos(ang), -torch.sin(ang), 0.0], [torch.sin(ang), torch.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    tgt = src @ R.T + torch.tensor([0.05, -0.02, 0.1]) + 1e-3 * torch.randn(300, 3, generator=g)
    Ta = get_pointdsc_pose(loaded, src, tgt, "cuda")
    Tb = get_pointdsc_pose(direct, src, tgt, "cuda")
    assert torch.equal(Ta, Tb)
    assert float((Ta[:3, :3] - R).abs().max()) < 2e-2


@pytest.fixture(scope="module")
def tiny_nets():
    """Two Oryon networks with full interface widths and one layer per CLIP tower, different seeds: `src` provides the tensors a
    checkpoint holds, `dst` is the freshly constructed network they are loaded into."""
    from oryon_amd.backbone.clip import CLIPConfig
    from oryon_amd.net import Oryon, default_model_args
    cfg = CLIPConfig(v_layers=1, t_layers=1)
    torch.manual_seed(11)
    src = Oryon(default_model_args(), "cpu", clip_cfg=cfg).eval()
    torch.manual_seed(12)
    dst = Oryon(default_model_args(), "cpu", clip_cfg=cfg).eval()
    return src, dst, cfg


def _batch():
    gen = torch.Generator().manual_seed(0)
    toks = torch.randint(1, 49000, (1, 80, 77), generator=gen)
    toks[..., 10] = 49407
    toks[..., 11:] = 0
    return {"anchor": {"rgb": torch.rand(1, 3, 224, 224, generator=gen)}, "query": {"rgb": torch.rand(1, 3, 224, 224, generator=gen)},
            "prompt_tokens": toks.contiguous()}


def _to_catseg_key(k):
    """Inverse of net.py:104-133: where a tensor of the Oryon network lives in catseg.pth (None = not part of CATSeg)."""
    T = "sem_seg_head.predictor.transformer."
    if k.startswith("fusion.clip_conv."):
        return None                                                   # new in Oryon (net.py:100: initialised, never loaded)
    if k.startswith("fusion."):
        return T + k[len("fusion."):]
    if k.startswith("decoder.decoder"):
        return T + "decoder" + k[len("decoder.decoder"):]
    if k.startswith("decoder.head"):
        return T + "head" + k[len("decoder.head"):]
    if k.startswith("vlm.clip_model."):
        return "sem_seg_head.predictor.clip_model." + k[len("vlm.clip_model."):]
    return None


def test_load_catseg_checkpoint_remaps_every_tensor(tmp_path, tiny_nets):
    import copy
    src, dst0, _ = tiny_nets
    dst = copy.deepcopy(dst0)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    src_sd = src.state_dict()
    model, moved = {}, {}
    for k, v in src_sd.items():
        ck = _to_catseg_key(k)
        if ck is not None:
            model[ck] = v.clone()
            moved[k] = ck
    # every decoder / fusion / CLIP tensor has a place in the file except fusion.clip_conv.*; Swin (torchvision weights) is not in it
    assert all(k.startswith(("guidance_backbone.", "fusion.clip_conv.")) for k in src_sd if k not in moved), \
        [k for k in src_sd if k not in moved and not k.startswith(("guidance_backbone.", "fusion.clip_conv."))][:8]
    # what else a detectron2 CATSeg file carries and the remap must ignore
    model["backbone.stem.conv1.weight"] = torch.randn(4, 3, 3, 3)
    model["sem_seg_head.predictor.text_features_test"] = torch.randn(5, 7)
    model["sem_seg_head.predictor.upsample1.weight"] = torch.randn(8, 8, 2, 2)
    model["criterion.empty_weight"] = torch.ones(3)
    path = tmp_path / "catseg.pth"
    torch.save({"model": model, "iteration": 79999}, path)
    dst.load_catseg_checkpoint(str(path))
    after = dst.state_dict()
    for k, ck in moved.items():
        assert torch.equal(after[k], src_sd[k]), (k, ck)
    for k in after:
        if k not in moved:
            assert torch.equal(after[k], before[k]), k                  # Swin guidance and fusion.clip_conv stay as constructed
    assert any(k.startswith("decoder.decoder") for k in moved) and any(k.startswith("decoder.head") for k in moved)
    assert sum(k.startswith("vlm.clip_model.") for k in moved) > 20 and sum(k.startswith("fusion.layers.") for k in moved) > 20
    # forward == a network that received the same tensors directly
    direct = copy.deepcopy(dst0)
    res = direct.load_state_dict({k: src_sd[k] for k in moved}, strict=False)
    assert res.unexpected_keys == []
    xs = _batch()
    with torch.no_grad():
        a, b = dst(xs), direct(xs)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # ... and differs from the freshly constructed one (the load did something)
    with torch.no_grad():
        c = copy.deepcopy(dst0)(xs)
    assert not torch.equal(a["featmap_a"], c["featmap_a"])


def test_load_oryon_checkpoint_from_lightning_file(tmp_path, tiny_nets):
    import copy
    import run_test
    src, dst0, _ = tiny_nets
    dst = copy.deepcopy(dst0)
    src_sd = src.state_dict()
    blob = {"state_dict": {"model." + k: v.clone() for k, v in src_sd.items()}, "epoch": 19, "global_step": 12345,
            "pytorch-lightning_version": "1.9.0", "optimizer_states": [], "lr_schedulers": []}
    blob["state_dict"]["loss.feature_loss.temperature"] = torch.tensor(0.07)      # a non-network entry of the LightningModule
    path = tmp_path / "epoch=0019.ckpt"
    torch.save(blob, path)
    stats = run_test.load_oryon_checkpoint(dst, str(path))
    assert stats == {"tensors": len(src_sd), "missing": 0, "unexpected": 0}, stats
    after = dst.state_dict()
    assert set(after) == set(src_sd)
    for k, v in src_sd.items():
        assert torch.equal(after[k], v), k
    xs = _batch()
    with torch.no_grad():
        a, b = dst(xs), src(xs)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a bare state dict (no Lightning wrapper, no prefix) loads as well
    dst2 = copy.deepcopy(dst0)
    torch.save(src_sd, tmp_path / "plain.pth")
    stats2 = run_test.load_oryon_checkpoint(dst2, str(tmp_path / "plain.pth"))
    assert stats2["missing"] == 0 and stats2["unexpected"] == 0
    assert all(torch.equal(dst2.state_dict()[k], v) for k, v in src_sd.items())
