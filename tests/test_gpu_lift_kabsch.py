"""GPU parity tests for K2 (scale/validate/lift) and K8 (batched weighted Kabsch + 3x3 SVD)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name), allow_pickle=False).items()}


def names(prefix):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, prefix + "*.npz")))


@pytest.mark.parametrize("name", names("g2_lift_"))
def test_lift_pairs_bit_exact(name):
    from oryon_amd import ops
    g = load(name)
    dev = "cuda"
    corrs = torch.from_numpy(g["corrs"].astype(np.int32)).to(dev)[None]
    da = torch.from_numpy(g["depth_a"].astype(np.float32)).to(dev)[None]
    dq = torch.from_numpy(g["depth_q"].astype(np.float32)).to(dev)[None]
    ca = torch.from_numpy(g["cam_a"]).to(torch.float32).to(dev)[None]
    cq = torch.from_numpy(g["cam_q"]).to(torch.float32).to(dev)[None]
    pa, pq, n = ops.lift_pairs(corrs, None, tuple(g["feat_hw"]), da, dq, ca, cq)
    m = int(n.item())
    assert m == int(g["valid"].sum())
    assert np.array_equal(pa[0, :m].cpu().numpy(), g["pcd_a"])      # bit-exact vs the reference
    assert np.array_equal(pq[0, :m].cpu().numpy(), g["pcd_q"])


@pytest.mark.parametrize("name", names("g2_lift_"))
def test_lift_pcd_dropin(name):
    from oryon_amd import pcd
    g = load(name)
    dev = "cuda"
    da = torch.from_numpy(g["depth_a"].astype(np.float32)).to(dev)
    pix = torch.from_numpy(g["pix_a"].astype(np.int64)).to(dev)
    out = pcd.lift_pcd(da.unsqueeze(-1), torch.from_numpy(g["cam_a"]).to(dev), (pix[:, 1], pix[:, 0]))
    assert out.dtype == torch.float32 and out.device.type == "cuda"
    # torch's GPU `x / 1000.` multiplies by a rounded reciprocal; the golden (CPU) uses a true division
    assert np.array_equal((out.cpu() / 1000.).numpy(), g["pcd_a"])


def test_lift_status_skips_and_ncorr():
    from oryon_amd import ops
    g = load("g2_lift_nocs.npz")
    dev = "cuda"
    B = 3
    corrs = torch.from_numpy(g["corrs"].astype(np.int32)).to(dev)[None].repeat(B, 1, 1).contiguous()
    da = torch.from_numpy(g["depth_a"].astype(np.float32)).to(dev)[None].repeat(B, 1, 1).contiguous()
    dq = torch.from_numpy(g["depth_q"].astype(np.float32)).to(dev)[None].repeat(B, 1, 1).contiguous()
    ca = torch.from_numpy(g["cam_a"]).to(torch.float32).to(dev)[None].repeat(B, 1).contiguous()
    status = torch.tensor([0, 2, 0], dtype=torch.int32, device=dev)
    ncorr = torch.tensor([300, 300, 100], dtype=torch.int32, device=dev)
    pa, pq, n = ops.lift_pairs(corrs, ncorr, tuple(g["feat_hw"]), da, dq, ca, ca, status)
    n = n.cpu().numpy()
    assert n[1] == 0 and n[0] == int(g["valid"].sum()) and n[2] == int(g["valid"][:100].sum())


def test_kabsch_golden():
    from oryon_amd import ops
    g = load("g3_kabsch.npz")
    dev = "cuda"
    A, B, w = (torch.from_numpy(g[k]).to(dev) for k in ("A", "B", "w"))
    T = ops.kabsch_batched(A, B, w).cpu().numpy()
    T0 = ops.kabsch_batched(A, B, None).cpu().numpy()
    for b in range(A.shape[0]):
        R = T[b, :3, :3]
        assert abs(np.linalg.det(R) - 1) < 1e-5 and np.abs(R @ R.T - np.eye(3)).max() < 1e-5
        if b == 5:      # single effective point: H == 0, the rotation is arbitrary in the reference too
            continue
        assert np.abs(T[b] - g["T"][b]).max() <= 1e-4 * max(1.0, np.abs(g["T"][b]).max()), b
        assert np.abs(T0[b] - g["T_noweights"][b]).max() <= 1e-4 * max(1.0, np.abs(g["T_noweights"][b]).max()), b


def test_kabsch_large_batch_vs_oracle():
    from oracle import oryon_oracle as orc
    from oryon_amd import ops
    gen = torch.Generator().manual_seed(5)
    nb, m = 3200, 40
    A = torch.randn(nb, m, 3, generator=gen) * 0.1 + torch.tensor([0.0, 0.0, 0.8])
    ang = torch.rand(nb, generator=gen) * 1.0
    ax = torch.nn.functional.normalize(torch.randn(nb, 3, generator=gen), dim=1)
    K = torch.zeros(nb, 3, 3)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    R = torch.eye(3)[None] + torch.sin(ang)[:, None, None] * K + (1 - torch.cos(ang))[:, None, None] * (K @ K)
    Bp = A @ R.transpose(1, 2) + torch.randn(nb, 1, 3, generator=gen) * 0.1 + 0.003 * torch.randn(nb, m, 3, generator=gen)
    w = torch.rand(nb, m, generator=gen)
    T = ops.kabsch_batched(A.cuda(), Bp.cuda(), w.cuda()).cpu()
    Tref = orc.kabsch(A, Bp, w)
    assert float((T - Tref).abs().max()) < 1e-4
    # size-independent property: residual of the fitted transform is at the noise level
    res = (A @ T[:, :3, :3].transpose(1, 2) + T[:, None, :3, 3] - Bp).norm(dim=-1).mean()
    assert float(res) < 0.01


def test_pose_metrics_on_device_match_reference_golden():
    """f3: oryon_pose_metrics against the outputs of the reference's utils/metrics.py (tests/golden/g7_metrics.npz): ADD within 2e-3
    relative of the reference's float16 statistic (same float16 point transform, fp32 instead of half norms / mean), ADD-S within 1e-5,
    rotation / translation errors within 1e-3 deg / 1e-4 cm; and a two-model batch through the offset table."""
    import os
    from oryon_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_metrics.npz"))
    dev = "cuda"
    pred, gt = torch.from_numpy(g["pred"]).float().to(dev), torch.from_numpy(g["gt"]).float().to(dev)
    pts = torch.from_numpy(g["pcd"]).float().to(dev)
    out = ops.pose_metrics(pred, gt, pts).cpu().numpy()
    np.testing.assert_allclose(out[:, 0], g["add"].astype(np.float64), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(out[:, 1], g["adds"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out[:, 2], g["theta"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(out[:, 3], g["shift"], rtol=1e-5, atol=1e-4)
    # two models of different sizes in one call
    pts2 = torch.cat((pts, pts[:37] * 0.5))
    off = torch.tensor([0, pts.shape[0], pts.shape[0] + 37], dtype=torch.int32)
    which = torch.tensor([i % 2 for i in range(pred.shape[0])], dtype=torch.int32)
    out2 = ops.pose_metrics(pred, gt, pts2, off, which).cpu().numpy()
    small = ops.pose_metrics(pred, gt, pts[:37] * 0.5).cpu().numpy()
    assert np.allclose(out2[0::2], out[0::2], rtol=1e-6) and np.allclose(out2[1::2], small[1::2], rtol=1e-6)
