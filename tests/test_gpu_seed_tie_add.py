"""what the seed-tie caveat costs in the metric that matters.

When fewer than S = int(0.1 n) strictly positive NMS maxima exist - object-sized clouds: every REAL275 / TOYL pair - the reference
completes its seed list out of an exact tie at key 0 in `argsort`'s implementation-defined order (PointDSC.py:217); this build takes
ascending index order there.  The poses of the two then agree to ~3e-3 instead of 1e-4 (DESIGN.md "parity caveats").  This test puts a
number on it: 300 (ORYON_TIE_PAIRS=1000 for the full run) object-sized synthetic registration problems (clouds <= 0.28 m across, 500 putative correspondences, 30-80 %
inliers, 1 mm noise, some with duplicated rows), registered by the HIP path (oryon_pointdsc_register) and by the CPU oracle
(oracle.oryon_oracle.pointdsc_forward = the reference's forward), scored with the reference's own metric - ADD / ADD-S with the
float16 model transform (oryon_pose_metrics, utils/evaluator.py:206-256) against the generating pose, recall at 0.1 d.
north_star bar: the recalls differ by <= 0.1 pt."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = dict(num_layers=12, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1, inlier_threshold=0.1)
import os

# 300 problems keep the suite short (the CPU oracle is ~0.3 s per problem): one flipped pair would be 0.33 pt, so the assertion below
# then means "no pair changes side".  ORYON_TIE_PAIRS=1000 is the run DESIGN.md quotes (0.00 pt, 2 of 1000 poses differ by > 1e-4).
N_PAIRS, N_CORR, EXTENT = int(os.environ.get("ORYON_TIE_PAIRS", "300")), 500, 0.08


def _rand_rot(g):
    q = torch.randn(4, generator=g, dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float64)


def _problem(seed):
    g = torch.Generator().manual_seed(90000 + seed)
    inl = 0.3 + 0.5 * float(torch.rand(1, generator=g))
    src = (torch.rand(N_CORR, 3, generator=g) - 0.5) * 2 * EXTENT + torch.tensor([0.0, 0.0, 0.8])
    R = _rand_rot(g).float()
    t = torch.randn(3, generator=g) * 0.1
    tgt = src @ R.T + t + 0.001 * torch.randn(N_CORR, 3, generator=g)
    n_out = int(N_CORR * (1 - inl))
    idx = torch.randperm(N_CORR, generator=g)[:n_out]
    tgt[idx] = (torch.rand(n_out, 3, generator=g) - 0.5) * 2 * EXTENT + tgt.mean(0)
    if seed % 4 == 0:                                    # duplicated correspondences, as the matcher's with-replacement sampling produces
        src[-40:] = src[:40]
        tgt[-40:] = tgt[:40]
    T = torch.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return src, tgt, T


def test_add_recall_of_the_hip_path_equals_the_oracles_on_object_sized_clouds():
    from oracle import oryon_oracle as orc
    from oryon_amd import ops
    from oryon_amd.pointdsc import PointDSC
    P = orc.analytic_pointdsc_params(12, 128)
    solver = PointDSC(in_dim=6, num_layers=12, num_channels=128, num_iterations=10, ratio=0.1, sigma_d=0.1, k=40, nms_radius=0.1)
    solver.load_state_dict(P, strict=True)
    solver = solver.cuda().eval()
    probs = [_problem(i) for i in range(N_PAIRS)]
    gt = torch.stack([p[2] for p in probs])
    # HIP path, 100 pairs per call
    hip = []
    for b0 in range(0, N_PAIRS, 100):
        chunk = probs[b0:b0 + 100]
        src = torch.zeros((len(chunk), 512, 3)); tgt = torch.zeros((len(chunk), 512, 3))
        for i, (s_, t_, _) in enumerate(chunk):
            src[i, :N_CORR], tgt[i, :N_CORR] = s_, t_
        n = torch.full((len(chunk),), N_CORR, dtype=torch.int32, device="cuda")
        T, _, st = solver.register(src.cuda(), tgt.cuda(), n, torch.zeros(len(chunk), dtype=torch.int32, device="cuda"))
        assert st.tolist() == [0] * len(chunk)
        hip.append(T.cpu())
    hip = torch.cat(hip)
    # CPU oracle (the reference's forward), and how many strictly positive NMS maxima each problem has
    orc_T, n_pos = [], []
    S = int(N_CORR * CFG["ratio"])
    with torch.no_grad():
        for s_, t_, _ in probs:
            r = orc.pointdsc_forward(s_, t_, P, CFG, return_all=True)
            orc_T.append(r["final_trans"].reshape(4, 4))
            conf, sd = r["confidence"].reshape(-1), r["src_dist"].reshape(N_CORR, N_CORR)
            lm = ((conf[:, None] >= conf[None, :]) | (sd >= CFG["nms_radius"])).all(dim=1)
            n_pos.append(int((lm & (conf > 0)).sum()))
    orc_T = torch.stack(orc_T)
    n_pos = np.array(n_pos)
    assert (n_pos < S).mean() > 0.95, (n_pos < S).mean()        # the regime the caveat is about: the seed list is completed out of a tie
    # the reference's metric on both pose sets
    g = torch.Generator().manual_seed(5)
    model = (torch.rand(1000, 3, generator=g) - 0.5) * 2 * EXTENT
    diam = 2 * EXTENT * math.sqrt(3.0)
    m_h = ops.pose_metrics(hip.cuda(), gt.cuda(), model.cuda()).cpu().numpy()
    m_o = ops.pose_metrics(orc_T.cuda(), gt.cuda(), model.cuda()).cpu().numpy()
    rec = lambda m, col: float((m[:, col] < 0.1 * diam).mean()) * 100.0
    add_h, add_o, adds_h, adds_o = rec(m_h, 0), rec(m_o, 0), rec(m_h, 1), rec(m_o, 1)
    dT = (hip - orc_T).abs().amax(dim=(1, 2)).numpy()
    print(f"object-sized clouds, {N_PAIRS} problems, {int((n_pos < S).sum())} with fewer than S = {S} positive maxima (median {int(np.median(n_pos))})")
    print(f"  max |T_hip - T_oracle| {dT.max():.2e} (median {np.median(dT):.2e}); pairs above 1e-4: {int((dT > 1e-4).sum())}, above 3e-3: {int((dT > 3e-3).sum())}")
    print(f"  ADD-0.1d   recall: HIP {add_h:.2f} %  oracle {add_o:.2f} %   |diff| {abs(add_h - add_o):.2f} pt")
    print(f"  ADD-S-0.1d recall: HIP {adds_h:.2f} %  oracle {adds_o:.2f} %   |diff| {abs(adds_h - adds_o):.2f} pt")
    print(f"  mean |ADD_hip - ADD_oracle| {np.abs(m_h[:, 0] - m_o[:, 0]).mean() * 1e3:.4f} mm, max {np.abs(m_h[:, 0] - m_o[:, 0]).max() * 1e3:.4f} mm "
          f"(0.1 d = {0.1 * diam * 1e3:.1f} mm)")
    assert 20.0 < add_o < 100.0                                  # the threshold separates: neither everything nor nothing passes
    assert abs(add_h - add_o) <= 0.1 + 1e-9 and abs(adds_h - adds_o) <= 0.1 + 1e-9
    assert dT.max() < 2e-2
