"""Pose-accuracy evaluation on the device (oryon_pose_metrics, oryon_pose_bop_errors) against what the reference's own
Evaluator / my_mssd / my_mspd produced for the same batch (tests/golden/g9_bop_metrics.npz, written by tools/gen_goldens.py from
utils/evaluator.py + bop_toolkit_lib/pose_error.py)."""
import os

import numpy as np
import pytest
import torch

from oryon_amd import evaluation as ev
from tests.test_evaluation import _check_against_reference_evaluator, _g9_objects

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _concat(objs, names):
    pts = [objs[k]["pts"] for k in names]
    syms = [objs[k]["syms"] for k in names]
    po = torch.tensor(np.concatenate(([0], np.cumsum([p.shape[0] for p in pts]))), dtype=torch.int32)
    so = torch.tensor(np.concatenate(([0], np.cumsum([s.shape[0] for s in syms]))), dtype=torch.int32)
    return torch.from_numpy(np.concatenate(pts)), po, torch.from_numpy(np.concatenate(syms)), so


def test_mssd_mspd_kernel_matches_reference():
    from oryon_amd import ops
    g = np.load(os.path.join(GOLD, "g9_bop_metrics.npz"))
    objs, _ = _g9_objects(g)
    names = g["model_names"].tolist()
    cls = g["cls"].tolist()
    keep = [i for i in range(len(cls)) if i not in g["failures"].tolist()]
    pq = np.stack([ev.Evaluator.effective_pose(g["rel"][i] @ g["anchor"][i], g["rel"][i]) for i in keep])
    pts, po, syms, so = _concat(objs, names)
    which = torch.tensor([names.index(cls[i]) for i in keep], dtype=torch.int32)
    K = torch.from_numpy(np.tile(g["K"], (len(keep), 1, 1)))
    out = ops.pose_bop_errors(torch.from_numpy(pq).cuda(), torch.from_numpy(g["gt"][keep]).cuda(), K.cuda(), pts.cuda(), po, syms.cuda(), so,
                              which).cpu().numpy()
    np.testing.assert_allclose(out[:, 0], g["mssd_raw"][keep], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out[:, 1], g["mspd_raw"][keep], rtol=1e-9, atol=1e-9)
    # every model point (the BOP definition) instead of the reference's first three: never smaller, and equal to the numpy restatement
    full = ops.pose_bop_errors(torch.from_numpy(pq).cuda(), torch.from_numpy(g["gt"][keep]).cuda(), K.cuda(), pts.cuda(), po, syms.cuda(), so,
                               which, max_points=0).cpu().numpy()
    assert (full[:, 0] >= out[:, 0] - 1e-9).all()
    for j, i in enumerate(keep):
        o = objs[cls[i]]
        assert abs(full[j, 0] - ev.mssd_error(pq[j], g["gt"][i], o["pts"], o["syms"], max_points=None)) <= 1e-9 * max(1.0, full[j, 0])
        assert abs(full[j, 1] - ev.mspd_error(pq[j], g["gt"][i], g["K"], o["pts"], o["syms"], max_points=None)) <= 1e-9 * max(1.0, full[j, 1])


def test_device_evaluation_matches_reference_evaluator():
    """evaluate_batch(device='cuda'): ADD(-S) / R / t from oryon_pose_metrics, MSSD / MSPD from oryon_pose_bop_errors, the reference's
    bookkeeping on top - every registered list, the means and the LaTeX row equal the reference Evaluator's."""
    g = np.load(os.path.join(GOLD, "g9_bop_metrics.npz"))
    objs, _ = _g9_objects(g)
    n = len(g["cls"])
    status = [2 if i in g["failures"].tolist() else 0 for i in range(n)]
    E = ev.Evaluator("g9")
    ev.evaluate_batch(E, pred_pose_rel=g["rel"], anchor_pose=g["anchor"], gt_pose=g["gt"], K=np.tile(g["K"], (n, 1, 1)), status=status,
                      cls_ids=g["cls"].tolist(), instance_ids=[f"inst{i}" for i in range(n)], objects=objs,
                      iou_a=[np.float32(0.5 + 0.04 * i) for i in range(n)], iou_q=[np.float32(0.9 - 0.05 * i) for i in range(n)], device="cuda")
    _check_against_reference_evaluator(E, g, rt_atol=2e-2)


def test_round_to_half_boundaries():
    """The float64 -> float16 pose rounding inside the kernel is a single round-to-nearest-even (numpy's astype), also where a detour
    through float32 would round twice: translations just above / below a half-way point between two float16 values."""
    from oryon_amd import ops
    h = np.float16(0.9375)
    ulp = float(np.spacing(h))
    vals = [float(h) + ulp / 2, float(h) + ulp / 2 + 1e-9, float(h) + ulp / 2 - 1e-9, float(h) + ulp / 2 + 3e-8, -(float(h) + ulp / 2 + 1e-9)]
    pred = np.tile(np.eye(4), (len(vals), 1, 1))
    gt = np.tile(np.eye(4), (len(vals), 1, 1))
    for i, v in enumerate(vals):
        pred[i, 2, 3] = v
        gt[i, 2, 3] = 1.0 if v > 0 else -1.0
    pts = torch.tensor([[0.0, 0.0, 0.0]], dtype=torch.float64)
    syms = torch.tensor(np.eye(4)[None, :3, :], dtype=torch.float64)
    K = torch.eye(3, dtype=torch.float64).repeat(len(vals), 1, 1)
    out = ops.pose_bop_errors(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), K.cuda(), pts.cuda(), torch.tensor([0, 1], dtype=torch.int32),
                              syms.cuda(), torch.tensor([0, 1], dtype=torch.int32)).cpu().numpy()
    for i, v in enumerate(vals):
        want = abs(float(np.float16(v) * 1000) - float(np.float16(gt[i, 2, 3]) * 1000))
        assert out[i, 0] == want, (v, out[i, 0], want)
