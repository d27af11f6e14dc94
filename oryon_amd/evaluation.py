"""Pose-accuracy metrics and the prediction CSV of the reference's test loop, restated (CPU / numpy; off the throughput
path - SURVEY.md §8f-3):

    compute_add / compute_adds     utils/metrics.py:194-220 (+ np_transform_pcd utils/pcd.py:127-133: the reference transforms
                                   the model points in FLOAT16, which must be replicated for 0.1-point parity)
    compute_RT_distances           utils/metrics.py:222-259 (degrees, centimetres)
    mask_iou                       utils/metrics.py:18-40
    format_pred_line / read_pred_csv   pipeline.py:490-497 and scripts/evaluation/compute_metrics.py:14-47
    get_symmetry_transformations / format_sym_set   bop_toolkit_lib/misc.py:43-90, :402-411 (the symmetry set of a BOP model)
    mssd_error / mspd_error        bop_toolkit_lib/pose_error.py:370-427 (my_mssd / my_mspd) behind the float16 pose rounding of
                                   utils/evaluator.py:258-265
    Evaluator                      utils/evaluator.py:82-128 (thresholds), :206-288 (register_eval / register_test), :290-338
                                   (register_test_failure), :340-440 (means, LaTeX row, JSON): the accumulator of the test loop;
                                   per-pair errors come from the device (ops.pose_metrics, ops.pose_bop_errors) or from the numpy
                                   restatements in this file.  VSD / AR need the OpenGL renderer (out of scope).
"""
from __future__ import annotations

import json
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
from scipy.spatial import cKDTree


def transform_points_f16(pcd: np.ndarray, R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """Rigidly move model points with every operand rounded to float16 first (utils/pcd.py:127-133)."""
    return np.dot(np.asarray(pcd.astype(np.float16)), R.astype(np.float16).T) + t.astype(np.float16)


def compute_add(pcd: np.ndarray, pred_pose: np.ndarray, gt_pose: np.ndarray) -> float:
    """ADD: mean distance between corresponding model points under the two poses."""
    a = transform_points_f16(pcd, pred_pose[:3, :3], pred_pose[:3, 3])
    b = transform_points_f16(pcd, gt_pose[:3, :3], gt_pose[:3, 3])
    return np.mean(np.linalg.norm(a - b, axis=1))


def compute_adds(pcd: np.ndarray, pred_pose: np.ndarray, gt_pose: np.ndarray) -> float:
    """ADD-S: mean distance from every predicted model point to its nearest ground-truth model point."""
    a = transform_points_f16(pcd, pred_pose[:3, :3], pred_pose[:3, 3])
    b = transform_points_f16(pcd, gt_pose[:3, :3], gt_pose[:3, 3])
    d, _ = cKDTree(b.astype(np.float64)).query(a.astype(np.float64), k=1)
    return np.mean(d)


def compute_RT_distances(pose1: np.ndarray, pose2: np.ndarray):
    """Rotation angle (degrees) and translation distance (centimetres, poses in metres); batched or not."""
    if pose1 is None or pose2 is None:
        return -1
    if pose1.ndim == 2:
        pose1, pose2 = pose1[None], pose2[None]
    def unit_det(P):
        R = P[:, :3, :3]
        return R / np.cbrt(np.linalg.det(R))[:, None, None]
    R = np.matmul(unit_det(pose1), unit_det(pose2).transpose(0, 2, 1))
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1 + 1e-12, 1 - 1e-12)
    theta = np.arccos(c) * 180 / np.pi
    theta[np.isnan(theta)] = 180.0
    shift = np.linalg.norm(pose1[:, :3, 3] - pose2[:, :3, 3], axis=-1) * 100
    return theta, shift


def mask_iou(mask1: np.ndarray, mask2: np.ndarray) -> np.ndarray:
    """IoU of binary masks [B,H,W] = |and| / |or| per sample; like the reference an empty union gives NaN (0/0)
    (utils/metrics.py:18-40)."""
    a, b = mask1.reshape(mask1.shape[0], -1) != 0, mask2.reshape(mask2.shape[0], -1) != 0
    inter, union = (a & b).sum(1).astype(np.float32), (a | b).sum(1).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return inter / union


def format_pred_line(id_a: str, id_q: str, iou_a, iou_q, pred_pose: np.ndarray) -> str:
    """`id_a,id_q,<12 floats of pose[:3,:] row-major, space separated>,iou_a,iou_q` (pipeline.py:490-497)."""
    return ",".join([id_a, id_q, " ".join(str(n) for n in pred_pose[:3, :].flatten()), str(iou_a), str(iou_q)]) + "\n"


def read_pred_csv(path: str) -> List[Dict]:
    out = []
    with open(path) as fh:
        for line in fh:
            if not line.strip():
                continue
            id_a, id_q, pose_txt, iou_a, iou_q = line.strip().split(",")
            P = np.eye(4)
            P[:3, :] = np.array([float(x) for x in pose_txt.split(" ")]).reshape(3, 4)
            out.append(dict(id_a=id_a, id_q=id_q, pose=P, iou_a=float(iou_a), iou_q=float(iou_q)))
    return out


def add_accuracy(adds: np.ndarray, diameters: np.ndarray, frac: float = 0.1) -> float:
    """ADD(-S)-0.1d: share of instances whose error is below `frac` of the object diameter."""
    return float(np.mean(np.asarray(adds) < frac * np.asarray(diameters)))


# ------------------------------------------------------------------------------------------------ BOP symmetry sets, MSSD, MSPD
def rotation_matrix3(angle: float, direction) -> np.ndarray:
    """3x3 rotation by `angle` about `direction` (bop_toolkit_lib/transform.py:302-345 without the homogeneous row / point)."""
    sina, cosa = math.sin(angle), math.cos(angle)
    d = np.array(direction[:3], dtype=np.float64)
    d = d / math.sqrt(np.dot(d, d))
    R = np.diag([cosa, cosa, cosa])
    R += np.outer(d, d) * (1.0 - cosa)
    d = d * sina
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    return R


def get_symmetry_transformations(model_info: Dict, max_sym_disc_step: float = 0.05) -> List[Dict]:
    """The symmetry set of a BOP `models_info.json` entry (bop_toolkit_lib/misc.py:43-90): identity + discrete symmetries, each
    combined with the discretised continuous ones.  The reference's datasets call it with max_sym_disc_step=0.05
    (utils/data/nocs.py:139, utils/data/toyl.py:233)."""
    trans_disc = [{"R": np.eye(3), "t": np.array([[0, 0, 0]]).T}]
    for sym in model_info.get("symmetries_discrete", []):
        m = np.reshape(sym, (4, 4))
        trans_disc.append({"R": m[:3, :3], "t": m[:3, 3].reshape((3, 1))})
    trans_cont = []
    for sym in model_info.get("symmetries_continuous", []):
        axis = np.array(sym["axis"])
        offset = np.array(sym["offset"]).reshape((3, 1))
        steps = int(np.ceil(np.pi / max_sym_disc_step))
        step = 2.0 * np.pi / steps
        for i in range(steps):
            R = rotation_matrix3(i * step, axis)
            trans_cont.append({"R": R, "t": -R.dot(offset) + offset})
    trans = []
    for td in trans_disc:
        if trans_cont:
            for tc in trans_cont:
                trans.append({"R": tc["R"].dot(td["R"]), "t": tc["R"].dot(td["t"]) + tc["t"]})
        else:
            trans.append(td)
    return trans


def format_sym_set(syms: Sequence[Dict]) -> np.ndarray:
    """[N,3,4] array [R|t] of a symmetry set (bop_toolkit_lib/misc.py:402-411)."""
    return np.concatenate([np.stack([np.asarray(s_["R"]) for s_ in syms]), np.stack([np.asarray(s_["t"]) for s_ in syms])], axis=2)


def _pose_f16_mm(pose: np.ndarray):
    """utils/evaluator.py:258-262: the pose rounded to float16, translation times 1000 in float16 arithmetic."""
    p16 = np.asarray(pose).astype(np.float16)
    return p16[:3, :3], np.expand_dims(p16[:3, 3], axis=1) * 1000


def _sym_poses(R_gt, t_gt, syms):
    R = R_gt[None] @ syms[:, :3, :3]
    t = (R_gt[None] @ syms[:, :3, 3, None]) + t_gt[None]
    return R, t


REFERENCE_BOP_POINTS = 3     # bop_toolkit_lib/pose_error.py:345: np_transform slices `pts[:, :3]` on the POINT axis of its [1,N,3]
                             # input, so the reference's my_mssd / my_mspd run over the first three model points; None = all points


def mssd_error(pred_pose: np.ndarray, gt_pose: np.ndarray, pts_mm: np.ndarray, syms: np.ndarray, max_points=REFERENCE_BOP_POINTS) -> float:
    """MSSD of one pair in millimetres (my_mssd on the float16-rounded poses)."""
    pts_mm = pts_mm[:max_points] if max_points else pts_mm
    Re, te = _pose_f16_mm(pred_pose)
    Rg, tg = _pose_f16_mm(gt_pose)
    est = pts_mm @ Re.T + te.T
    Rs, ts = _sym_poses(Rg, tg, syms)
    gts = pts_mm[None] @ np.swapaxes(Rs, -1, -2) + np.swapaxes(ts, -1, -2)
    return float(np.linalg.norm(est[None] - gts, axis=2).max(axis=1).min())


def mspd_error(pred_pose: np.ndarray, gt_pose: np.ndarray, K: np.ndarray, pts_mm: np.ndarray, syms: np.ndarray,
               max_points=REFERENCE_BOP_POINTS) -> float:
    """MSPD of one pair in pixels (my_mspd on the float16-rounded poses)."""
    pts_mm = pts_mm[:max_points] if max_points else pts_mm
    Re, te = _pose_f16_mm(pred_pose)
    Rg, tg = _pose_f16_mm(gt_pose)
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)

    def project(R, t):
        x = (pts_mm[None] @ np.swapaxes(R, -1, -2) + np.swapaxes(t, -1, -2)) @ K.T
        return x[:, :, :2] / x[:, :, 2, None]
    Rs, ts = _sym_poses(Rg, tg, syms)
    return float(np.linalg.norm(project(Re[None], te[None]) - project(Rs, ts), axis=2).max(axis=1).min())


class Evaluator:
    """The reference's test-time accumulator (utils/evaluator.py) without VSD / AR: one list per metric, one entry per pair.
    `register_test` takes the per-pair ERRORS of a batch (from the device kernels or the numpy functions above) and applies the
    reference's bookkeeping: zero-pose rule, failed-pose count, ADD(S)-0.1d against the ADD diameter, MSSD / MSPD recall means,
    rotation / translation recalls; `register_test_failure` is the automatic failure of an invalid detection or a matcher that
    returned nothing (pipeline.py:335-350: every score 0)."""

    def __init__(self, exp_tag: str = "", compute_iou: bool = True):
        self.exp_tag = exp_tag
        self.compute_iou = compute_iou
        self.mssd_rec = np.arange(0.05, 0.51, 0.05)
        self.mspd_rec = np.arange(5, 51, 5)
        self.pose_recall_th = [(5, 10), (10, 20), (15, 30)]
        self.metrics: Dict[str, list] = {}
        self.counts: Dict[str, list] = {}
        self.init_test()

    def init_test(self) -> None:
        self.metrics, self.counts = {}, {}
        if self.compute_iou:
            for k in ("Anchor IoU", "Query IoU", "Mean IoU", "IoU > .25", "IoU > .5", "IoU > .75"):
                self.metrics[k] = []
        for k in ("R error", "T error", "ADD(S)-0.1d", "MSSD", "MSPD"):
            self.metrics[k] = []
        for k in ("Missing segm", "Failed pose", "Zero pose"):
            self.counts[k] = []
        for r_th, t_th in self.pose_recall_th:
            self.metrics[f"Recall ({r_th}deg, {t_th}cm)"] = []
        self.metrics["instance_id"], self.metrics["cls_id"] = [], []

    @staticmethod
    def effective_pose(pred_pose: np.ndarray, pred_pose_rel: np.ndarray) -> np.ndarray:
        """utils/evaluator.py:229-231: a relative pose with at most one non-zero entry scores as the identity."""
        return np.eye(4, dtype=pred_pose.dtype) if np.count_nonzero(pred_pose_rel) <= 1 else pred_pose

    def register_test(self, *, pred_pose_rel: np.ndarray, rot_deg: float, trans_cm: float, add_s: float, add_diam: float, mssd_mm: float,
                      mspd_px: float, bop_diam_mm: float, cls_id, instance_id, iou_a: Optional[float] = None,
                      iou_q: Optional[float] = None) -> None:
        """One pair that went through the registration.  The errors must have been computed on `effective_pose`."""
        if self.compute_iou:
            mean = (iou_a + iou_q) / 2.0
            self.metrics["Anchor IoU"].append(float(iou_a)); self.metrics["Query IoU"].append(float(iou_q))
            self.metrics["Mean IoU"].append(float(mean))
            for k, th in (("IoU > .25", 0.25), ("IoU > .5", 0.5), ("IoU > .75", 0.75)):
                self.metrics[k].append(int(mean > th))
        self.counts["Missing segm"].append(0)
        self.counts["Failed pose"].append(int((np.asarray(pred_pose_rel) == np.eye(4)).all()))
        self.counts["Zero pose"].append(int(np.count_nonzero(pred_pose_rel) <= 1))
        self.metrics["R error"].append(float(rot_deg))
        self.metrics["T error"].append(float(trans_cm))
        for r_th, t_th in self.pose_recall_th:
            self.metrics[f"Recall ({r_th}deg, {t_th}cm)"].append(float(rot_deg <= r_th and trans_cm <= t_th))
        self.metrics["ADD(S)-0.1d"].append(float(add_s <= add_diam * 0.1))
        self.metrics["MSSD"].append(float((mssd_mm < self.mssd_rec * bop_diam_mm).mean()))
        self.metrics["MSPD"].append(float((mspd_px < self.mspd_rec).mean()))
        self.metrics["cls_id"].append(cls_id)
        self.metrics["instance_id"].append(instance_id)

    def register_test_failure(self, *, cls_id, instance_id, iou_a: Optional[float] = None, iou_q: Optional[float] = None) -> None:
        for k in ("R error", "T error", "ADD(S)-0.1d", "MSSD", "MSPD"):
            self.metrics[k].append(0.0)
        if self.compute_iou:
            self.metrics["Anchor IoU"].append(float(iou_a)); self.metrics["Query IoU"].append(float(iou_q))
            for k in ("Mean IoU", "IoU > .25", "IoU > .5", "IoU > .75"):
                self.metrics[k].append(0.0)
        self.counts["Missing segm"].append(1)
        self.counts["Failed pose"].append(0)
        self.counts["Zero pose"].append(0)
        for r_th, t_th in self.pose_recall_th:
            self.metrics[f"Recall ({r_th}deg, {t_th}cm)"].append(0)
        self.metrics["cls_id"].append(cls_id)
        self.metrics["instance_id"].append(instance_id)

    def get_means(self, cls_id=None) -> Dict[str, float]:
        sel = None if cls_id is None else np.asarray(self.metrics["cls_id"]) == cls_id
        out = {}
        for name, value in self.metrics.items():
            if name not in ("cls_id", "instance_id") and len(value) > 0:
                v = np.asarray(value)
                out[name] = float((v if sel is None else v[sel]).mean())
        return out

    def get_latex_str(self, cls_id=None) -> str:
        """The reference's table row with VSD / AR left out (its own compute_vsd=False format, utils/evaluator.py:431-440)."""
        m = self.get_means(cls_id)
        tag = self.exp_tag if cls_id is None else cls_id
        s_ = f"{tag} & - & - & {m['MSSD'] * 100:.1f} & {m['MSPD'] * 100:.1f} & {m['ADD(S)-0.1d'] * 100:.1f} &"
        s_ += f" {m['Mean IoU'] * 100:.1f} \\\\" if self.compute_iou else " - \\\\"
        return s_ + (" \n" if cls_id is None else "")

    def test_summary(self) -> List[str]:
        return [self.get_latex_str(c) for c in np.unique(self.metrics["cls_id"]).tolist()]

    def save(self, fh) -> None:
        d = dict(self.metrics)
        d.update(self.counts)
        json.dump(d, fh)


def evaluate_batch(evaluator: Evaluator, *, pred_pose_rel: np.ndarray, anchor_pose: np.ndarray, gt_pose: np.ndarray, K: np.ndarray,
                   status: Sequence[int], cls_ids: Sequence, instance_ids: Sequence[str], objects: Dict, iou_a=None, iou_q=None,
                   device: Optional[str] = None) -> None:
    """What the per-sample loop of FPM_Pipeline.test_step registers for a batch (pipeline.py:313-350): pairs whose status is not
    PAIR_OK are automatic failures (`register_test_failure`), the others are scored on pred_q = pred_pose_rel @ anchor_pose (fp32,
    pipeline.py:320) after the zero-pose rule.  objects[cls_id] = {'pts' [N,3] mm (float64), 'diameter' (BOP, mm), 'syms' [S,3,4]}.
    device: a torch device string -> per-pair errors from the HIP kernels (oryon_pose_metrics, oryon_pose_bop_errors); None -> the
    numpy restatements in this file."""
    n = len(status)
    rel = np.asarray(pred_pose_rel, dtype=np.float32)
    pred_q = np.matmul(rel, np.asarray(anchor_pose, dtype=np.float32))
    for i in range(n):
        pred_q[i] = Evaluator.effective_pose(pred_q[i], rel[i])
    ok = [i for i in range(n) if int(status[i]) == 0]
    errs = {}
    if ok and device is not None:
        import torch
        from . import ops
        keys = list(dict.fromkeys(cls_ids[i] for i in ok))
        pts_mm = [np.asarray(objects[k]["pts"], dtype=np.float64) for k in keys]
        syms = [np.asarray(objects[k]["syms"], dtype=np.float64) for k in keys]
        po = torch.tensor(np.concatenate(([0], np.cumsum([p_.shape[0] for p_ in pts_mm]))), dtype=torch.int32)
        so = torch.tensor(np.concatenate(([0], np.cumsum([s_.shape[0] for s_ in syms]))), dtype=torch.int32)
        which = torch.tensor([keys.index(cls_ids[i]) for i in ok], dtype=torch.int32)
        pq, gq = torch.from_numpy(pred_q[ok]), torch.from_numpy(np.asarray(gt_pose)[ok])
        met = ops.pose_metrics(pq.to(device), gq.to(device, torch.float32), torch.from_numpy(np.concatenate(pts_mm) / 1000.0).float().to(device),
                               po, which).cpu().numpy()
        bop = ops.pose_bop_errors(pq.to(device), gq.to(device), torch.from_numpy(np.asarray(K, dtype=np.float64)[ok]).to(device),
                                  torch.from_numpy(np.concatenate(pts_mm)).to(device), po, torch.from_numpy(np.concatenate(syms)).to(device),
                                  so, which).cpu().numpy()
        for j, i in enumerate(ok):
            sym = objects[cls_ids[i]]["syms"].shape[0] > 1
            errs[i] = (float(met[j, 2]), float(met[j, 3]), float(met[j, 1] if sym else met[j, 0]), float(bop[j, 0]), float(bop[j, 1]))
    elif ok:
        for i in ok:
            o = objects[cls_ids[i]]
            pts_m = np.asarray(o["pts"]) / 1000.0
            th, sh = compute_RT_distances(pred_q[i], np.asarray(gt_pose[i]))
            add = compute_adds(pts_m, pred_q[i], gt_pose[i]) if o["syms"].shape[0] > 1 else compute_add(pts_m, pred_q[i], gt_pose[i])
            errs[i] = (float(th[0]), float(sh[0]), float(add), mssd_error(pred_q[i], gt_pose[i], o["pts"], o["syms"]),
                       mspd_error(pred_q[i], gt_pose[i], K[i], o["pts"], o["syms"]))
    for i in range(n):
        ia = None if iou_a is None else float(iou_a[i])
        iq = None if iou_q is None else float(iou_q[i])
        if i not in errs:
            evaluator.register_test_failure(cls_id=cls_ids[i], instance_id=instance_ids[i], iou_a=ia, iou_q=iq)
            continue
        o = objects[cls_ids[i]]
        rot, tr, add, ms, mp = errs[i]
        evaluator.register_test(pred_pose_rel=rel[i], rot_deg=rot, trans_cm=tr, add_s=add, add_diam=extent_diameter(o["pts"]) / 1000.0,
                                mssd_mm=ms, mspd_px=mp, bop_diam_mm=float(o["diameter"]), cls_id=cls_ids[i], instance_id=instance_ids[i],
                                iou_a=ia, iou_q=iq)


def extent_diameter(pts: np.ndarray) -> float:
    """The "ADD diameter" of the reference: the largest side of the model's axis-aligned bounding box (utils/pcd.py:16-20), not the
    BOP diameter; same unit as pts."""
    xyz = np.asarray(pts)[:, :3]
    return float(np.max(xyz.max(axis=0) - xyz.min(axis=0)))
