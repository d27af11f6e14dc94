"""Pose-accuracy metrics and the prediction CSV of the reference's test loop, restated (CPU / numpy; off the throughput
path - SURVEY.md §8f-3):

    compute_add / compute_adds     utils/metrics.py:194-220 (+ np_transform_pcd utils/pcd.py:127-133: the reference transforms
                                   the model points in FLOAT16, which must be replicated for 0.1-point parity)
    compute_RT_distances           utils/metrics.py:222-259 (degrees, centimetres)
    mask_iou                       utils/metrics.py:18-40
    format_pred_line / read_pred_csv   pipeline.py:490-497 and scripts/evaluation/compute_metrics.py:14-47
"""
from __future__ import annotations

from typing import Dict, List

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
