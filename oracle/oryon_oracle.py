"""CPU oracle for the Oryon hot path (TEST INFRASTRUCTURE — never imported by the product).

A from-scratch restatement (torch-CPU / numpy, fp32) of the reference algorithm for
    masks -> ROI -> cosine nearest neighbour -> sample -> scale/validate -> lift -> PointDSC -> pose
Each function cites the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; `oryon_amd` must never do so.

Parity pin: every function here is checked against outputs of the *real* reference (imported from
/root/reference in the build container by `tools/gen_goldens.py`) stored in `tests/golden/*.npz`
(`tests/test_oracle_goldens.py`).

Two matcher forms are provided:
  * `cosine_nn_broadcast`  - the reference's own evaluation order ([N1,N2,C] broadcast through
                             torch.cosine_similarity); this is what `cpu_baseline` times.
  * `cosine_nn_gemm`       - same result via normalise + mm (fast; indices identical except at
                             sub-1e-6 near-ties).
The bit-exact integer oracle for the HIP matcher is the C restatement in `oryon_oracle.c`
(k-ordered fmaf chain == what v_mfma_f32_32x32x2_f32 computes).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS_COS = 1e-8  # torch.cosine_similarity default eps (utils/pcd.py:28 uses the default)


# ----------------------------------------------------------------------------------------------
# masks / ROI                                                       utils/pcd.py:184-185
# ----------------------------------------------------------------------------------------------
def roi_from_mask(mask: torch.Tensor) -> torch.Tensor:
    """Row-major (y, x) coordinates of pixels equal to 1  (utils/pcd.py:184-185)."""
    return torch.nonzero(mask == 1)


def resize_mask_nearest(mask: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """GT mask -> featmap grid with legacy 'nearest' (src = floor(dst * in/out)); pipeline.py:408-411."""
    m = mask.clone().to(torch.float)[None, None]
    return F.interpolate(m, tuple(out_hw), mode="nearest").squeeze().to(torch.int)


def predicted_mask(logits: torch.Tensor, threshold: float = 0.5) -> torch.Tensor:
    """sigmoid(logit) > th -> {0,1}  (losses.py:58-59)."""
    return torch.where(torch.sigmoid(logits) > threshold, 1, 0)


def sample_select(n_items: int, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Exactly n indices from range(n_items); replacement only if n > n_items (utils/misc.py:242-254)."""
    w = torch.ones(n_items, dtype=torch.float64)
    return torch.multinomial(w, n, replacement=(n > n_items), generator=generator)


# ----------------------------------------------------------------------------------------------
# matcher                                                           utils/pcd.py:22-33, 177-216
# ----------------------------------------------------------------------------------------------
def gather_roi_feats(feats: torch.Tensor, roi: torch.Tensor) -> torch.Tensor:
    """[C,H,W] channel-planar map -> [N,C] rows at roi (y,x)  (utils/pcd.py:192-193)."""
    return feats[:, roi[:, 0], roi[:, 1]].T.contiguous()


def cosine_nn_broadcast(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference evaluation order: dist = 0.5*(1 - cos(a_i, b_j)) over a [N1,N2,C] broadcast,
    then row amin / argmin (utils/pcd.py:28-29, 202-204)."""
    dist = 0.5 * (-1 * F.cosine_similarity(a.unsqueeze(1), b.unsqueeze(0), dim=2) + 1)
    return torch.amin(dist, dim=1), torch.argmin(dist, dim=1)


def cosine_nn_gemm(a: torch.Tensor, b: torch.Tensor, chunk: int = 1024) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same quantity through x/max(|x|,eps) and a matrix product; processes `chunk` anchor rows at a time."""
    an = a / torch.linalg.vector_norm(a, dim=1, keepdim=True).clamp_min(EPS_COS)
    bn = b / torch.linalg.vector_norm(b, dim=1, keepdim=True).clamp_min(EPS_COS)
    mins, args = [], []
    for s in range(0, an.shape[0], chunk):
        d = 0.5 * (1.0 - an[s:s + chunk] @ bn.T)
        mins.append(torch.amin(d, dim=1))
        args.append(torch.argmin(d, dim=1))
    if not mins:
        return a.new_zeros(0), torch.zeros(0, dtype=torch.long)
    return torch.cat(mins), torch.cat(args)


def match_presample(feats1, feats2, mask1, mask2, threshold: float, roi1_override=None, form: str = "gemm"):
    """Deterministic part of nn_correspondences (everything before the second multinomial draw).

    Returns dict(roi1, roi2, min_dist, argmin, valid) ; utils/pcd.py:184-205.
    `roi1_override` lets a caller inject an already-subsampled anchor ROI (the first RNG draw)."""
    roi1 = roi_from_mask(mask1) if roi1_override is None else roi1_override
    roi2 = roi_from_mask(mask2)
    f1 = gather_roi_feats(feats1, roi1).to(torch.float32)
    f2 = gather_roi_feats(feats2, roi2).to(torch.float32)
    if roi1.shape[0] == 0 or roi2.shape[0] == 0:
        z = torch.zeros(roi1.shape[0])
        return dict(roi1=roi1, roi2=roi2, min_dist=z, argmin=torch.zeros(roi1.shape[0], dtype=torch.long),
                    valid=torch.zeros(roi1.shape[0], dtype=torch.bool))
    fn = cosine_nn_gemm if form == "gemm" else cosine_nn_broadcast
    min_dist, arg = fn(f1, f2)
    return dict(roi1=roi1, roi2=roi2, min_dist=min_dist, argmin=arg, valid=min_dist < threshold)


def nn_correspondences(feats1, feats2, mask1, mask2, threshold: float, max_corrs: int,
                       subsample_source: Optional[int], form: str = "gemm"):
    """Whole matcher incl. the two global-RNG draws, same call order as utils/pcd.py:177-216.
    Returns int64 [max_corrs,4] (y1,x1,y2,x2) or None."""
    roi1 = roi_from_mask(mask1)
    if subsample_source is not None and roi1.shape[0] > subsample_source:
        roi1 = roi1[sample_select(roi1.shape[0], subsample_source)]
    pre = match_presample(feats1, feats2, mask1, mask2, threshold, roi1_override=roi1, form=form)
    keep = torch.nonzero(pre["valid"]).squeeze(1)
    if keep.shape[0] > 1:
        pairs = torch.cat((pre["roi1"][keep], pre["roi2"][pre["argmin"]][keep]), dim=1)
        return pairs[sample_select(pairs.shape[0], max_corrs)]
    return None


# ----------------------------------------------------------------------------------------------
# coordinates + lift                         utils/coordinates.py:5-48, pipeline.py:443-460, utils/pcd.py:35-74
# ----------------------------------------------------------------------------------------------
def scale_validate_truncate(corrs: torch.Tensor, feat_hw, size_a, size_q):
    """(y,x) featmap coords -> original-image integer pixels, dropping rows outside either image.

    fp32 op order is the reference's: scale = fp32(target)/fp32(source); y' = fp32(y)*scale;
    valid = 0 <= y' < H and 0 <= x' < W on both sides; trunc to int64 (pipeline.py:447-456)."""
    def one(c, hw):
        c = c.clone().to(torch.float32)
        sy = torch.tensor(float(hw[0]), dtype=torch.float32) / torch.tensor(float(feat_hw[0]), dtype=torch.float32)
        sx = torch.tensor(float(hw[1]), dtype=torch.float32) / torch.tensor(float(feat_hw[1]), dtype=torch.float32)
        c[:, 0] = c[:, 0] * sy
        c[:, 1] = c[:, 1] * sx
        ok = (c[:, 0] >= 0) & (c[:, 0] < float(hw[0])) & (c[:, 1] >= 0) & (c[:, 1] < float(hw[1]))
        return c, ok
    ca, oka = one(corrs[:, :2], size_a)
    cq, okq = one(corrs[:, 2:], size_q)
    ok = oka & okq
    return ca[ok].to(torch.long), cq[ok].to(torch.long), ok


def lift_points(depth: torch.Tensor, cam9: torch.Tensor, x_idx: torch.Tensor, y_idx: torch.Tensor) -> torch.Tensor:
    """Pin-hole lift of selected pixels, millimetres in -> millimetres out, fp32 (utils/pcd.py:44-74).
    X = (x - cx) * z / fx ; Y = (y - cy) * z / fy ; Z = z  (left-to-right, no fma)."""
    z = depth[y_idx, x_idx].to(torch.float32)
    fx, cx, fy, cy = (cam9[i].to(torch.float32) for i in (0, 2, 4, 5))
    xf, yf = x_idx.to(torch.float32), y_idx.to(torch.float32)
    return torch.stack(((xf - cx) * z / fx, (yf - cy) * z / fy, z), dim=1)


def lift_pair(depth_a, depth_q, cam_a9, cam_q9, corrs, feat_hw, size_a, size_q):
    """pipeline.py:443-460 assembled: returns (pcd_a, pcd_q) in metres and the row-validity mask."""
    ca, cq, ok = scale_validate_truncate(corrs, feat_hw, size_a, size_q)
    pa = lift_points(depth_a, cam_a9, ca[:, 1], ca[:, 0]) / 1000.0
    pq = lift_points(depth_q, cam_q9, cq[:, 1], cq[:, 0]) / 1000.0
    return pa, pq, ok


# ----------------------------------------------------------------------------------------------
# weighted Kabsch                                                   models/pointdsc/common.py:7-45
# ----------------------------------------------------------------------------------------------
def kabsch(A: torch.Tensor, B: torch.Tensor, w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched weighted rigid fit B ~ R A + t -> [bs,4,4].  Negative weights are clipped to 0,
    centroids use sum(w)+1e-6, R = V diag(1,1,det(V U^T)) U^T with U,S,V = svd(A_c^T W B_c)."""
    bs = A.shape[0]
    if w is None:
        w = torch.ones_like(A[:, :, 0])
    w = torch.where(w < 0, torch.zeros_like(w), w)
    den = w.sum(dim=1, keepdim=True)[:, :, None] + 1e-6
    ca = (A * w[:, :, None]).sum(dim=1, keepdim=True) / den
    cb = (B * w[:, :, None]).sum(dim=1, keepdim=True) / den
    H = (A - ca).transpose(1, 2) @ (w[:, :, None] * (B - cb))
    dev = A.device                                             # the reference takes H to the host for the SVD (common.py:33)
    U, _, Vh = torch.linalg.svd(H.cpu())
    U, Vh = U.to(dev), Vh.to(dev)
    V = Vh.transpose(1, 2)
    d = torch.det(V @ U.transpose(1, 2))
    D = torch.eye(3, device=dev).repeat(bs, 1, 1)
    D[:, 2, 2] = d
    R = V @ D @ U.transpose(1, 2)
    t = cb.transpose(1, 2) - R @ ca.transpose(1, 2)
    T = torch.eye(4, device=dev).repeat(bs, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3:4] = t
    return T


# ----------------------------------------------------------------------------------------------
# PointDSC                                                          models/pointdsc/PointDSC.py
# ----------------------------------------------------------------------------------------------
def pairwise_norm(p: torch.Tensor) -> torch.Tensor:
    """[n,3] -> [n,n] Euclidean distances computed from coordinate differences (PointDSC.py:151)."""
    return torch.norm(p[:, None, :] - p[None, :, :], dim=-1)


def sc_matrix(src: torch.Tensor, tgt: torch.Tensor, sigma_d: float):
    """Spatial-consistency matrix clamp(1 - (|si-sj| - |ti-tj|)^2 / sigma_d^2, 0) and src distances
    (PointDSC.py:150-153)."""
    sd = pairwise_norm(src)
    diff = sd - pairwise_norm(tgt)
    sig = torch.tensor(sigma_d, dtype=torch.float32)
    return torch.clamp(1.0 - diff ** 2 / sig ** 2, min=0), sd


def _conv_bn(x, W, b, bn=None, relu=False):
    """1x1 Conv1d (+ eval-mode BatchNorm1d) (+ ReLU) on row-major features x[n,Cin] -> [n,Cout]."""
    y = x @ W.T + b
    if bn is not None:
        g, beta, mean, var = bn
        y = (y - mean) / torch.sqrt(var + 1e-5) * g + beta
    return torch.relu(y) if relu else y


def encoder_forward(corr_pos: torch.Tensor, SC: torch.Tensor, P: Dict[str, torch.Tensor], num_layers: int,
                    return_all: bool = False):
    """NonLocalNet on row-major features [n,C] (PointDSC.py:27-45, 65-77).
    P uses the reference state-dict names (prefix 'encoder.')."""
    def g(name):
        return P["encoder." + name]

    def bn(prefix):
        return (g(prefix + ".weight"), g(prefix + ".bias"), g(prefix + ".running_mean"), g(prefix + ".running_var"))

    feat = _conv_bn(corr_pos, g("layer0.weight")[:, :, 0], g("layer0.bias"))
    trace = [feat]
    for i in range(num_layers):
        pc = f"blocks.PointCN_layer_{i}"
        nl = f"blocks.NonLocal_layer_{i}"
        feat = _conv_bn(feat, g(pc + ".0.weight")[:, :, 0], g(pc + ".0.bias"), bn(pc + ".1"), relu=True)
        C = feat.shape[1]
        q = _conv_bn(feat, g(nl + ".projection_q.weight")[:, :, 0], g(nl + ".projection_q.bias"))
        k = _conv_bn(feat, g(nl + ".projection_k.weight")[:, :, 0], g(nl + ".projection_k.bias"))
        v = _conv_bn(feat, g(nl + ".projection_v.weight")[:, :, 0], g(nl + ".projection_v.bias"))
        logits = SC * ((q @ k.T) / (C ** 0.5))
        wgt = torch.softmax(logits, dim=-1)
        msg = wgt @ v
        m = _conv_bn(msg, g(nl + ".fc_message.0.weight")[:, :, 0], g(nl + ".fc_message.0.bias"), bn(nl + ".fc_message.1"), relu=True)
        m = _conv_bn(m, g(nl + ".fc_message.3.weight")[:, :, 0], g(nl + ".fc_message.3.bias"), bn(nl + ".fc_message.4"), relu=True)
        m = _conv_bn(m, g(nl + ".fc_message.6.weight")[:, :, 0], g(nl + ".fc_message.6.bias"))
        feat = feat + m
        trace.append(feat)
    return (feat, trace) if return_all else feat


def confidence_head(feat: torch.Tensor, P: Dict[str, torch.Tensor]) -> torch.Tensor:
    """classification MLP C->32->32->1 (PointDSC.py:107-113,171)."""
    h = _conv_bn(feat, P["classification.0.weight"][:, :, 0], P["classification.0.bias"], relu=True)
    h = _conv_bn(h, P["classification.2.weight"][:, :, 0], P["classification.2.bias"], relu=True)
    return _conv_bn(h, P["classification.4.weight"][:, :, 0], P["classification.4.bias"])[:, 0]


def nms_local_max(src_dist: torch.Tensor, score: torch.Tensor, radius: float) -> torch.Tensor:
    """is_local_max_i = AND_j (score_i >= score_j  OR  d_ij >= R)   (PointDSC.py:212-216)."""
    rel = (score[:, None] >= score[None, :]) | (src_dist >= radius)
    return rel.all(dim=1)


def stable_desc_order(v: torch.Tensor) -> torch.Tensor:
    """Descending order with ties in ascending index order (the deterministic stand-in for
    torch.argsort(descending=True), whose tie order is implementation-defined)."""
    return torch.argsort(v, descending=True, stable=True)


def pick_seeds(src_dist, score, radius: float, max_num: int) -> torch.Tensor:
    """NMS seeds (PointDSC.py:199-217) with the stable tie order."""
    keyed = score * nms_local_max(src_dist, score, radius).float()
    return stable_desc_order(keyed)[:max_num]


def knn_rows(feat_n: torch.Tensor, rows: torch.Tensor, k: int) -> torch.Tensor:
    """k nearest neighbours (feature space, self dropped as 'rank 0') for the given rows only.
    dist = 2 - 2 f f^T ; ascending, ties by index  (common.py:48-69 ; PointDSC.py:250-252)."""
    d = 2.0 - 2.0 * (feat_n[rows] @ feat_n.T)
    order = torch.argsort(d, dim=1, descending=False, stable=True)
    return order[:, 1:k + 1]


def power_iteration(M: torch.Tensor, num_iterations: int) -> torch.Tensor:
    """Leading eigenvector of every [k,k] block with the joint allclose early exit
    (PointDSC.py:338-358).  M: [S,k,k] -> [S,k]."""
    v = torch.ones_like(M[:, :, 0:1])
    last = v
    for _ in range(num_iterations):
        v = torch.bmm(M, v)
        v = v / (torch.norm(v, dim=1, keepdim=True) + 1e-6)
        if torch.allclose(v, last):
            break
        last = v
    return v.squeeze(-1)


def seed_hypotheses(seeds, feat_n, src, tgt, sigma: float, sigma_d: float, k: int, num_iterations: int,
                    inlier_threshold: float):
    """cal_seed_trans (PointDSC.py:234-336): per-seed kNN consensus set -> weights -> Kabsch -> fitness."""
    n = feat_n.shape[0]
    k = min(k, n - 1)
    idx = knn_rows(feat_n, seeds, k)                           # [S,k]
    fk = feat_n[idx]                                           # [S,k,C]
    sig = torch.tensor(sigma, dtype=torch.float32)
    sigd = torch.tensor(sigma_d, dtype=torch.float32)
    fM = torch.clamp(1 - (1 - fk @ fk.transpose(1, 2)) / sig ** 2, min=0)
    sk, tk = src[idx], tgt[idx]                                # [S,k,3]
    dS = ((sk[:, :, None, :] - sk[:, None, :, :]) ** 2).sum(-1) ** 0.5
    dT = ((tk[:, :, None, :] - tk[:, None, :, :]) ** 2).sum(-1) ** 0.5
    sM = torch.clamp(1 - (dS - dT) ** 2 / sigd ** 2, min=0)
    M = fM * sM
    ar = torch.arange(k)
    M[:, ar, ar] = 0
    w = power_iteration(M, num_iterations)
    w = w / (w.sum(dim=-1, keepdim=True) + 1e-6)
    T = kabsch(sk, tk, w)                                      # [S,4,4]
    pred = torch.einsum("snm,km->snk", T[:, :3, :3], src).transpose(1, 2) + T[:, None, :3, 3]
    L2 = torch.norm(pred - tgt[None], dim=-1)                  # [S,n]
    fitness = (L2 < inlier_threshold).float().mean(dim=-1)
    best = int(torch.argmax(fitness))
    labels = (L2[best] < inlier_threshold).float()
    return dict(knn_idx=idx, weights=w, seed_trans=T, fitness=fitness, best=best, trans=T[best], labels=labels, M=M)


def post_refinement(T: torch.Tensor, src: torch.Tensor, tgt: torch.Tensor, inlier_threshold: float = 0.10,
                    max_iter: int = 20) -> torch.Tensor:
    """Iterative re-fit on inliers with weights 1/(1+(d/tau)^2); stops when the inlier count repeats
    (PointDSC.py:403-438; the reference uses tau=0.10 for 3DMatch-style thresholds, 1.2 otherwise)."""
    tau = 0.10 if inlier_threshold == 0.10 else 1.2
    prev = 0
    for _ in range(max_iter):
        warped = (T[:3, :3] @ src.T + T[:3, 3:4]).T
        L2 = torch.norm(warped - tgt, dim=-1)
        inl = L2 < tau
        cnt = int(inl.sum())
        if abs(cnt - prev) < 1:
            break
        prev = cnt
        w = 1 / (1 + (L2 / tau) ** 2)
        T = kabsch(src[inl][None], tgt[inl][None], w[inl][None])[0]
    return T


def pointdsc_forward(src: torch.Tensor, tgt: torch.Tensor, P: Dict[str, torch.Tensor], cfg: dict,
                     return_all: bool = False):
    """get_pointdsc_pose + PointDSC.forward(testing) for one pair (utils/pointdsc/init.py:10-29;
    PointDSC.py:128-197).  src/tgt: [n,3] metres.  cfg keys: num_layers, num_iterations, ratio,
    sigma_d, k, nms_radius, inlier_threshold."""
    src = src.float()
    tgt = tgt.float()
    corr_pos = torch.cat([src, tgt], dim=-1)
    corr_pos = corr_pos - corr_pos.mean(0)
    SC, sd = sc_matrix(src, tgt, cfg["sigma_d"])
    feat = encoder_forward(corr_pos, SC, P, cfg["num_layers"])
    feat_n = F.normalize(feat, p=2, dim=-1)
    conf = confidence_head(feat, P)
    n = src.shape[0]
    seeds = pick_seeds(sd, conf, cfg["nms_radius"], int(n * cfg["ratio"]))
    hyp = seed_hypotheses(seeds, feat_n, src, tgt, float(P["sigma"][0]), float(P["sigma_spat"][0]),
                          cfg["k"], cfg["num_iterations"], cfg["inlier_threshold"])
    T = post_refinement(hyp["trans"], src, tgt, cfg["inlier_threshold"])
    if return_all:
        return dict(SC=SC, src_dist=sd, feat=feat, feat_n=feat_n, confidence=conf, seeds=seeds, **hyp, final_trans=T)
    return T


# ----------------------------------------------------------------------------------------------
# analytic PointDSC parameters (shared by the golden generator and the tests; no blobs committed)
# ----------------------------------------------------------------------------------------------
def pointdsc_param_shapes(num_layers: int, C: int, in_dim: int = 6):
    """(name, shape) list in reference state-dict order (PointDSC.py:9-25,49-63,97-113)."""
    out = [("sigma", (1,)), ("sigma_spat", (1,)), ("encoder.layer0.weight", (C, in_dim, 1)), ("encoder.layer0.bias", (C,))]

    def bn(prefix, c):
        return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
                (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]
    for i in range(num_layers):
        pc = f"encoder.blocks.PointCN_layer_{i}"
        nl = f"encoder.blocks.NonLocal_layer_{i}"
        out += [(pc + ".0.weight", (C, C, 1)), (pc + ".0.bias", (C,))] + bn(pc + ".1", C)
        out += [(nl + ".fc_message.0.weight", (C // 2, C, 1)), (nl + ".fc_message.0.bias", (C // 2,))] + bn(nl + ".fc_message.1", C // 2)
        out += [(nl + ".fc_message.3.weight", (C // 2, C // 2, 1)), (nl + ".fc_message.3.bias", (C // 2,))] + bn(nl + ".fc_message.4", C // 2)
        out += [(nl + ".fc_message.6.weight", (C, C // 2, 1)), (nl + ".fc_message.6.bias", (C,))]
        for p in ("q", "k", "v"):
            out += [(nl + f".projection_{p}.weight", (C, C, 1)), (nl + f".projection_{p}.bias", (C,))]
    out += [("classification.0.weight", (32, C, 1)), ("classification.0.bias", (32,)),
            ("classification.2.weight", (32, 32, 1)), ("classification.2.bias", (32,)),
            ("classification.4.weight", (1, 32, 1)), ("classification.4.bias", (1,))]
    return out


def _splitmix_uniform(tensor_id: int, numel: int, seed: int) -> np.ndarray:
    """Portable counter-based uniform(-1,1): splitmix64 of (seed, tensor_id, element index), float64."""
    with np.errstate(over="ignore"):
        x = (np.arange(numel, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        x = x + np.uint64((tensor_id + 1) * 0xD1B54A32D192ED03 % (1 << 64)) + np.uint64((seed + 1) * 0x8CB92BA72F3D8DD7 % (1 << 64))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53) * 2.0 - 1.0


def analytic_pointdsc_params(num_layers: int, C: int, sigma_d: float = 0.10, seed: int = 0, in_dim: int = 6) -> Dict[str, torch.Tensor]:
    """Deterministic closed-form parameter set (no blobs committed): every element is a splitmix64
    hash of (seed, tensor index, element index) mapped to uniform(-1,1) and scaled per tensor kind.
    Conv weights get a He-like amplitude (full-rank, keeps per-point variation alive through 12
    layers); BatchNorm gets non-trivial gamma/beta/mean/var so that folding is exercised."""
    P: Dict[str, torch.Tensor] = {}
    for t, (name, shape) in enumerate(pointdsc_param_shapes(num_layers, C, in_dim)):
        numel = int(np.prod(shape)) if len(shape) else 1
        u = _splitmix_uniform(t, numel, seed)
        if name == "sigma":
            val = np.array([1.0])
        elif name == "sigma_spat":
            val = np.array([sigma_d])
        elif name.endswith("num_batches_tracked"):
            P[name] = torch.tensor(0, dtype=torch.long)
            continue
        elif name.endswith("running_var"):
            val = 1.0 + 0.3 * u
        elif name.endswith("running_mean"):
            val = 0.1 * u
        elif name.endswith(".weight") and len(shape) == 1:
            val = 1.0 + 0.2 * u                    # BN gamma
        elif len(shape) == 3:
            fan_in = shape[1]
            if name == "encoder.layer0.weight":
                gain = 8.0                          # coordinates are O(0.1) m
            elif "projection_q" in name or "projection_k" in name:
                gain = 0.45                         # keeps attention logits O(1)
            elif "projection_v" in name:
                gain = 0.5
            elif "fc_message.6" in name:
                gain = 0.3                          # residual branch: no growth over 12 layers
            elif "PointCN" in name:
                gain = 1.05
            else:
                gain = 1.0
            val = u * math.sqrt(3.0) * math.sqrt(2.0 / fan_in) * gain
        else:
            val = 0.05 * u                         # biases / BN beta
        P[name] = torch.tensor(val.reshape(shape), dtype=torch.float32)
    return P


# ----------------------------------------------------------------------------------------------
# closed-form parameters / inputs for the backbone goldens (fusion, decoder): no blobs committed
# ----------------------------------------------------------------------------------------------
def hashed_tensor(shape, tensor_id: int, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    """Deterministic uniform(-scale, scale) tensor from the splitmix64 stream (tensor_id, element)."""
    numel = int(np.prod(shape)) if len(shape) else 1
    return torch.tensor((_splitmix_uniform(tensor_id, numel, seed) * scale).reshape(shape), dtype=torch.float32)


def analytic_state_dict(reference_state: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Fill a state dict (names/shapes taken from `reference_state`) with closed-form values: matrices / conv kernels
    get a 1/sqrt(fan_in) amplitude, norm scales sit around 1, biases are small, integer buffers are kept."""
    out = {}
    for t, (name, ref) in enumerate(reference_state.items()):
        if not torch.is_floating_point(ref) or name.endswith("attn_mask"):
            out[name] = ref.clone()
            continue
        shape = tuple(ref.shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[name] = hashed_tensor(shape, t, seed, math.sqrt(3.0 / fan_in) * 1.4)
        elif name.endswith("weight"):
            out[name] = 1.0 + hashed_tensor(shape, t, seed, 0.1)
        else:
            out[name] = hashed_tensor(shape, t, seed, 0.05)
    return out


# ----------------------------------------------------------------------------------------------
# data path in front of the network: preprocess_item -> resize -> collate, reference form on the CPU
# ----------------------------------------------------------------------------------------------
def tv_resize(img: torch.Tensor, size, mode: str) -> torch.Tensor:
    """torchvision 0.13 `transforms.functional.resize` on a tensor [C,H,W] (functional_tensor.py::resize; the package is
    third-party and absent here): integer / non-float inputs are cast to float32, resampled with torch `interpolate`
    (bilinear: align_corners=False, no antialias; nearest: legacy rule), then rounded and cast back."""
    out_dtype = img.dtype
    need_cast = out_dtype not in (torch.float32, torch.float64)
    x = img.to(torch.float32) if need_cast else img
    x = F.interpolate(x[None], size=list(size), mode=mode, align_corners=False if mode == "bilinear" else None)[0]
    if need_cast:
        if out_dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            x = torch.round(x)
        x = x.to(out_dtype)
    return x


def preprocess_item(item: dict) -> dict:
    """utils/data/common.py:41-75, pixel arithmetic included (rgb -> CHW float64 in [0,1])."""
    item = dict(item)
    item["metadata"] = dict(item["metadata"])
    item["rgb"] = torch.tensor(np.asarray(item["rgb"]).transpose(2, 0, 1) / 255.0)
    item["hw_size"] = tuple(np.asarray(item["mask"]).shape)
    item["mask"] = torch.tensor(np.asarray(item["mask"]))
    item["depth"] = torch.tensor(np.asarray(item["depth"]))
    item["orig_rgb"] = item["rgb"].clone()
    item["orig_depth"] = item["depth"].clone()
    item["eval_depth"] = item["depth"].clone()
    item["metadata"]["poses"] = [torch.tensor(np.asarray(v)) for v in item["metadata"]["poses"]]
    mask = torch.where(item["mask"] == item["metadata"]["mask_ids"][0], 1, 0)
    item["mask"] = mask
    ys, xs = np.nonzero(mask.numpy() == 1)
    y1, x1, y2, x2 = (int(ys.min()), int(xs.min()), int(ys.max()), int(xs.max())) if ys.size else (0, 0, 2, 2)
    item["metadata"]["boxes"] = torch.tensor([y1, x1, y2 - y1, x2 - x1])
    return item


def resize_item(item: dict, coords: torch.Tensor, size) -> Tuple[dict, torch.Tensor]:
    """utils/augmentations.py:133-149."""
    H, W = item["mask"].shape
    item["rgb"] = tv_resize(item["rgb"], size, "bilinear")
    item["mask"] = tv_resize(item["mask"][None], size, "nearest")[0]
    item["depth"] = tv_resize(item["depth"][None], size, "bilinear")[0]
    y1, x1, h, w = item["metadata"]["boxes"]
    item["metadata"]["boxes"] = torch.tensor([y1 * (size[0] / float(H)), x1 * (size[1] / float(W)), h * (size[0] / float(H)),
                                              w * (size[1] / float(W))])
    c = coords.clone().to(torch.float32)
    c[:, 0] *= torch.tensor(size[0], dtype=torch.float32) / torch.tensor(H, dtype=torch.float32)
    c[:, 1] *= torch.tensor(size[1], dtype=torch.float32) / torch.tensor(W, dtype=torch.float32)
    return item, c


def collate_side(items) -> dict:
    """One side of CollateWrapper.__call__ (datasets.py:202-228)."""
    return {
        "rgb": torch.stack([it["rgb"].squeeze() for it in items]).to(torch.float32),
        "mask": torch.stack([it["mask"].squeeze() for it in items]).to(torch.uint8),
        "depth": torch.stack([it["depth"].squeeze() for it in items]).to(torch.float32),
        "orig_depth": [it["orig_depth"].squeeze() for it in items],
        "camera": torch.stack([torch.as_tensor(it["camera"]).squeeze() for it in items]),
        "pose": torch.stack([it["metadata"]["poses"][0].squeeze() for it in items]),
        "box": torch.stack([torch.as_tensor(it["metadata"]["boxes"]).squeeze() for it in items]),
        "sizes": torch.stack([torch.as_tensor(it["hw_size"]) for it in items]),
        "instance_id": [it["instance_id"] for it in items],
    }
