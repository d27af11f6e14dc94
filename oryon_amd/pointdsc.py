"""Drop-in counterparts of the reference's PointDSC entry points on the hot path:

    PointDSC(in_dim, num_layers, num_channels, num_iterations, ratio, inlier_threshold, sigma_d, k, nms_radius)
        .forward(data) -> {'final_trans', 'final_labels', 'M'}          (models/pointdsc/PointDSC.py:80-197)
    get_pointdsc_solver(ckpt_path, device)                              (utils/pointdsc/init.py:32-57)
    get_pointdsc_pose(model, pcd1, pcd2, device) -> [4,4] fp32 on cpu   (utils/pointdsc/init.py:10-29)
    rigid_transform_3d(A, B, weights=None, weight_threshold=0)          (models/pointdsc/common.py:7-45)

The nn.Module below only HOLDS parameters, under the reference's state-dict names, so reference checkpoints
(`model_best.pkl`) load unchanged; all arithmetic of the inference branch runs in liboryon_hip.so
(K3-K10).  Only the inference branch ('testing' in data) exists - training PointDSC is out of scope.
"""
from __future__ import annotations

import ctypes
import json
from typing import Dict, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib, ops
from ._lib import PointDSCConfig, check, lib, ptr, stream_ptr

N_ALIGN = 128


def rigid_transform_3d(A: Tensor, B: Tensor, weights: Optional[Tensor] = None, weight_threshold: float = 0) -> Tensor:
    """Weighted Kabsch, [bs,m,3] x2 (+ [bs,m]) -> [bs,4,4].  Like the reference, entries of `weights` below
    the threshold are zeroed IN PLACE (common.py:20)."""
    if weights is not None:
        weights[weights < weight_threshold] = 0
    return ops.kabsch_batched(A, B, weights)


class _NonLocalBlock(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.fc_message = nn.Sequential(
            nn.Conv1d(c, c // 2, kernel_size=1), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True),
            nn.Conv1d(c // 2, c // 2, kernel_size=1), nn.BatchNorm1d(c // 2), nn.ReLU(inplace=True),
            nn.Conv1d(c // 2, c, kernel_size=1))
        self.projection_q = nn.Conv1d(c, c, kernel_size=1)
        self.projection_k = nn.Conv1d(c, c, kernel_size=1)
        self.projection_v = nn.Conv1d(c, c, kernel_size=1)


class _NonLocalNet(nn.Module):
    def __init__(self, in_dim: int, num_layers: int, c: int):
        super().__init__()
        self.blocks = nn.ModuleDict()
        self.layer0 = nn.Conv1d(in_dim, c, kernel_size=1, bias=True)
        for i in range(num_layers):
            self.blocks[f"PointCN_layer_{i}"] = nn.Sequential(nn.Conv1d(c, c, kernel_size=1, bias=True), nn.BatchNorm1d(c),
                                                               nn.ReLU(inplace=True))
            self.blocks[f"NonLocal_layer_{i}"] = _NonLocalBlock(c)


class PointDSC(nn.Module):
    """Parameter container + HIP inference path; constructor defaults are the reference's (PointDSC.py:81-91)."""

    def __init__(self, in_dim=6, num_layers=6, num_channels=128, num_iterations=10, ratio=0.1, inlier_threshold=0.10,
                 sigma_d=0.10, k=40, nms_radius=0.10):
        super().__init__()
        self.num_iterations = num_iterations
        self.ratio = ratio
        self.num_channels = num_channels
        self.num_layers = num_layers
        self.in_dim = in_dim
        self.inlier_threshold = inlier_threshold
        self.k = k
        self.nms_radius = nms_radius
        self.sigma = nn.Parameter(torch.Tensor([1.0]).float(), requires_grad=True)
        self.sigma_spat = nn.Parameter(torch.Tensor([sigma_d]).float(), requires_grad=False)
        self.encoder = _NonLocalNet(in_dim, num_layers, num_channels)
        self.classification = nn.Sequential(
            nn.Conv1d(num_channels, 32, kernel_size=1, bias=True), nn.ReLU(inplace=True),
            nn.Conv1d(32, 32, kernel_size=1, bias=True), nn.ReLU(inplace=True),
            nn.Conv1d(32, 1, kernel_size=1, bias=True))
        for m in self.modules():                       # same initialisation scheme as PointDSC.py:115-121
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.xavier_normal_(m.weight, gain=1)
            elif isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._handle = None
        self._handle_version = None
        self._ws = None

    # ---- native handle management -------------------------------------------------------------
    def _params_version(self):
        """Cheap fingerprint of every parameter / buffer (in-place edits bump `_version`, re-assignment changes data_ptr); the
        tensor list is cached - walking state_dict() on every call cost ~1 ms in the per-sample loop."""
        tensors = getattr(self, "_tensor_cache", None)
        if tensors is None or self._tensor_cache_n != (len(self._parameters), self.training):
            tensors = [t for _, t in self.state_dict().items()]
            self._tensor_cache, self._tensor_cache_n = tensors, (len(self._parameters), self.training)
        return tuple((t._version, t.data_ptr()) for t in tensors)

    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .float() replace the parameter tensors
        self._tensor_cache = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._tensor_cache = None
        return super().load_state_dict(*a, **k)

    def train(self, mode: bool = True):
        """Inference-only holder: nothing here depends on the training flag, so an unchanged mode skips the walk over the ~200
        parameter-holding sub-modules (get_pointdsc_pose calls .eval() for every pair, like the reference)."""
        if self.training == mode and getattr(self, "_mode_walked", False):
            return self
        self._mode_walked = True
        return super().train(mode)

    def _ensure_handle(self, device):
        dev = _lib.require_gpu(device)
        ver = self._params_version()
        if self._handle is not None and self._handle_version == ver:
            return
        self._destroy_handle()
        cfg = PointDSCConfig(self.in_dim, self.num_layers, self.num_channels, self.num_iterations, float(self.ratio),
                             float(self.inlier_threshold), float(self.sigma_spat.detach().cpu()[0]), self.k,
                             float(self.nms_radius))
        h = ctypes.c_void_p()
        check(lib().oryon_pointdsc_create(ctypes.byref(h), ctypes.byref(cfg)), "oryon_pointdsc_create")
        try:
            for name, t in self.state_dict().items():
                if name.endswith("num_batches_tracked"):
                    continue
                host = t.detach().to("cpu", torch.float32).contiguous()
                check(lib().oryon_pointdsc_load_param(h, name.encode(), host.data_ptr(), host.numel()),
                      f"oryon_pointdsc_load_param({name})")
            with torch.cuda.device(dev):
                check(lib().oryon_pointdsc_finalize(h, stream_ptr(dev)), "oryon_pointdsc_finalize")
        except Exception:
            lib().oryon_pointdsc_destroy(h)
            raise
        self._handle, self._handle_version = h, ver

    def _destroy_handle(self):
        if getattr(self, "_handle", None) is not None:
            lib().oryon_pointdsc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def _workspace(self, B: int, n_cap: int, dev, slot: int = 0) -> Tensor:
        """One cached workspace per slot: registrations that may run concurrently (different HIP streams) use different slots."""
        need = lib().oryon_pointdsc_workspace_bytes(self._handle, B, n_cap)
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = torch.empty((need,), dtype=torch.uint8, device=dev)
            self._ws[slot] = ws
        return ws

    # ---- batched registration (the product path) ----------------------------------------------
    @ops._on_tensor_device
    def register(self, src: Tensor, tgt: Tensor, n: Tensor, status: Optional[Tensor] = None, want_labels: bool = False,
                 ws_slot: int = 0):
        """src,tgt [B,n_cap,3] fp32 CUDA (metres, n_cap % 128 == 0), n [B] int32 -> (T [B,4,4], labels|None, status [B])."""
        dev = _lib.require_gpu(src.device)
        self._ensure_handle(dev)
        B, n_cap = src.shape[0], src.shape[1]
        assert n_cap % N_ALIGN == 0 and src.dtype == torch.float32 and tgt.shape == src.shape
        ws = self._workspace(B, n_cap, dev, ws_slot)
        T = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
        labels = torch.empty((B, n_cap), dtype=torch.uint8, device=dev) if want_labels else None
        st_out = torch.empty((B,), dtype=torch.int32, device=dev)
        check(lib().oryon_pointdsc_register(self._handle, ptr(src.contiguous()), ptr(tgt.contiguous()), ptr(n), B, n_cap,
                                            ptr(status), ptr(ws), ws.numel(), ptr(T), ptr(labels), ptr(st_out),
                                            stream_ptr(dev)), "oryon_pointdsc_register")
        return T, labels, st_out

    # ---- stage-level views (parity tests) -----------------------------------------------------
    @ops._on_tensor_device
    def encode(self, src: Tensor, tgt: Tensor, n: Tensor):
        dev = _lib.require_gpu(src.device)
        self._ensure_handle(dev)
        B, n_cap = src.shape[0], src.shape[1]
        ws = self._workspace(B, n_cap, dev)
        feat = torch.empty((B, n_cap, self.num_channels), dtype=torch.float32, device=dev)
        conf = torch.empty((B, n_cap), dtype=torch.float32, device=dev)
        check(lib().oryon_pointdsc_encode(self._handle, ptr(src), ptr(tgt), ptr(n), B, n_cap, ptr(ws), ws.numel(), ptr(feat),
                                          ptr(conf), stream_ptr(dev)), "oryon_pointdsc_encode")
        return feat, conf

    def seed_cap(self, n_cap: int) -> int:
        return int(float(n_cap) * float(torch.tensor(self.ratio, dtype=torch.float32))) + 1

    @ops._on_tensor_device
    def pick_seeds_batched(self, src: Tensor, conf: Tensor, n: Tensor):
        dev = _lib.require_gpu(src.device)
        self._ensure_handle(dev)
        B, n_cap = src.shape[0], src.shape[1]
        S_cap = self.seed_cap(n_cap)
        seeds = torch.zeros((B, S_cap), dtype=torch.int32, device=dev)
        n_seeds = torch.empty((B,), dtype=torch.int32, device=dev)
        check(lib().oryon_pointdsc_seeds(self._handle, ptr(src), ptr(conf), ptr(n), B, n_cap, S_cap, ptr(seeds), ptr(n_seeds),
                                         stream_ptr(dev)), "oryon_pointdsc_seeds")
        return seeds, n_seeds

    @ops._on_tensor_device
    def hypotheses(self, src: Tensor, tgt: Tensor, feat: Tensor, n: Tensor, seeds: Tensor, n_seeds: Tensor):
        dev = _lib.require_gpu(src.device)
        self._ensure_handle(dev)
        B, n_cap = src.shape[0], src.shape[1]
        S_cap = seeds.shape[1]
        ws = self._workspace(B, n_cap, dev)
        seed_T = torch.zeros((B, S_cap, 4, 4), dtype=torch.float32, device=dev)
        fitness = torch.zeros((B, S_cap), dtype=torch.float32, device=dev)
        best = torch.empty((B,), dtype=torch.int32, device=dev)
        check(lib().oryon_pointdsc_hypotheses(self._handle, ptr(src), ptr(tgt), ptr(feat.contiguous()), ptr(n), ptr(seeds),
                                              ptr(n_seeds), B, n_cap, S_cap, ptr(ws), ws.numel(), ptr(seed_T), ptr(fitness),
                                              ptr(best), stream_ptr(dev)), "oryon_pointdsc_hypotheses")
        return seed_T, fitness, best

    @ops._on_tensor_device
    def refine(self, src: Tensor, tgt: Tensor, n: Tensor, T_in: Tensor) -> Tensor:
        dev = _lib.require_gpu(src.device)
        self._ensure_handle(dev)
        B, n_cap = src.shape[0], src.shape[1]
        T_out = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
        check(lib().oryon_pointdsc_refine(self._handle, ptr(src), ptr(tgt), ptr(n), B, n_cap, ptr(T_in.contiguous()), ptr(T_out),
                                          None, stream_ptr(dev)), "oryon_pointdsc_refine")
        return T_out

    # ---- reference-shaped forward -------------------------------------------------------------
    def forward(self, data: Dict) -> Dict:
        """data: corr_pos [bs,n,6], src_keypts/tgt_keypts [bs,n,3], 'testing' key required.  Returns final_trans [bs,4,4],
        final_labels [bs,n] float, M None (as the reference's inference branch; data['return_M'] = True adds the feature-similarity
        matrix of its training branch, PointDSC.py:158-163).
        The device encoder builds its input as cat(src,tgt) - mean over the rows (what utils/pointdsc/init.py:18-19 passes as
        corr_pos); a caller-supplied corr_pos that is anything else cannot be honoured and raises instead of being silently
        replaced."""
        if "testing" not in data:
            raise NotImplementedError("only the inference branch of PointDSC ('testing' in data) is implemented")
        src, tgt = data["src_keypts"], data["tgt_keypts"]
        dev = _lib.require_gpu(src.device)
        bs, n = src.shape[0], src.shape[1]
        cp = data.get("corr_pos")
        if cp is not None and n > 0:
            want = torch.cat([src, tgt], dim=-1).float()
            want = want - want.mean(dim=1, keepdim=True)
            scale = float(want.abs().max()) + 1e-12
            if tuple(cp.shape) != tuple(want.shape) or float((cp.to(dev).float() - want).abs().max()) > 1e-4 * scale + 1e-6:
                raise NotImplementedError("PointDSC.forward: corr_pos must be cat(src_keypts, tgt_keypts) minus its mean over the rows "
                                          "(utils/pointdsc/init.py:18-19); other encoder inputs are not supported by the device path")
        n_cap = ops.round_up(max(n, 1), N_ALIGN)
        sp = torch.zeros((bs, n_cap, 3), dtype=torch.float32, device=dev)
        tp = torch.zeros((bs, n_cap, 3), dtype=torch.float32, device=dev)
        sp[:, :n] = src.float()
        tp[:, :n] = tgt.float()
        nn_ = torch.full((bs,), n, dtype=torch.int32, device=dev)
        T, labels, _ = self.register(sp, tp, nn_, want_labels=True)
        M = None                                   # the reference's inference branch returns None here too (PointDSC.py:158-165)
        if data.get("return_M"):
            # on request: the feature-similarity matrix the reference builds on its training / validation branch
            # (PointDSC.py:158-163), from the device encoder's features - clamp(1 - (1 - f_i . f_j) / sigma^2, 0, 1), zero diagonal
            feat, _ = self.encode(sp, tp, nn_)
            f = torch.nn.functional.normalize(feat[:, :n], p=2, dim=-1)
            M = torch.clamp(1 - (1 - f @ f.transpose(1, 2)) / float(self.sigma) ** 2, min=0, max=1)
            M[:, torch.arange(n), torch.arange(n)] = 0
        return {"final_trans": T, "final_labels": labels[:, :n].float(), "M": M}


def get_pointdsc_pose(pointdsc_model: nn.Module, pcd1: Tensor, pcd2: Tensor, device: str) -> Tensor:
    """pcd1, pcd2: [N,3] corresponding points (metres) -> [4,4] fp32 on the CPU (utils/pointdsc/init.py:10-29)."""
    corr_pos = torch.cat([pcd1, pcd2], dim=-1)
    corr_pos = corr_pos - corr_pos.mean(0)
    data = {"corr_pos": corr_pos.unsqueeze(0).to(device).float(), "src_keypts": pcd1.unsqueeze(0).to(device).float(),
            "tgt_keypts": pcd2.unsqueeze(0).to(device).float(), "testing": True}
    pointdsc_model.eval()
    with torch.no_grad():
        res = pointdsc_model(data)
    return res["final_trans"].squeeze(0).cpu().to(torch.float32)


def get_pointdsc_solver(ckpt_path: str, device: str) -> nn.Module:
    """Builds PointDSC from the released config.json + model_best.pkl (utils/pointdsc/init.py:32-57)."""
    config = json.load(open(f"{ckpt_path}/snapshot/PointDSC_3DMatch_release/config.json", "r"))
    model = PointDSC(in_dim=config["in_dim"], num_layers=config["num_layers"], num_channels=config["num_channels"],
                     num_iterations=config["num_iterations"], ratio=config["ratio"], sigma_d=config["sigma_d"], k=config["k"],
                     nms_radius=config["inlier_threshold"]).to(device)
    state = torch.load(f"{ckpt_path}/snapshot/PointDSC_3DMatch_release/models/model_best.pkl", map_location=device)
    model.load_state_dict(state, strict=False)
    model.eval()
    for p in model.parameters():
        p.requires_grad = False
    return model
