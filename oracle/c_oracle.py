"""ctypes front-end of oracle/oryon_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboryon_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oryon_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_roi_from_mask.restype = ctypes.c_int
        _lib.orc_scale_validate_lift.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def roi_from_mask(mask: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(mask, dtype=np.int32)
    H, W = m.shape
    out = np.empty(H * W, dtype=np.int32)
    n = lib().orc_roi_from_mask(_p(m, ctypes.c_int32), H, W, _p(out, ctypes.c_int32))
    return out[:n].copy()


def match_lin(feat_a: np.ndarray, feat_q: np.ndarray, roi_a: np.ndarray, roi_q: np.ndarray, thr: float, anchor_rows=None,
              scalar: bool = False):
    """feat_*: [C,H,W] fp32; roi_*: int32 linear pixel indices.  Returns (min_dist, argmin, valid).
    anchor_rows: optional subset of anchor ROW indices to score (against all query rows); outputs then follow that subset.
    scalar: the plain one-chain-at-a-time loop nest (slow; pins the interleaved default bit for bit)."""
    fa = np.ascontiguousarray(feat_a, dtype=np.float32)
    fq = np.ascontiguousarray(feat_q, dtype=np.float32)
    C, HW = fa.shape[0], fa.shape[1] * fa.shape[2]
    ra = np.ascontiguousarray(roi_a, dtype=np.int32)
    rq = np.ascontiguousarray(roi_q, dtype=np.int32)
    n1, n2 = len(ra), len(rq)
    rows = None if anchor_rows is None else np.ascontiguousarray(anchor_rows, dtype=np.int32)
    n_out = n1 if rows is None else len(rows)
    md = np.empty(n_out, dtype=np.float32)
    am = np.empty(n_out, dtype=np.int32)
    va = np.empty(n_out, dtype=np.uint8)
    if scalar:
        assert rows is None
        lib().orc_match_f32_scalar(_p(fa, ctypes.c_float), _p(fq, ctypes.c_float), C, HW, _p(ra, ctypes.c_int32), n1,
                                   _p(rq, ctypes.c_int32), n2, ctypes.c_float(thr), _p(md, ctypes.c_float),
                                   _p(am, ctypes.c_int32), _p(va, ctypes.c_uint8))
    else:
        lib().orc_match_f32_rows(_p(fa, ctypes.c_float), _p(fq, ctypes.c_float), C, HW, _p(ra, ctypes.c_int32), n1,
                                 _p(rq, ctypes.c_int32), n2, ctypes.c_float(thr),
                                 None if rows is None else _p(rows, ctypes.c_int32), n_out, _p(md, ctypes.c_float),
                                 _p(am, ctypes.c_int32), _p(va, ctypes.c_uint8))
    return md, am, va.astype(bool)


def gather_normalise(feat: np.ndarray, roi_lin: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(feat, dtype=np.float32)
    C, HW = f.shape[0], f.shape[1] * f.shape[2]
    r = np.ascontiguousarray(roi_lin, dtype=np.int32)
    out = np.empty((len(r), C), dtype=np.float32)
    lib().orc_gather_normalise(_p(f, ctypes.c_float), C, HW, _p(r, ctypes.c_int32), len(r), _p(out, ctypes.c_float))
    return out


def match_presample(feat_a, feat_q, mask_a, mask_q, thr: float):
    """Same dict as oryon_oracle.match_presample (roi as [N,2] (y,x) int64)."""
    W = mask_a.shape[1]
    ra, rq = roi_from_mask(mask_a), roi_from_mask(mask_q)
    to_yx = lambda r, w: np.stack([r // w, r % w], axis=1).astype(np.int64).reshape(-1, 2)
    if len(ra) == 0 or len(rq) == 0:
        z = np.zeros(len(ra), dtype=np.float32)
        return dict(roi1=to_yx(ra, W), roi2=to_yx(rq, mask_q.shape[1]), min_dist=z,
                    argmin=np.zeros(len(ra), dtype=np.int64), valid=np.zeros(len(ra), dtype=bool))
    md, am, va = match_lin(feat_a, feat_q, ra, rq, thr)
    return dict(roi1=to_yx(ra, W), roi2=to_yx(rq, mask_q.shape[1]), min_dist=md, argmin=am.astype(np.int64), valid=va)


def lift_pair(depth_a, depth_q, cam_a, cam_q, corrs, feat_hw, size_a, size_q):
    da = np.ascontiguousarray(depth_a, dtype=np.float32)
    dq = np.ascontiguousarray(depth_q, dtype=np.float32)
    c = np.ascontiguousarray(corrs, dtype=np.int64)
    ca = np.ascontiguousarray(np.asarray(cam_a, dtype=np.float64).reshape(9))
    cq = np.ascontiguousarray(np.asarray(cam_q, dtype=np.float64).reshape(9))
    n = c.shape[0]
    ok = np.empty(n, dtype=np.uint8)
    pa = np.empty((n, 3), dtype=np.float32)
    pq = np.empty((n, 3), dtype=np.float32)
    m = lib().orc_scale_validate_lift(_p(c, ctypes.c_int64), n, int(feat_hw[0]), int(feat_hw[1]),
                                      _p(da, ctypes.c_float), int(size_a[0]), int(size_a[1]),
                                      _p(dq, ctypes.c_float), int(size_q[0]), int(size_q[1]),
                                      _p(ca, ctypes.c_double), _p(cq, ctypes.c_double),
                                      _p(ok, ctypes.c_uint8), _p(pa, ctypes.c_float), _p(pq, ctypes.c_float))
    return pa[:m].copy(), pq[:m].copy(), ok.astype(bool)
