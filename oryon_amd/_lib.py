"""ctypes binding of liboryon_hip.so (the C ABI declared in include/oryon_hip.h).

The product path has NO fallback: if the shared library is missing or the device is not gfx950, every
operator raises.  torch is used only as plumbing (device memory, streams)."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboryon_hip.so")
_lib = None


class OryonError(RuntimeError):
    pass


class EngineConfig(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("FH", c_int), ("FW", c_int), ("HA", c_int), ("WA", c_int), ("HQ", c_int), ("WQ", c_int),
                ("layout", c_int), ("dist_th", c_float), ("n_corrs", c_int), ("src_sampling", c_int), ("seed", c_uint64),
                ("round_f16", c_int), ("n_slots", c_int), ("overlap", c_int), ("gather_sets", c_int), ("reg_streams", c_int), ("reg_lag", c_int), ("screen", c_int), ("sample_first", c_int), ("x3_prefetch", c_int), ("stream_roles", c_int)]


class DecoderWeights(ctypes.Structure):
    """oryon_decoder_weights_t (include/oryon_hip.h): device pointers to StandardDecoder's fp32 parameters."""
    _fields_ = [("gp_w", c_void_p * 2), ("gp_b", c_void_p * 2), ("up_w", c_void_p * 3), ("up_b", c_void_p * 3), ("c1_w", c_void_p * 3),
                ("n1_g", c_void_p * 3), ("n1_b", c_void_p * 3), ("c2_w", c_void_p * 3), ("n2_g", c_void_p * 3), ("n2_b", c_void_p * 3),
                ("head_w", c_void_p), ("head_b", c_void_p)]


class FusionClassWeights(ctypes.Structure):
    """oryon_fusion_class_weights_t (include/oryon_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "wq", "bq", "wk", "bk", "wv", "bv", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")]


class PointDSCConfig(ctypes.Structure):
    _fields_ = [("in_dim", c_int), ("num_layers", c_int), ("num_channels", c_int), ("num_iterations", c_int),
                ("ratio", c_float), ("inlier_threshold", c_float), ("sigma_d", c_float), ("k", c_int),
                ("nms_radius", c_float)]


_P = c_void_p
_PROTOS = {
    "oryon_version": (c_char_p, []),
    "oryon_last_error": (c_char_p, []),
    "oryon_device_check": (c_int, [c_int]),
    "oryon_profile_events": (c_int, [_P, _P]),
    "oryon_dominant_kernel": (c_char_p, []),
    "oryon_quick_gelu_bf16": (c_int, [_P, _P, ctypes.c_int64, _P]),
    "oryon_add_layernorm_bf16": (c_int, [_P, _P, _P, _P, ctypes.c_int64, c_int, c_float, _P, _P, _P]),
    "oryon_add_layernorm_f32": (c_int, [_P, _P, _P, _P, ctypes.c_int64, c_int, c_float, _P, _P, _P]),
    "oryon_swin_window_attention_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_swin_window_attention_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_round_to_f16_f32": (c_int, [_P, _P, ctypes.c_int64, _P]),
    "oryon_rgb_resize_bilinear": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_resize_bilinear_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_roi_compact": (c_int, [_P, c_int, c_int, _P, _P, _P]),
    "oryon_mask_from_logits": (c_int, [_P, c_int64, c_float, _P, _P]),
    "oryon_mask_resize_nearest": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_roi_subsample": (c_int, [_P, _P, c_int, c_int, c_int, c_uint64, _P, _P]),
    "oryon_gather_normalise_f32": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P, _P]),
    "oryon_match_workspace_bytes": (c_size_t, [c_int, c_int]),
    "oryon_match_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "oryon_match_screened_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "oryon_gather_normalise_q8": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "oryon_gather_q8": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "oryon_match_screened8_raw_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oryon_match_screened8_raw": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                          _P, _P, c_float, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "oryon_match_corrs_i8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "oryon_match_corrs_i8": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                     _P, _P, c_float, c_int, c_int, c_int, c_uint64, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P,
                                     c_size_t, _P]),
    "oryon_x3_range_flag": (c_int, [POINTER(c_int), c_int, _P]),
    "oryon_gather_mx6": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "oryon_match_corrs_mx6": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int,
                                      _P, _P, c_float, c_int, c_int, c_int, c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P,
                                      c_size_t, _P]),
    "oryon_gather_mx6_x3": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "oryon_match_corrs_mx6_x3": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                         _P, _P, c_float, c_int, c_int, c_int, c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P,
                                         c_size_t, _P]),
    "oryon_match_screened8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "oryon_match_screened8": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P,
                                      _P, _P, c_size_t, _P]),
    "oryon_match_screened": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "oryon_sample_first_gate": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "oryon_sample_first_merge": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    "oryon_select_corrs": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_uint64, _P, _P,
                                   _P, _P, _P, _P, _P]),
    "oryon_lift_pairs": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, _P, _P, _P,
                                 _P, _P, _P]),
    "oryon_lift_points": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "oryon_kabsch_batched": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "oryon_split_f16x3": (c_int, [_P, c_int64, _P, _P, _P]),
    "oryon_linear_f16x3": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "oryon_linear_f16x3_acc": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "oryon_mha_f16x3": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_pose_metrics": (c_int, [_P, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "oryon_pose_bop_workspace_bytes": (c_size_t, [c_int, c_int]),
    "oryon_pose_bop_errors": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P]),
    "oryon_engine_arena_bytes": (c_size_t, [POINTER(EngineConfig), c_void_p]),
    "oryon_engine_create": (c_int, [POINTER(c_void_p), POINTER(EngineConfig), c_void_p, _P, c_size_t]),
    "oryon_engine_destroy": (None, [c_void_p]),
    "oryon_engine_warm_streams": (c_int, []),
    "oryon_engine_set_stream_roles": (c_int, [c_void_p, c_int]),
    "oryon_engine_stream_roles": (c_int, [c_void_p, POINTER(c_int)]),
    "oryon_engine_buffer": (c_int, [c_void_p, c_int, c_char_p, POINTER(c_size_t), POINTER(c_size_t)]),
    "oryon_engine_geometry": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "oryon_engine_submit": (c_int, [c_void_p, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "oryon_engine_wait": (c_int, [c_void_p, c_int, _P]),
    "oryon_engine_set_timing": (c_int, [c_void_p, c_int]),
    "oryon_engine_timing": (c_int, [c_void_p, c_int64, POINTER(c_float)]),
    "oryon_engine_elapsed": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int, POINTER(c_float)]),
    "oryon_engine_gather_ms": (c_int, [c_void_p, c_int64, POINTER(c_float)]),
    "oryon_engine_config_bytes": (c_size_t, []),
    "oryon_engine_host_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "oryon_engine_x3_steps": (c_int, [c_void_p, POINTER(c_int64)]),
    "oryon_engine_feedback": (c_int, [_P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "oryon_fusion_window_attention_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "oryon_fusion_class_layer_f32": (c_int, [_P, _P, POINTER(FusionClassWeights), c_int, _P, _P]),
    "oryon_conv24_image_bytes": (c_int64, [c_int, c_int, c_int]),
    "oryon_conv24_pack_f16x3": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "oryon_conv24_f16x3": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "oryon_decoder_create": (c_int, [POINTER(DecoderWeights), POINTER(c_void_p), _P]),
    "oryon_decoder_destroy": (None, [c_void_p]),
    "oryon_decoder_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "oryon_decoder_workspace_layout": (c_int, [c_int, c_int, c_int, POINTER(c_int64)]),
    "oryon_decoder_forward": (c_int, [c_void_p, _P, _P, _P, c_int, c_int, c_int, _P, c_int64, _P, _P, c_int, c_int, _P]),
    "oryon_pointdsc_create": (c_int, [POINTER(c_void_p), POINTER(PointDSCConfig)]),
    "oryon_pointdsc_destroy": (None, [c_void_p]),
    "oryon_pointdsc_load_param": (c_int, [c_void_p, c_char_p, _P, c_int64]),
    "oryon_pointdsc_finalize": (c_int, [c_void_p, _P]),
    "oryon_pointdsc_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "oryon_pointdsc_register": (c_int, [c_void_p, _P, _P, _P, c_int, c_int, _P, _P, c_size_t, _P, _P, _P, _P]),
    "oryon_pointdsc_encode": (c_int, [c_void_p, _P, _P, _P, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "oryon_pointdsc_seeds": (c_int, [c_void_p, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "oryon_pointdsc_hypotheses": (c_int, [c_void_p, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P, _P,
                                          _P, _P]),
    "oryon_pointdsc_refine": (c_int, [c_void_p, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
}

EXPORTS = tuple(_PROTOS)


def lib():
    """Load liboryon_hip.so (once).  Raises OryonError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OryonError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C oryon_amd/csrc` - there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        missing = [name for name in _PROTOS if not hasattr(L, name)]
        if missing:
            raise OryonError(f"{LIB_PATH} lacks symbols declared in include/oryon_hip.h: {missing} (stale build?)")
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.oryon_engine_config_bytes() != ctypes.sizeof(EngineConfig):
            raise OryonError(f"{LIB_PATH}: oryon_engine_config_t is {L.oryon_engine_config_bytes()} bytes in the library, "
                             f"{ctypes.sizeof(EngineConfig)} in oryon_amd/_lib.py (stale build?)")
        _lib = L
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        raise OryonError(f"{what or 'liboryon_hip'} failed ({code}): {lib().oryon_last_error().decode()}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI takes contiguous buffers"
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    """The current torch HIP stream of `device` as a raw hipStream_t."""
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise OryonError("oryon_amd operators run on an MI355X (torch device 'cuda'); no CPU fallback exists")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    check(lib().oryon_device_check(idx), "oryon_device_check")
    return torch.device("cuda", idx)
