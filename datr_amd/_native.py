"""ctypes binding of libdatr_hip.so (the C ABI declared in include/datr_hip.h).

The library is built in-tree by `datr_amd/csrc/Makefile` (see __graft_entry__.build) and is
the ONLY compute backend of the hot path: there is no CPU or eager-PyTorch fallback.  If it is
missing or an entry point is absent the import of this module raises.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DATR_HIP_LIB points at an alternative build of the same ABI (e.g. the -DDATR_PROBE build used
# for kernel ablations); there is still no non-native fallback.
LIB_PATH = os.environ.get("DATR_HIP_LIB") or os.path.join(_HERE, "lib", "libdatr_hip.so")
ABI_VERSION = 2

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p

# name -> argtypes ; every function returns int (DATR_OK / DATR_E*)
_SIGNATURES = {
    "datr_msda_forward_f32": [_vp] * 5 + [_i64] * 7 + [_vp, _vp],
    "datr_msda_forward_f64": [_vp] * 5 + [_i64] * 7 + [_vp, _vp],
    "datr_msda_backward_f32": [_vp] * 6 + [_i64] * 7 + [_vp, _vp, _vp, _vp],
    "datr_msda_backward_f64": [_vp] * 6 + [_i64] * 7 + [_vp, _vp, _vp, _vp],
    "datr_msda_forward_tiled_f32": [_vp] * 7 + [_i64] * 7 + [_vp, _vp],
    "datr_msda_backward_tiled_f32": [_vp] * 8 + [_i64] * 7 + [_vp, _vp, _vp, _vp],
    "datr_msda_backward_pyramid_f32": [_vp] * 9 + [_i64] * 7 + [_vp, _vp, _vp, _vp],
    "datr_msda_backward_pyramid_query_f32": [_vp] * 7 + [_i64] * 7 + [_vp, _vp, _vp],
    "datr_msda_backward_strided_f32": [_vp] * 6 + [_i64] * 7 + [_vp, _i64, _vp, _vp, _vp],
    "datr_msda_backward_query_tiled_f32": [_vp] * 8 + [_i64] * 7 + [_vp, _vp, _vp, _vp],
    "datr_msda_forward_pyramid_f32": [_vp] * 8 + [_i64] * 7 + [_vp, _vp],
    "datr_msda_pyramid_plan": [_vp, _vp] + [_i64] * 7 + [_vp, _vp],
    "datr_msda_uses_fast_path": [_i64] * 5,
    "datr_affine_act_forward_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, _vp, _vp],
    "datr_affine_act_backward_f32": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, _vp, _vp, _vp],
    "datr_affine_act_backward2_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int, _vp, _vp, _vp],
    "datr_add_layernorm_forward_f32": [_vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp, _vp, _vp],
    "datr_add_layernorm_backward_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp],
    "datr_add_layernorm_backward_colsum_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "datr_add_layernorm_forward_query_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "datr_add_layernorm_backward_fanin_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "datr_normalize_pad_u8_f32": [_vp, _i64, _i64, _vp, _vp, _i64, _i64, ctypes.c_int, _vp, _vp, _vp],
    "datr_lsap_f32": [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "datr_colsum_f32": [_vp, _i64, _i64, _vp, _vp, _vp],
    "datr_msda_prologue_forward_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "datr_msda_prologue_backward_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "datr_mha_forward_d32_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_float, _vp, _vp, _vp],
    "datr_mha_backward_d32_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_float,
                                  _vp, _vp, _vp, _vp, _vp],
    "datr_match_cost_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, ctypes.c_float,
                            ctypes.c_float, ctypes.c_float, _vp, _vp, _vp],
    "datr_box_loss_forward_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "datr_box_loss_backward_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "datr_sine_embed_f32": [_vp, _vp, _i64, _i64, _vp, _vp],
    "datr_topk_rows_f32": [_vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "datr_nms_f32": [_vp, _vp, _vp, _i64, ctypes.c_float, _vp, _vp, _vp],
    "datr_gemm_f32": [ctypes.c_int, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _vp],
    "datr_relu_bwd_bias_f32": [_vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "datr_resize_bilinear_u8": [_vp, _i64, _i64, ctypes.c_int, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp,
                                _vp, _vp],
    "datr_conv3x3_wino_wgrad_nhwc_f32": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "datr_zero_rows_f32": [_vp, _vp, _i64, _i64, _vp],
    "datr_ema_update_f32": [_vp, _vp, _i64, ctypes.c_double, _vp],
    "datr_refine_boxes_forward_f32": [_vp, _vp, _i64, ctypes.c_float, _vp, _vp],
    "datr_layernorm_class_max_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_float, _vp, _vp, _vp, _vp],
    "datr_class_prototypes_forward_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "datr_class_prototypes_backward_f32": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp],
    "datr_contrast_loss_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, _vp, _vp, _vp, _vp],
    "datr_stack_linear_forward_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "datr_stack_linear_backward_f32": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "datr_refine_boxes_backward_f32": [_vp, _vp, _vp, _i64, ctypes.c_float, _vp, _vp, _vp],
    "datr_grad_norm_clip_coef_f32": [_vp, _vp, _i64, ctypes.c_float, _vp, _vp, _vp],
    "datr_adamw_step_f32": [_vp, _i64, _vp, _i64, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp],
    "datr_pixel_ops_u8": [_vp, _vp, _i64, _vp, _i64, _vp, _vp],
    "datr_box_blur_u8": [_vp, _vp, _i64, _i64, _i64, ctypes.c_uint32, ctypes.c_uint32, _i64, _vp],
    "datr_groupnorm_nhwc_forward_f32": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _vp, _vp, _vp,
                                        _vp, _vp],
    "datr_groupnorm_nhwc_backward_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp],
    "datr_wino_weights_f32": [_vp, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, _vp, _vp],
    "datr_wino_weights_pair_f32": [_vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp],
    "datr_conv3x3_wino_nhwc_f32": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, _vp],
    "datr_add_n_f32": [_vp, _i64, _i64, _vp, _vp],
    "datr_even_pixels_nhwc_f32": [_vp, _i64, _i64, _i64, _i64, _vp, _vp],
    "datr_even_pixels_scatter_nhwc_f32": [_vp, _i64, _i64, _i64, _i64, _vp, _vp],
    "datr_stem_conv7x7_bn_relu_nhwc_f32": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp],
    "datr_conv3x3s2_weights_f32": [_vp] + [_i64] * 6 + [_vp, _vp, _vp],
    "datr_conv3x3s2_forward_nhwc_f32": [_vp, _vp, _vp, _vp, ctypes.c_float] + [_i64] * 5 + [_vp, _vp, _i64, _vp],
    "datr_conv3x3s2_dgrad_nhwc_f32": [_vp, _vp] + [_i64] * 5 + [_vp, _vp, _i64, _vp],
    "datr_conv3x3s2_wgrad_nhwc_f32": [_vp, _vp] + [_i64] * 5 + [_vp] + [_i64] * 4 + [_vp, _i64, _vp],
    "datr_conv3x3_cout1_forward_f32": [_vp, _i64, _i64, _i64, _vp, _vp, _vp],
    "datr_conv3x3_cout1_backward_f32": [_vp, _i64, _i64, _i64, _vp, ctypes.c_float, _vp, _vp, _vp, _vp],
    "datr_focal_loss_forward_f32": [_vp, _vp, _i64, _i64, _i64, ctypes.c_float, ctypes.c_float,
                                    _vp, _vp, _vp],
    "datr_focal_loss_backward_f32": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_float,
                                     ctypes.c_float, _vp, _vp],
}


class GemmEpilogue(ctypes.Structure):
    """`datr_gemm_epilogue` of include/datr_hip.h."""
    _fields_ = [("scale", _vp), ("shift", _vp), ("residual", _vp), ("ldr", _i64), ("gate", _vp), ("ldg", _i64),
                ("relu", ctypes.c_int), ("colsum", _vp), ("rowsum_a", _vp)]


class WinoLevel(ctypes.Structure):
    """`datr_wino_level` of include/datr_hip.h."""
    _fields_ = [("x", _vp), ("y", _vp), ("gate", _vp), ("H", _i64), ("W", _i64)]


class C1Level(ctypes.Structure):
    """`datr_c1_level` of include/datr_hip.h."""
    _fields_ = [("x", _vp), ("y", _vp), ("dx", _vp), ("H", _i64), ("W", _i64)]


class WinoWgradLevel(ctypes.Structure):
    """`datr_wino_wgrad_level` of include/datr_hip.h."""
    _fields_ = [("x", _vp), ("dy", _vp), ("H", _i64), ("W", _i64)]


class NativeLibraryError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C datr_amd/csrc`.  datr_amd has no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.datr_abi_version.restype = ctypes.c_int
    if lib.datr_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"libdatr_hip.so ABI {lib.datr_abi_version()} != expected {ABI_VERSION}")
    lib.datr_strerror.restype = ctypes.c_char_p
    lib.datr_strerror.argtypes = [ctypes.c_int]
    lib.datr_focal_scratch_floats.restype = ctypes.c_int64
    lib.datr_focal_scratch_floats.argtypes = [_i64, _i64]
    lib.datr_add_layernorm_partial_floats.restype = ctypes.c_int64
    lib.datr_add_layernorm_partial_floats.argtypes = [_i64]
    lib.datr_relu_bwd_bias_partial_rows.restype = ctypes.c_int64
    lib.datr_relu_bwd_bias_partial_rows.argtypes = [_i64]
    lib.datr_groupnorm_partial_floats.restype = ctypes.c_int64
    lib.datr_groupnorm_partial_floats.argtypes = [_i64, _i64, _i64, _i64]
    lib.datr_wino_wgrad_partial_floats.restype = ctypes.c_int64
    lib.datr_wino_wgrad_partial_floats.argtypes = [_vp, _i64, _i64, _i64, _i64]
    lib.datr_conv3x3s2_workspace_floats.restype = ctypes.c_int64
    lib.datr_conv3x3s2_workspace_floats.argtypes = [_i64] * 5
    lib.datr_conv3x3_cout1_partial_floats.restype = ctypes.c_int64
    lib.datr_conv3x3_cout1_partial_floats.argtypes = [_vp, _i64, _i64]
    lib.datr_ema_piece_elements.restype = ctypes.c_int64
    lib.datr_ema_piece_elements.argtypes = []
    lib.datr_adamw_piece_elements.restype = ctypes.c_int64
    lib.datr_adamw_piece_elements.argtypes = []
    lib.datr_gemm_workspace_floats.restype = ctypes.c_int64
    lib.datr_gemm_workspace_floats.argtypes = [ctypes.c_int, _i64, _i64, _i64, ctypes.c_int]
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    return lib


lib = _load()


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what}: {lib.datr_strerror(code).decode()} (code {code})")


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`with on_device(t.device):` around a native launch -- `torch.cuda.device(device)` when `device` is not the
    process' current one, nothing otherwise (one rank per GPU: the current device IS the tensors' device, and the
    context manager's two device switches per launch, ~800 launches per training step, were host time)."""
    if not isinstance(device, torch.device):
        device = torch.device(device)
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_ptr(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device`, as an integer for ctypes.  Through torch's raw-stream
    accessor where it exists (0.3 us; `torch.cuda.current_stream(device).cuda_stream` builds a Stream object per call,
    5 us, ~220 times per training step)."""
    if _RAW_STREAM is not None:
        idx = device.index
        return _RAW_STREAM(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream
