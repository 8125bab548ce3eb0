"""ctypes binding of libadamdehaze_hip.so (the C ABI declared in include/adam_dehaze_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, a RuntimeError is
raised.  torch is used only for device memory (data_ptr) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

# ADH_LIB_PATH: A/B runs of an alternative build of the same library (development only)
_LIB_PATH = os.environ.get("ADH_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                           "libadamdehaze_hip.so")
_lib: Optional[C.CDLL] = None

ACT_NONE, ACT_RELU = 0, 1

i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class ConvDesc(C.Structure):
    """struct adh_conv_desc (include/adam_dehaze_hip.h)."""
    _fields_ = [
        ("in_", vp), ("out", vp), ("wp", vp), ("scale", vp), ("shift", vp), ("residual", vp), ("stats", vp),
        ("N", i32), ("IH", i32), ("IW", i32), ("Cin", i32), ("in_cstride", i32),
        ("OH", i32), ("OW", i32), ("Cout", i32), ("out_cstride", i32), ("res_cstride", i32),
        ("VH", i32), ("VW", i32),
        ("in_sy", i32), ("in_sx", i32),
        ("out_sy", i32), ("out_sx", i32), ("out_oy", i32), ("out_ox", i32),
        ("KH", i32), ("KW", i32),
        ("dy0", i32), ("dx0", i32), ("dstep_y", i32), ("dstep_x", i32),
        ("act", i32), ("NcP", i32),
    ]


class FpnLevels(C.Structure):
    """struct adh_fpn_levels."""
    _fields_ = [("f", C.c_void_p * 4), ("H", C.c_int32 * 4), ("W", C.c_int32 * 4), ("cs", C.c_int32 * 4), ("scale", C.c_float * 4),
                ("nlevels", C.c_int32)]


class AdamTensor(C.Structure):
    """struct adh_adam_tensor."""
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64), ("step", i32), ("repeats", i32)]


class WLayout(C.Structure):
    """struct adh_wlayout."""
    _fields_ = [
        ("K", i32), ("Nc", i32), ("KHt", i32), ("KWt", i32),
        ("tap_off0", i32), ("tap_off_sy", i32), ("tap_off_sx", i32),
        ("stride_k", i32), ("stride_n", i32),
    ]


PD = C.POINTER(ConvDesc)
PL = C.POINTER(WLayout)

# name -> argtypes (restype is always int)
_SIGNATURES = {
    "adh_version": [],
    "adh_conv_lds_bytes": [PD],
    "adh_conv_num_blocks": [PD],
    "adh_pack_weights": [vp, vp, PL, vp],
    "adh_conv_forward": [vp, PD],
    "adh_conv_wino_supported": [PD],
    "adh_conv_wino_forward": [vp, PD],
    "adh_conv_wino_num_blocks": [PD],
    "adh_conv_wino32_supported": [PD],
    "adh_conv_wino32_num_blocks": [PD],
    "adh_conv_wino32_forward": [vp, PD],
    "adh_conv_wino32_forward_multi": [vp, PD, i32],
    "adh_pack_weights_wino32": [vp, vp, PL, vp],
    "adh_conv_wino32_forward_bf16x3": [vp, PD],
    "adh_conv_wino32_forward_multi_bf16x3": [vp, PD, i32],
    "adh_pack_weights_wino32_bf16x3": [vp, vp, PL, vp],
    "adh_conv_wino43_supported": [PD],
    "adh_conv_wino43_num_blocks": [PD],
    "adh_conv_wino43_forward": [vp, PD],
    "adh_conv_wino43_dgrad_bnred": [vp, PD, vp],
    "adh_pack_weights_wino43": [vp, vp, PL, vp],
    "adh_conv_wino43_set_persistent": [i32],
    "adh_conv_wino43_forward_bf16x3": [vp, PD],
    "adh_conv_wino43_dgrad_bnred_bf16x3": [vp, PD, vp],
    "adh_pack_weights_wino43_bf16x3": [vp, vp, PL, vp],
    "adh_pack_weights_wino": [vp, vp, PL, vp],
    "adh_conv_wgrad": [vp, PD, vp, i32],
    "adh_conv_wgrad_wino_groups": [PD],
    "adh_conv_wgrad_wino": [vp, PD, vp, i32],
    "adh_wgrad_reduce_wino": [vp, vp, i32, i32, i32, PL, vp, i32],
    "adh_conv_wgrad_wino32_groups": [PD],
    "adh_conv_wgrad_wino32_classes": [PD],
    "adh_conv_wgrad_wino32_tiles": [PD],
    "adh_conv_wgrad_wino32_launches": [PD],
    "adh_conv_wgrad_wino32": [vp, PD, vp, i32],
    "adh_conv_wgrad_wino32_multi": [vp, PD, i32, vp, i32],
    "adh_wgrad_reduce_wino32": [vp, vp, i32, PD, i32, i32, PL, vp, i32],
    "adh_conv_wgrad_small_slabs": [PD],
    "adh_conv_wgrad_small": [vp, PD, vp, i32, i32],
    "adh_wgrad_reduce_small": [vp, vp, i32, i32, i32, PL, vp, i32],
    "adh_conv_fewout_supported": [PD],
    "adh_conv_fewout_forward": [vp, PD],
    "adh_pack_weights_fewout": [vp, vp, PL, vp],
    "adh_conv_fewin_supported": [PD],
    "adh_conv_fewin_num_blocks": [PD],
    "adh_conv_fewin_forward": [vp, PD],
    "adh_pack_weights_fewin": [vp, vp, PL, vp],
    "adh_conv_stem_num_blocks": [PD],
    "adh_conv_stem_forward": [vp, PD],
    "adh_pack_weights_stem": [vp, vp, PL, vp],
    "adh_conv_wgrad_stem_slabs": [PD],
    "adh_conv_wgrad_stem": [vp, PD, vp, i32],
    "adh_conv_wgrad_wino43_groups": [PD],
    "adh_conv_wgrad_wino43_strips": [PD],
    "adh_conv_wgrad_wino43": [vp, PD, vp, i32],
    "adh_wgrad_reduce_wino43": [vp, vp, i32, i32, i32, PL, vp, i32],
    "adh_conv_wgrad_slabs": [PD, i32],
    "adh_conv_wgrad_groups": [PD],
    "adh_wgrad_reduce": [vp, vp, i32, i32, i32, PL, vp, i32],
    "adh_wgrad_reduce_packed": [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32],
    "adh_bn_finalize": [vp, vp, i32, i32, i32, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp],
    "adh_bn_fold_eval": [vp, i32, vp, vp, vp, vp, f32, vp, vp, vp],
    "adh_bn_eval_bwd_vectors": [vp, i32, i32, vp, vp, vp, vp],
    "adh_bn_apply": [vp, vp, i32, vp, vp, vp, i32, i32, vp, i32, i64, i32, vp],
    "adh_bn_bwd_num_blocks": [i64, i32],
    "adh_bn_bwd_reduce": [vp, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp, vp],
    "adh_bn_bwd_finalize": [vp, vp, i32, i32, f64, vp, vp, vp, vp, i32, vp],
    "adh_bn_bwd_finalize_centered": [vp, vp, i32, i32, i32, f64, vp, vp, vp, vp, i32, vp],
    "adh_bn_partial_sums": [vp, vp, i32, i32, i32, f64, vp],
    "adh_bn_finalize_sums": [vp, vp, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp],
    "adh_bn_bwd_finalize_sums": [vp, vp, vp, i32, vp, vp, vp, vp, i32, vp],
    "adh_bn_bwd_apply": [vp, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, i32, i64, i32, vp, vp],
    "adh_cbam_pool": [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp],
    "adh_cbam_pool_num_blocks": [i32],
    "adh_cbam_mlp": [vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "adh_cbam_spatial_stats": [vp, vp, i32, vp, i32, i32, i32, vp, vp],
    "adh_cbam_apply": [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32],
    "adh_cbam_bwd_a": [vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, vp],
    "adh_cbam_bwd_b": [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, i32],
    "adh_cbam_bwd_b_num_blocks": [i32, i32, i32],
    "adh_cbam_bwd_c": [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, vp, i32],
    "adh_cbam_bwd_d_scratch_floats": [i32, i32, i32],
    "adh_cbam_bwd_d": [vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp],
    "adh_cbam_bwd_e": [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32],
    "adh_image_to_nhwc8": [vp, vp, i32, i32, i32, vp],
    "adh_image_normalize_to_nhwc8": [vp, vp, i32, i32, i32, f32, f32, f32, f32, f32, f32, vp],
    "adh_image_normalize_bwd": [vp, vp, i32, i32, i32, i32, f32, f32, f32, vp],
    "adh_nchw_to_nhwc": [vp, vp, i32, i32, i32, i32, vp, i32],
    "adh_nhwc_to_nchw": [vp, vp, i32, i32, i32, i32, i32, vp],
    "adh_head_blend": [vp, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp],
    "adh_head_blend_bwd": [vp, i32, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, i32],
    "adh_head_blend_bwd_num_blocks": [i32, i32, i32],
    "adh_softmax3": [vp, vp, f32, i32, vp],
    "adh_softmax3_bwd": [vp, vp, vp, i32, f32, i32, vp],
    "adh_soft_blend": [vp, vp, vp, vp, vp, i32, i64, vp],
    "adh_soft_blend_bwd": [vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, i32],
    "adh_argmax3": [vp, vp, i32, vp],
    "adh_route_compact": [vp, vp, i32, vp, vp],
    "adh_gather_images": [vp, vp, vp, i32, i64, vp],
    "adh_scatter_images": [vp, vp, vp, i32, i64, vp],
    "adh_reduce_num_blocks": [i64],
    "adh_l1_partial": [vp, vp, vp, i64, vp],
    "adh_mse_partial": [vp, vp, vp, i64, vp],
    "adh_sum_partials": [vp, vp, i32, f64, vp],
    "adh_l1_bwd": [vp, vp, vp, i64, f32, vp, vp],
    "adh_mse_bwd": [vp, vp, vp, i64, f32, vp, vp],
    "adh_cross_entropy3": [vp, vp, vp, i32, vp, vp],
    "adh_lpips_s2d": [vp, vp, i32, i32, i32, vp, vp, i32, i32, vp],
    "adh_lpips_s2d_bwd": [vp, vp, i32, i32, i32, vp, i32, i32, vp],
    "adh_lpips_layer": [vp, vp, vp, vp, i32, i32, i32, vp, i32],
    "adh_lpips_layer_num_blocks": [i32],
    "adh_lpips_layer_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "adh_rows_sum": [vp, vp, i32, i32, f32, vp, i32],
    "adh_adam_step": [vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, i32],
    "adh_adam_chunk_elems": [],
    "adh_upsample_nearest_add": [vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32],
    "adh_rpn_decode": [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, f32, f32, vp, vp],
    "adh_nms_words": [i32],
    "adh_nms_sorted": [vp, vp, vp, i32, f32, vp, vp],
    "adh_roi_align_fpn": [vp, vp, vp, i32, i32, vp],
    "adh_box_postprocess": [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp],
    "adh_augment_num_blocks": [i64],
    "adh_paired_augment": [vp, vp, vp, i32, i32, i32, vp, i32, vp],
    "adh_adam_multi": [vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, i32, i32, i32],
    "adh_apply_fog": [vp, vp, vp, vp, i32, i32, i32, vp],
    "adh_psnr_num_blocks": [i64],
    "adh_psnr": [vp, vp, vp, i32, i64, f32, vp, i32, vp, vp],
    "adh_ssim_num_blocks": [i32, i32],
    "adh_ssim_gray": [vp, vp, vp, i32, i32, i32, f32, vp, i32, vp],
    "adh_add_inplace": [vp, vp, vp, i64],
    "adh_axpby_strided": [vp, vp, i32, vp, i32, i64, i32, f32, f32],
    "adh_maxpool": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    "adh_maxpool_bwd": [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32],
    "adh_mul": [vp, vp, vp, vp, i64],
    "adh_avgpool": [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32],
    "adh_global_avgpool_bwd": [vp, vp, i32, i32, i32, vp, i32],
    "adh_bilinear": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32],
    "adh_bilinear_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32],
}

# functions that return a count / size rather than a status code
_VALUE_FUNCS = {"adh_version", "adh_conv_fewout_supported", "adh_conv_fewin_supported", "adh_conv_fewin_num_blocks", "adh_conv_wgrad_wino32_launches", "adh_conv_wino_supported", "adh_conv_wino_num_blocks", "adh_conv_wino32_supported", "adh_conv_wino43_supported", "adh_conv_wino43_num_blocks", "adh_conv_wino43_set_persistent",
                "adh_conv_wino32_num_blocks", "adh_conv_wgrad_wino_groups", "adh_conv_wgrad_wino32_groups", "adh_conv_wgrad_wino32_classes", "adh_conv_wgrad_wino32_tiles", "adh_conv_wgrad_wino43_groups", "adh_conv_wgrad_wino43_strips", "adh_conv_wgrad_small_slabs", "adh_conv_wgrad_stem_slabs", "adh_conv_stem_num_blocks", "adh_conv_wgrad_slabs", "adh_conv_wgrad_groups", "adh_conv_lds_bytes", "adh_conv_num_blocks", "adh_bn_bwd_num_blocks",
                "adh_cbam_pool_num_blocks", "adh_cbam_bwd_b_num_blocks", "adh_head_blend_bwd_num_blocks",
                "adh_reduce_num_blocks", "adh_lpips_layer_num_blocks", "adh_adam_chunk_elems", "adh_nms_words", "adh_augment_num_blocks", "adh_psnr_num_blocks", "adh_cbam_bwd_d_scratch_floats",
                "adh_ssim_num_blocks"}

_ERRORS = {-1: "ADH_E_ARG (bad argument: shape / alignment / null pointer)",
           -2: "ADH_E_LAUNCH (hip kernel launch failed)",
           -3: "ADH_E_UNSUPPORTED (configuration not supported by the kernels)"}


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the shared library (once). Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C adam-dehaze_amd/csrc`). There is no CPU fallback on the product path.")
        lib = C.CDLL(_LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (bench.py's
    roofline leg).  `work` is the algorithmic FLOP (or byte) count of the launch."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []   # (name, start_event, end_event, algorithmic work, executed work)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, work, work_exec in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += work
            a[3] += work_exec
        return {k: {"launches": v[0], "seconds": v[1], "work": v[2], "work_exec": v[3]} for k, v in agg.items()}


TIMER: Optional[KernelTimer] = None


def call(name: str, *args, work: float = 0.0, work_exec: Optional[float] = None, family: Optional[str] = None):
    """Invoke a status-returning entry point on the current stream; raise on failure.  `family`: the name the launch is
    accounted under by an installed timer (an entry point that launches another entry point's kernel)."""
    timer = TIMER
    if timer is not None and (family or name) in timer.names:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(stream_ptr(), *args)
        e1.record()
        timer.records.append((family or name, e0, e1, work, work if work_exec is None else work_exec))
    else:
        rc = getattr(load(), name)(stream_ptr(), *args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERRORS.get(rc, rc)}")


def value(name: str, *args) -> int:
    """Invoke a query entry point (returns a count/size; negative = error)."""
    assert name in _VALUE_FUNCS
    rc = getattr(load(), name)(*args)
    if rc < 0:
        raise RuntimeError(f"{name} failed: {_ERRORS.get(rc, rc)}")
    return rc


def require_cuda(t: torch.Tensor, what: str = "input") -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"adam-dehaze_amd: {what} is on {t.device}; this implementation runs on MI355X through the HIP library only "
            "(no CPU fallback). Move the module and its inputs to a cuda device.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"adam-dehaze_amd: {what} must be float32, got {t.dtype}")
