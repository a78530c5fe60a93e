"""ctypes binding of libtrajdiff_hip.so (the C ABI declared in include/trajdiff_hip.h).

There is NO fallback: if the shared library is missing or a call fails, the product raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDX_LIB_PATH") or os.path.join(_HERE, "libtrajdiff_hip.so")   # (override: A/B runs of two builds)

TDX_F32, TDX_BF16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_SWISH = 0, 1, 2

_vp, _i32, _i64, _u64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class KSeg(C.Structure):
    _fields_ = [("src", _vp), ("up_scale", _f32), ("reserved0", _i32), ("src_ld", _i32), ("ch_off", _i32), ("cin", _i32), ("taps", _i32),
                ("stride", _i32), ("lin", _i32), ("lvirt", _i32), ("reserved", _i32)]


class SegNorm(C.Structure):
    _fields_ = [("mr", _vp), ("gamma", _vp), ("beta", _vp), ("cpg", _i32), ("act", _i32), ("mr_ld", _i32), ("mr_off", _i32)]


class ConvDesc(C.Structure):
    _fields_ = [("seg", KSeg * 4), ("nseg", _i32), ("dtype", _i32), ("w", _vp), ("bias", _vp), ("ebias", _vp),
                ("tbias", _vp), ("ebias_ld", _i32), ("ebias_off", _i32), ("res", _vp), ("res_ld", _i32), ("act", _i32),
                ("out", _vp), ("out_ld", _i32), ("out_off", _i32), ("B", _i32), ("lout", _i32), ("cout", _i32),
                ("reserved", _i32), ("gn_gamma", _vp), ("gn_beta", _vp), ("gn_out", _vp), ("gn_ld", _i32), ("gn_off", _i32),
                ("gn_act", _i32), ("gn_cpg", _i32), ("gn_eps", _f32), ("reserved2", _i32), ("ws", _vp), ("ws_bytes", _i64),
                ("norm", SegNorm * 4), ("stats_out", _vp), ("stats_out_ld", _i32), ("stats_out_off", _i32)]


class PackDesc(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i32), ("cin", _i32), ("taps", _i32), ("dst_ld", _i32), ("sr", _i64), ("sc", _i64),
                ("st", _i64), ("tap_map", _i32 * 3), ("reserved", _i32)]


_SIGNATURES = {
    "tdx_last_error": (C.c_char_p, []),
    "tdx_version": (_i32, []),
    "tdx_device_ok": (_i32, []),
    "tdx_conv_gemm": (_i32, [C.POINTER(ConvDesc), _vp]),
    "tdx_conv_gemm_gn_fusable": (_i32, [C.POINTER(ConvDesc)]),
    "tdx_conv_gemm_norm_fusable": (_i32, [C.POINTER(ConvDesc)]),
    "tdx_conv_gemm_force_small": (_i32, [_i32]),
    "tdx_conv_gemm_debug": (_i32, [_i32]),
    "tdx_groupnorm_act": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "tdx_groupnorm_finalize": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "tdx_stem_conv": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_stem_conv_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_head_conv_norm": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "tdx_head_conv": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_attention": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_embed_gather": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "tdx_ddpm_update": (_i32, [_vp, _vp, _vp, _vp, _i64, _u64, _vp, _i64, _vp]),
    "tdx_ddim_update": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "tdx_step_prepare": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "tdx_q_sample": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "tdx_didi_items": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "tdx_didi_denorm": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "tdx_cell_index": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "tdx_metric_workspace_bytes": (_i64, [_i32]),
    "tdx_point_distribution": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "tdx_length_distribution": (_i32, [_vp, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tdx_jsd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "tdx_train_workspace_floats": (_i64, []),
    "tdx_mse_loss_grad": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "tdx_mse_loss_grad_static": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "tdx_colsum": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "tdx_colsum_pair": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "tdx_conv_wgrad": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(KSeg), _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "tdx_conv_wgrad_bias": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(KSeg), _i32, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i64, _vp]),
    "tdx_conv_wgrad_debug": (_i32, [_i32]),
    "tdx_pack_weights": (_i32, [_vp, _i32, _vp]),
    "tdx_groupnorm_act_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32,
                                     _i32, _i32, _vp]),
    "tdx_groupnorm_act_dropout": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _f32, _u64, C.c_uint32, _vp, _vp]),
    "tdx_groupnorm_act_dropout_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32,
                                             _i32, _i32, _f32, _u64, C.c_uint32, _vp, _vp]),
    "tdx_interleave2": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_upsample_fold": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_act_fwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_act_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_traj_rowsum": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_embedding_bwd": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_attention_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "tdx_dropout": (_i32, [_vp, _i32, _i32, _i32, _f32, _u64, C.c_uint32, _vp, _vp]),
    "tdx_adamw_step": (_i32, [_vp, _vp, _vp, _vp, _i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i32, _vp]),
    "tdx_adamw_step_multi": (_i32, [_vp, _i32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i32, _vp]),
    "tdx_cross_attention": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_mha_small": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_chanfirst_to_rows": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_rows_to_chanfirst": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tdx_resample_rows": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_condmix_gather": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "tdx_stem_conv_n": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tdx_head_conv_n": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
}
EXPORTS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


class TdxError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdxError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the hot path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise TdxError(f"{what or 'libtrajdiff_hip'} failed (code {rc}): {lib().tdx_last_error().decode(errors='replace')}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return TDX_F32
    if dt == torch.bfloat16:
        return TDX_BF16
    raise TdxError(f"unsupported compute dtype {dt}; use torch.float32 or torch.bfloat16")


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise TdxError(f"{name} must live on an MI355X (got device {t.device}); the HIP hot path has no CPU fallback")
